#!/usr/bin/env python3
"""Device time of a regen under handle options: python tools/regen_opt.py sdobj 4 project_grid=0 project_grid=2048 project_batch=5+project_grid=4096
(every combination names all the options it varies; medians of `reps` regens, two passes over the list)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sdmesh_pkg
pkg = sdmesh_pkg.load()
scene, k = sys.argv[1], int(sys.argv[2])
combos = [[(kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split("+")] for a in sys.argv[3:]] or [[]]
h = pkg.CudaHandler(0, scene)
h.set_option("timing", 1)
for _ in range(300): h.regen(k)
for rnd in range(2):
    for combo in combos:
        for name, val in combo: h.set_option(name, val)
        for _ in range(50): h.regen(k)
        t0 = time.perf_counter()
        ms = [h.regen(k).device_ms for _ in range(300)]
        wall = (time.perf_counter() - t0) / 300 * 1e3
        print(json.dumps({"opt": "+".join(f"{a}={b}" for a, b in combo), "device_ms": round(float(np.median(ms)), 5), "wall_ms": round(wall, 5)}), flush=True)
