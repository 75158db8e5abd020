#!/usr/bin/env python3
"""A/B of chain options by per-stage device time (option "profile") and by whole-regen device / wall time.
usage: stage_ab.py scene levels opt=v[,opt=v...] [opt=v,...] ...   (each argument is one configuration; "-" = defaults)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sdmesh_pkg  # noqa: E402

scene, k = sys.argv[1], int(sys.argv[2])
pkg = sdmesh_pkg.load()
for cfg in sys.argv[3:]:
    h = pkg.CudaHandler(0, scene)
    if cfg != "-":
        for kv in cfg.split(","):
            name, v = kv.split("=")
            h.set_option(name, int(v))
    h.regen(k)
    for _ in range(10):
        h.regen(k)
    walls = []
    for _ in range(3):                       # three timed bursts: the spread says how much of a difference is noise
        t0 = time.perf_counter()
        for _ in range(200):
            h.regen(k)
        walls.append((time.perf_counter() - t0) / 200)
    wall = min(walls)
    h.set_option("timing", 1)                # the event pair costs wall time: device time is measured in its own pass
    dev = [h.regen(k).device_ms for _ in range(100)]
    h.set_option("timing", 0)
    h.set_option("profile", 1)
    acc = {}
    for _ in range(30):
        h.regen(k)
        per = {}
        for name, ms in h.profile_stages():
            per[name] = per.get(name, 0.0) + ms
        for name, ms in per.items():
            acc.setdefault(name, []).append(ms)
    st = {n: round(float(np.median(v)) * 1e3, 1) for n, v in acc.items()}
    print(f"{cfg:40s} wall {wall * 1e3:.4f} (max {max(walls) * 1e3:.4f}) ms  device {np.median(dev):.4f} ms  stages(us) {st}", flush=True)
    h.close()
