#!/usr/bin/env python3
"""A/B of one integer option on the device-resident regen.  usage: pmult.py option v1,v2,... [scene]"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sdmesh_pkg
pkg = sdmesh_pkg.load()
opt, vals = sys.argv[1], [int(v) for v in sys.argv[2].split(",")]
scene = sys.argv[3] if len(sys.argv) > 3 else "sdobj"
h = pkg.CudaHandler(0, scene)
for rnd in range(2):
    for k in (3, 4, 5):
        for v in vals:
            h.set_option(opt, v)
            h.regen(k); h.regen(k)
            ms = [h.regen(k).device_ms for _ in range(60)]
            print(f"N={32 << k} {opt}={v}: med {statistics.median(ms):.4f} min {min(ms):.4f}", flush=True)
