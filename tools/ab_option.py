#!/usr/bin/env python3
"""A/B of one handle option on the device-resident regen: median device ms (HIP events) per value, interleaved.
usage: ab_option.py scene k option v0 v1 [v2 ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sdmesh_pkg  # noqa: E402
scene, k, opt, vals = sys.argv[1], int(sys.argv[2]), sys.argv[3], [int(v) for v in sys.argv[4:]]
pkg = sdmesh_pkg.load()
h = pkg.CudaHandler(0, scene)
h.set_option("timing", 1)
h.regen(k)
for rnd in range(3):
    for v in vals:
        h.set_option(opt, v)
        h.regen(k); h.regen(k)
        ms = sorted(h.regen(k).device_ms for _ in range(80))
        print(opt, v, "median device ms", round(ms[40], 4), "p10", round(ms[8], 4), flush=True)
