#!/usr/bin/env python3
"""A/B under rocprofv3: plain regens without / with the column-count hook of k_tile_dedupe.  usage: ab_colcnt.py k reps"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sdmesh_pkg  # noqa: E402
k, reps = int(sys.argv[1]), int(sys.argv[2])
pkg = sdmesh_pkg.load()
h = pkg.CudaHandler(0, "sdobj")
h.set_option("timing", 1)
for mode in (0, 1, 0, 1):
    h.set_option("colcnt_debug", mode)
    h.regen(k)
    ms = sorted(h.regen(k).device_ms for _ in range(reps))
    print("colcnt", mode, "median device ms", round(ms[len(ms) // 2], 4))
