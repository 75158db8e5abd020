#!/usr/bin/env python3
"""Throughput with several regens in flight (one handle each) vs one at a time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sdmesh_pkg
pkg = sdmesh_pkg.load()
scene = sys.argv[1] if len(sys.argv) > 1 else "sdobj"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
for depth in (1, 2, 3, 4):
    hs = [pkg.CudaHandler(0, scene) for _ in range(depth)]
    for h in hs:
        h.regen(k); h.set_option("graph", 1); h.regen(k); h.regen(k)
    steps = 120
    t0 = time.perf_counter()
    for h in hs:
        h.regen_begin(k)
    done = 0
    i = 0
    while done < steps:
        h = hs[i % depth]
        info = h.regen_end()
        done += 1
        if done + depth - 1 < steps + depth - 1 and done <= steps - depth:
            h.regen_begin(k)
        i += 1
    dt = time.perf_counter() - t0
    N = 32 << k
    print(f"{scene} N={N} in flight={depth}: {dt / steps * 1e3:.4f} ms per regen  {N ** 3 * steps / dt / 1e9:.1f} Gvox/s  tris={info.triangles}", flush=True)
    for h in hs:
        h.close()
