#!/usr/bin/env python3
"""Regens in flight on one GPU (one handle each, sdmesh_regen_begin / _end): ms per regen by depth."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sdmesh_pkg
pkg = sdmesh_pkg.load()
levels = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for depth in (1, 2, 3, 4, 6, 8, 12):
    hs = [pkg.CudaHandler(0, "sdobj") for _ in range(depth)]
    for h in hs:
        h.regen(levels)
    n = 600
    best = None
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(min(depth, n)):
            hs[i % depth].regen_begin(levels)
        for i in range(n):
            h = hs[i % depth]
            h.regen_end()
            if i + depth < n:
                h.regen_begin(levels)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        best = dt if best is None else min(best, dt)
    print(f"{32 << levels}^3 depth {depth}: {best * 1e3:.4f} ms per regen", flush=True)
    for h in hs:
        h.close()
