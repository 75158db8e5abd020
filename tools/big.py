#!/usr/bin/env python3
"""Largest grids: device-resident regen at 2048^3 and 4096^3 (O(surface) memory).  usage: big.py [scene]"""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sdmesh_pkg
pkg = sdmesh_pkg.load()
scene = sys.argv[1] if len(sys.argv) > 1 else "sdobj"
h = pkg.CudaHandler(0, scene)
for k in (6, 7):
    t = time.time(); info = h.regen(k); t1 = time.time() - t
    ms = [h.regen(k).device_ms for _ in range(5)]
    free, total = torch.cuda.mem_get_info()
    N = 32 << k
    print(f"{scene} N={N}: voxels={info.voxel_count[k]} tris={info.triangles} verts={info.vertices} first={t1:.2f}s "
          f"device_ms med={statistics.median(ms):.3f} Gvox/s={N**3 / statistics.median(ms) / 1e6:.0f} "
          f"Mtris/s={info.triangles / statistics.median(ms) / 1e3:.0f} status={list(info.status)} hbm_used={(total - free) / 2**30:.1f} GiB", flush=True)
