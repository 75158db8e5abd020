#!/usr/bin/env python3
"""Phase stamps of k_tile_dedupe (option "project_diag" = 2): median / p90 cycles per phase over the tiles.
usage: tile_phases.py [scene] [levels]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sdmesh_pkg  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "sdobj"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
pkg = sdmesh_pkg.load()
h = pkg.CudaHandler(0, scene)
h.regen(k)
for _ in range(5):
    h.regen(k)
h.set_option("project_diag", 2)
h.regen(k)
d = h.project_diag().reshape(-1, 8)
d = d[d[:, 7] != 0].astype(np.int64)
names = ["loads+prefix+scan1+lattice", "scan2", "hash clear+sync", "LDS inserts+sync", "occupancy+scan3", "midpoints+stores+sync", "tri_e stores+sync"]
print("tiles", len(d), "total cycles per tile: median", int(np.median(d[:, 7] - d[:, 0])), "p90", int(np.percentile(d[:, 7] - d[:, 0], 90)), "max", int((d[:, 7] - d[:, 0]).max()))
for i, n in enumerate(names):
    dt = d[:, i + 1] - d[:, i]
    print(f"  {n:32s} median {int(np.median(dt)):7d}  p90 {int(np.percentile(dt, 90)):7d}")
