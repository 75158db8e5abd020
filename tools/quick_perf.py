#!/usr/bin/env python3
"""Quick device-time probe (GPU box): regen at several levels, direct vs shared-edge, dense pass."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import sdmesh_pkg  # noqa: E402

pkg = sdmesh_pkg.load()
h = pkg.CudaHandler(0)
for scene in ("sdobj", "sphere", "csg3"):
    h.set_scene(scene)
    for k in (3, 4, 5):
        h.regen(k)                     # capacity discovery (synchronous)
        ms, wall = [], []
        for _ in range(20):
            t = time.perf_counter()
            info = h.regen(k)
            wall.append((time.perf_counter() - t) * 1e3)
            ms.append(info.device_ms)
        N = 32 << k
        m = float(np.median(ms))
        print(f"{scene:6s} N={N:5d} vox={info.voxel_count[k]:8d} uniq={info.unique_edges:8d} tris={info.triangles:8d} "
              f"verts={info.vertices:8d} device_ms med={m:.3f} min={min(ms):.3f} wall_med={np.median(wall):.3f}  "
              f"Gvox/s={N**3 / m / 1e6:.1f} Mtris/s={info.triangles / m / 1e3:.1f}", flush=True)
h.set_scene("sdobj")
h.set_option("direct", 1)
for k in (3, 4):
    h.regen(k)
    ms = [h.regen(k).device_ms for _ in range(5)]
    print(f"sdobj direct(no edge sharing) N={32 << k} device_ms med={np.median(ms):.3f}", flush=True)
h.set_option("direct", 0)
for scene in ("sphere", "sdobj"):
    h.set_scene(scene)
    for N in (256, 512):
        buf = torch.empty((N + 1) ** 3, dtype=torch.float32, device="cuda:0")
        ms = h.time_kernel(0, N, buf.data_ptr(), 20)
        b = 4.0 * (N + 1) ** 3
        print(f"dense {scene:6s} N={N} ms={ms:.4f} GB/s={b / ms / 1e6:.0f} frac_of_8TB/s={b / ms / 1e6 / 8000:.3f}", flush=True)
