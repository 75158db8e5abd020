#!/usr/bin/env python3
"""P1 (dense sample, sphere + workload scene) and the write ceiling at N = 512 by HIP events: first 50 of a burst, then steady state."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sdmesh_pkg
pkg = sdmesh_pkg.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
h = pkg.CudaHandler(0, "sphere")
buf = torch.empty((N + 1) ** 3, dtype=torch.float32, device="cuda:0")
B = 4.0 * (N + 1) ** 3
for scene in ("sphere", "sdobj"):
    h.set_scene(scene)
    h.time_kernel(0, N, buf.data_ptr(), 20)
    first = h.time_kernel(0, N, buf.data_ptr(), 50)
    h.time_kernel(0, N, buf.data_ptr(), 200)
    ms = [h.time_kernel(0, N, buf.data_ptr(), 50) for _ in range(3)]
    print(scene, "first50 %.5f ms" % first, "steady", " ".join("%.5f" % m for m in ms), "ms  -> %.1f GB/s = %.3f of 8 TB/s" % (B / min(ms) / 1e6, B / min(ms) / 1e6 / 8000))
ms = h.time_kernel(2, N, buf.data_ptr(), 50)
print("stream_fill %.5f ms -> %.1f GB/s" % (ms, B / ms / 1e6))
