#!/usr/bin/env python3
"""Workload for a kernel trace: 300 back-to-back launches of the P1 sphere kernel and of the pure write stream at N = 512
(tools/window_avg.py then shows how the per-launch time develops over a long burst)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sdmesh_pkg
pkg = sdmesh_pkg.load()
h = pkg.CudaHandler(0, "sphere")
buf = torch.empty(513 ** 3, dtype=torch.float32, device="cuda:0")
print("P1", h.time_kernel(0, 512, buf.data_ptr(), 300))
torch.cuda.synchronize()
print("fill", h.time_kernel(2, 512, buf.data_ptr(), 300))
