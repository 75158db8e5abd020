#!/usr/bin/env python3
"""Workload for the PMC passes: 5 launches of the P1 dense sample kernel per scene at N = 512."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sdmesh_pkg
pkg = sdmesh_pkg.load()
h = pkg.CudaHandler(0, "sphere")
N = 512
buf = torch.empty((N + 1) ** 3, dtype=torch.float32, device="cuda:0")
for scene in ("sphere", "sdobj"):
    h.set_scene(scene)
    for _ in range(5):
        ms = h.sample_dense(N, buf.data_ptr())
    print(scene, "last launch ms", ms)
