#!/usr/bin/env python3
"""Sweep k_sample_dense launch parameters on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sdmesh_pkg
pkg = sdmesh_pkg.load()
h = pkg.CudaHandler(0, "sphere")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
buf = torch.empty((N + 1) ** 3, dtype=torch.float32, device="cuda:0")
b = 4.0 * (N + 1) ** 3
print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a", "affinity", len(os.sched_getaffinity(0)))
h.set_scene("sphere"); h.set_option("dense_rows", 0); h.set_option("dense_nt", 1)
for contig in (0, 1):
    h.set_option("dense_contig", contig)
    for grid in (2048, 4096, 8192, 12288, 16384, 20480, 24576, 32768, 49152, 65535):
        h.set_option("dense_grid", grid)
        ms = min(h.time_kernel(0, N, buf.data_ptr(), 30) for _ in range(3))
        print(f"sphere contig={contig} grid={grid:6d} ms={ms:.4f} GB/s={b / ms / 1e6:.0f} frac={b / ms / 1e6 / 8000:.3f}", flush=True)
h.set_option("dense_contig", 0)
for scene in ("sdobj",):
    h.set_scene(scene)
    for rows in (0, 1):
        h.set_option("dense_rows", rows)
        for nt in (1,):
            h.set_option("dense_nt", nt)
            for grid in (1024, 2048, 4096, 8192, 16384, 32768, 65535):
                h.set_option("dense_grid", grid)
                ms = min(h.time_kernel(0, N, buf.data_ptr(), 30) for _ in range(3))
                print(f"{scene} rows={rows} nt={nt} grid={grid:6d} ms={ms:.4f} GB/s={b / ms / 1e6:.0f} frac={b / ms / 1e6 / 8000:.3f}", flush=True)
