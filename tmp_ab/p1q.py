import os, sys, torch
sys.path.insert(0, os.getcwd())
import sdmesh_pkg
pkg = sdmesh_pkg.load()
h = pkg.CudaHandler(0, "sphere")
n = 513 ** 3
buf = torch.empty(n, dtype=torch.float32, device="cuda")
h.time_kernel(0, 512, buf.data_ptr(), 250)
for rep in range(3):
    for q1 in (1, 0):
        h.set_option("dense_q1", q1)
        h.time_kernel(0, 512, buf.data_ptr(), 50)
        ms = h.time_kernel(0, 512, buf.data_ptr(), 100)
        print("sphere q1", q1, "ms", round(ms, 5), "GB/s", round(4 * n / ms / 1e6, 1), flush=True)
