import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
import sdmesh_pkg
pkg = sdmesh_pkg.load()
h = pkg.CudaHandler(0, "sphere")
h.set_option("dense_rows", 0)
for N in (511, 512, 255, 127):
    S = N + 1
    for q1 in (0, 1):
        h.set_option("dense_q1", q1)
        buf = torch.full((S ** 3,), float("nan"), dtype=torch.float32, device="cuda:0")
        h.sample_dense(N, buf.data_ptr())
        a = buf.cpu().numpy()
        nan = np.isnan(a)
        print("N", N, "q1", q1, "nan count", int(nan.sum()), "first nan", int(np.argmax(nan)) if nan.any() else -1, "total", a.size)
