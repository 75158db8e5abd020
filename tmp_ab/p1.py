import os, sys, torch
sys.path.insert(0, os.getcwd())
import sdmesh_pkg
pkg = sdmesh_pkg.load()
h = pkg.CudaHandler(0, "sdobj")
buf = torch.empty(513 ** 3, dtype=torch.float32, device="cuda")
for sc in ("sdobj", "sphere", "anim"):
    h.set_scene(sc, [0.25] if sc == "anim" else None)
    h.time_kernel(0, 512, buf.data_ptr(), 250)
    ms = h.time_kernel(0, 512, buf.data_ptr(), 50)
    print(sc, "P1 ms", round(ms, 5), "GB/s", round(4 * 513 ** 3 / ms / 1e6, 1))
