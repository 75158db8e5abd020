#!/usr/bin/env python3
"""PCIe-inclusive time of the reference-shaped host API (host buffers in and out at every call, src/cuda/mod.rs:124,204):
create_cuda_voxel_field -> k x refine_voxel_field -> voxel_field_to_mesh, against the device-resident regen."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sdmesh_pkg
pkg = sdmesh_pkg.load()
h = pkg.CudaHandler(0, "sdobj")
h.set_option("timing", 1)
for k in (3, 4):
    ts, tm = [], []
    for rep in range(6):
        t0 = time.perf_counter()
        f = pkg.CudaHandler.create_cuda_voxel_field()
        for _ in range(k):
            h.refine_voxel_field(f)
        t1 = time.perf_counter()
        obj = h.voxel_field_to_mesh(f)
        t2 = time.perf_counter()
        ts.append(t1 - t0); tm.append(t2 - t1)
    r = [h.regen(k).device_ms for _ in range(20)]
    print(f"N={32 << k}: host API refine x{k} {statistics.median(ts) * 1e3:.2f} ms + voxel_field_to_mesh {statistics.median(tm) * 1e3:.2f} ms "
          f"(voxels {len(f.voxels)}, tris {len(obj.polys) if hasattr(obj, 'polys') else '?'}) vs device-resident regen {statistics.median(r):.3f} ms", flush=True)
