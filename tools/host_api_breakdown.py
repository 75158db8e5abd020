#!/usr/bin/env python3
"""Where the PCIe-inclusive host API spends its time at 512^3: the C call sdmesh_voxel_field_to_mesh alone (upload + chain + counts),
sdmesh_download_mesh alone, and the Python ObjData assembly on top."""
import ctypes, os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sdmesh_pkg
pkg = sdmesh_pkg.load()
from bevy_signed_distance_mesh_generation_amd import handler as H
h = pkg.CudaHandler(0, "sdobj")
f = pkg.CudaHandler.create_cuda_voxel_field()
for _ in range(4):
    h.refine_voxel_field(f)
vox = np.ascontiguousarray(f.voxels, np.float32).reshape(-1, 3)
lib, hh = h._lib, h._h
tc, td, tp = [], [], []
for rep in range(8):
    nv, nt = ctypes.c_size_t(0), ctypes.c_size_t(0)
    t0 = time.perf_counter()
    rc = lib.sdmesh_voxel_field_to_mesh(hh, vox.ctypes.data, vox.shape[0], H._pt(f.voxel_size), None, 0, ctypes.byref(nv), None, 0, ctypes.byref(nt))
    t1 = time.perf_counter()
    assert rc == 0
    verts = np.empty((nv.value, 2, 3), np.float32); idx = np.empty((nt.value, 3), np.uint32)
    t2 = time.perf_counter()
    lib.sdmesh_download_mesh(hh, verts.ctypes.data, idx.ctypes.data)
    t3 = time.perf_counter()
    obj = H.ObjData(verts[:, 0].copy(), verts[:, 1].copy(), idx.copy())
    t4 = time.perf_counter()
    tc.append(t1 - t0); td.append(t3 - t2); tp.append((t2 - t1) + (t4 - t3))
m = lambda v: statistics.median(v) * 1e3
print(f"voxels {vox.shape[0]} ({vox.nbytes / 1e6:.1f} MB up), {nv.value} vertices + {nt.value} triangles ({(verts.nbytes + idx.nbytes) / 1e6:.1f} MB down)")
print(f"C call voxel_field_to_mesh (upload + chain + sync): {m(tc):.3f} ms; sdmesh_download_mesh: {m(td):.3f} ms; numpy alloc + ObjData copies: {m(tp):.3f} ms")
