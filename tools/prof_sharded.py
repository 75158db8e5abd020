#!/usr/bin/env python3
"""Workload for rocprofv3 / wall timing: the one-mesh steps on a ONE-rank RCCL communicator (what a 1-GPU box can run of the
multi-GPU protocol).  usage: prof_sharded.py zslab|xslab|plain scene k reps [option=value ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sdmesh_pkg  # noqa: E402

form, scene, k, reps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
pkg = sdmesh_pkg.load()
h = pkg.CudaHandler(0, scene)
for a in sys.argv[5:]:
    name, val = a.split("=")
    h.set_option(name, int(val))
if form != "plain":
    h.comm_init(pkg.CudaHandler.comm_unique_id(), 1, 0)
step = {"zslab": lambda: h.sharded_regen(k, 0), "xslab": lambda: h.distributed_regen(k)[0], "plain": lambda: h.regen(k)}[form]
for _ in range(5):
    info = step()
t0 = time.perf_counter()
for _ in range(reps):
    info = step()
dt = (time.perf_counter() - t0) / reps
print(form, scene, k, info.triangles, info.vertices, "ms_per_step", round(dt * 1e3, 4), "route", {"zslab": h.sharded_route, "xslab": h.distributed_route, "plain": lambda: None}[form]())
