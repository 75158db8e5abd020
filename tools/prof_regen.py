#!/usr/bin/env python3
"""Workload for rocprofv3: N device-resident regens of one scene/level.  usage: prof_regen.py scene k reps [graph]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sdmesh_pkg  # noqa: E402

scene, k, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
pkg = sdmesh_pkg.load()
h = pkg.CudaHandler(0, scene)
h.set_option("timing", 1)
h.regen(k)
if len(sys.argv) > 4:
    h.set_option("graph", int(sys.argv[4]))
    h.regen(k)
for _ in range(reps):
    info = h.regen(k)
print(scene, k, info.triangles, info.vertices, info.device_ms)
