#!/usr/bin/env python3
"""A/B of sdmesh_download_mesh on one / two streams, and the bench's host_api leg alone.  usage: host_api_ab.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sdmesh_pkg, bench  # noqa: E402
pkg = sdmesh_pkg.load()
for streams in (1, 2, 1, 2):
    orig = pkg.CudaHandler.__init__
    def init(self, *a, _o=orig, **k):
        _o(self, *a, **k); self.set_option("download_streams", streams)
    pkg.CudaHandler.__init__ = init
    r = bench.host_api_leg(pkg, 0, "sdobj", None, levels_list=(4,), reps=10)["512^3"]
    pkg.CudaHandler.__init__ = orig
    print("download_streams", streams, json.dumps({k: r[k] for k in ("refine_voxel_field_ms", "voxel_field_to_mesh_ms", "download_mesh_ms", "mesh_plus_download_ms", "whole_sequence_ms")}))
