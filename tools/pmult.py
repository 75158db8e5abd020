#!/usr/bin/env python3
"""Experiment: two refine levels per step vs one."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sdmesh_pkg
pkg = sdmesh_pkg.load()
h = pkg.CudaHandler(0, "sdobj")
def run(k, tag):
    h.regen(k); h.regen(k)
    ms = [h.regen(k).device_ms for _ in range(30)]
    print(f"N={32 << k} {tag}: device_ms med={np.median(ms):.4f} min={min(ms):.4f}", flush=True)
for k in (3, 4, 5):
    for mc in (32, 16, 8, 32):
        h.set_option("mask_chunk", mc); run(k, f"mask_chunk={mc}")
