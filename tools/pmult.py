#!/usr/bin/env python3
"""Experiment: grid multiplier of k_project (more, shorter waves)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sdmesh_pkg
pkg = sdmesh_pkg.load()
h = pkg.CudaHandler(0, "sdobj")
for k in (4, 5):
    for mult in (1, 2, 3, 4, 1):
        h.set_option("flip_mult", mult)
        h.regen(k); h.regen(k)
        ms = [h.regen(k).device_ms for _ in range(30)]
        print(f"N={32 << k} flip_mult={mult} (project_mult=3): device_ms med={np.median(ms):.4f} min={min(ms):.4f}", flush=True)
