#!/usr/bin/env python3
"""Experiment: launch-shape knobs (tile-kernel grid, k_flip grid multiplier)."""
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
    for tg in (1024, 2048, 4096, 1024):
        h.set_option("tgrid", tg); run(k, f"tgrid={tg} flip_mult=1")
    h.set_option("tgrid", 2048)
    for fm in (2, 3, 1):
        h.set_option("flip_mult", fm); run(k, f"tgrid=2048 flip_mult={fm}")
