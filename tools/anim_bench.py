#!/usr/bin/env python3
"""BASELINE config 5: per-frame animated SDF (sphere centre (0, 0.5 sin t, 0), t = frame / 60), 512^3, 120 frames:
frames/s with the eager planned chain and with the hipGraph-captured chain."""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sdmesh_pkg
pkg = sdmesh_pkg.load()
h = pkg.CudaHandler(0, "anim", [0.0])
k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
# widest frame first so the capacity plan covers the whole animation
for cy in (0.5, -0.5, 0.0):
    h.set_scene("anim", [cy]); h.regen(k)
for graph in (0, 1):
    h.set_option("graph", graph)
    for _ in range(5):
        h.regen(k)
    t0 = time.perf_counter(); dev = []; tris = 0
    for f in range(120):
        h.set_animation_time(f / 60.0)
        info = h.regen(k)
        dev.append(info.device_ms); tris += info.triangles
    dt = time.perf_counter() - t0
    print(f"graph={graph} N={32 << k}: {120 / dt:.1f} frames/s  wall/frame={dt / 120 * 1e3:.3f} ms  device med={np.median(dev):.3f} ms  "
          f"Mtris/s={tris / dt / 1e6:.0f}", flush=True)
