#!/usr/bin/env python3
"""Wall-clock A/B of option settings (synchronous regen calls back to back; static scene and animated scene).
usage: wall_ab.py "opt=v,..." ..."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sdmesh_pkg
pkg = sdmesh_pkg.load()
h = pkg.CudaHandler(0, "sdobj")
ha = pkg.CudaHandler(0, "anim", [0.5])
for cy in (0.5, -0.5, 0.0):
    ha.set_scene("anim", [cy]); ha.regen(4)
for rnd in range(3):
    for cfg in sys.argv[1:]:
        for kv in cfg.split(","):
            o, v = kv.split("="); h.set_option(o, int(v)); ha.set_option(o, int(v))
        line = []
        for k in (3, 4):
            for _ in range(20): h.regen(k)
            t = time.perf_counter(); dev = []
            for _ in range(300): dev.append(h.regen(k).device_ms)
            w = (time.perf_counter() - t) / 300
            line.append(f"N={32 << k} wall {w * 1e3:.4f} dev {statistics.median(dev):.4f}")
        for f in range(20):
            ha.set_animation_time(f / 60.0); ha.regen(4)
        t = time.perf_counter()
        for f in range(240):
            ha.set_animation_time(f / 60.0); ha.regen(4)
        w = (time.perf_counter() - t) / 240
        line.append(f"anim512 wall {w * 1e3:.4f} ({1 / w:.0f} fps)")
        print(f"{cfg}: " + " | ".join(line), flush=True)
