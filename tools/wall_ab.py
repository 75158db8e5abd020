#!/usr/bin/env python3
"""Wall-clock A/B of option settings (synchronous regen calls back to back).  usage: wall_ab.py "opt=v,..." ..."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sdmesh_pkg
pkg = sdmesh_pkg.load()
h = pkg.CudaHandler(0, "sdobj")
for rnd in range(3):
    for k in (3, 4):
        for cfg in sys.argv[1:]:
            for kv in cfg.split(","):
                o, v = kv.split("="); h.set_option(o, int(v))
            for _ in range(20): h.regen(k)
            t = time.perf_counter(); dev = []
            for _ in range(300): dev.append(h.regen(k).device_ms)
            w = (time.perf_counter() - t) / 300
            print(f"N={32 << k} {cfg}: wall {w * 1e3:.4f} ms  device {statistics.median(dev):.4f} ms", flush=True)
