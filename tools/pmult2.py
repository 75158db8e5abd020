#!/usr/bin/env python3
"""A/B of option settings on the device-resident regen.  usage: pmult2.py "opt=v,opt=v" "opt=v,..." ... [--scene s]"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sdmesh_pkg
pkg = sdmesh_pkg.load()
args = [a for a in sys.argv[1:] if not a.startswith("--")]
scene = "sdobj"
for a in sys.argv[1:]:
    if a.startswith("--scene="): scene = a.split("=")[1]
h = pkg.CudaHandler(0, scene)
for rnd in range(2):
    for k in (3, 4, 5):
        for cfg in args:
            for kv in cfg.split(","):
                o, v = kv.split("="); h.set_option(o, int(v))
            h.regen(k); h.regen(k)
            ms = [h.regen(k).device_ms for _ in range(60)]
            print(f"N={32 << k} {cfg}: med {statistics.median(ms):.4f} min {min(ms):.4f}", flush=True)
