#!/usr/bin/env python3
"""Same-box A/B of two builds of libsdmesh.so (boxes of the pool differ by several %, so cross-call comparisons
mislead).  usage: ab_perf.py base.so [new.so] -- alternates the two libraries in child processes, 3 rounds."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = {"base": os.path.abspath(sys.argv[1]),
        "new": os.path.abspath(sys.argv[2]) if len(sys.argv) > 2 else os.path.join(ROOT, "bevy-signed-distance-mesh-generation_amd", "libsdmesh.so")}
CHILD = r'''
import os, sys, statistics
sys.path.insert(0, %r)
import sdmesh_pkg
pkg = sdmesh_pkg.load()
h = pkg.CudaHandler(0, "sdobj")
out = []
for k in (3, 4, 5):
    h.regen(k); h.regen(k)
    ms = [h.regen(k).device_ms for _ in range(80)]
    out.append("N=%%d med %%.4f min %%.4f" %% (32 << k, statistics.median(ms), min(ms)))
print(" | ".join(out))
''' % ROOT
for rnd in range(3):
    for name, path in libs.items():
        env = dict(os.environ, SDMESH_LIB=path)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
        print(f"{name:5s} {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
