#!/usr/bin/env python3
"""P1 dense sampling in steady state (bench.py's steady_ms) for the given scenes and grids, optionally over the values of a handle option:
python tools/p1_time.py sdobj,sphere 256,512,1024 [dense_occ=3+dense_nt=0 dense_occ=4+dense_nt=2 ...]  (every combination names all the options it varies)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sdmesh_pkg
import bench
pkg = sdmesh_pkg.load()
scenes = (sys.argv[1] if len(sys.argv) > 1 else "sdobj").split(",")
grids = [int(g) for g in (sys.argv[2] if len(sys.argv) > 2 else "256,512,1024").split(",")]
h = pkg.CudaHandler(0, "sphere")
combos = [[(kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split("+")] for a in sys.argv[3:]] or [[]]   # e.g. dense_occ=3+dense_nt=0
big = torch.empty((max(grids) + 1) ** 3, dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
for sc in scenes:
    h.set_scene(sc)
    for g in grids:
      for combo in combos:
        which = 0                                  # ("which=2": the write-ceiling kernel k_stream_fill instead of P1, "which=1": k_sign_bits)
        for name, val in combo:
            if name == "which": which = val
            else: h.set_option(name, val)
        ms, ws = bench.steady_ms(h, which, g, big.data_ptr(), 600 if g <= 256 else 50 if g <= 512 else 20)
        print(json.dumps({"scene": sc, "grid": g, "opt": "+".join(f"{k}={v}" for k, v in combo), "ms": round(ms, 5), "frac": round(4.0 * (g + 1) ** 3 / (ms * 1e-3) / 8e12, 4), "windows_ms": [round(w, 5) for w in ws]}), flush=True)
