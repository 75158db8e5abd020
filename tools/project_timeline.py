#!/usr/bin/env python3
"""Per-wave timeline of the projection kernel (option "project_diag"): lifetime histogram, tail, per-SIMD load.
usage: project_timeline.py [scene] [levels] [out.json] [raw.npy]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sdmesh_pkg  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "sdobj"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
pkg = sdmesh_pkg.load()
h = pkg.CudaHandler(0, scene)
h.regen(k)
for _ in range(5):
    h.regen(k)
h.set_option("project_diag", 1)
res = []
for rep in range(3):
    h.regen(k)
    d = h.project_diag()
    if len(sys.argv) > 4 and rep == 2:
        np.save(sys.argv[4], d)                  # raw per-wave records (row = wave id) for offline analysis
    d = d[d[:, 1] != 0]
    t0 = d[:, 0].min()
    start, end, rounds = (d[:, 0] - t0).astype(np.float64), (d[:, 1] - t0).astype(np.float64), (d[:, 2] & 0x0FFFFFFF).astype(np.int64)
    hw = d[:, 2] >> 32
    simd = ((hw >> 4) & 3).astype(np.int64); cu = ((hw >> 8) & 15).astype(np.int64); sh = ((hw >> 12) & 1).astype(np.int64)
    se = ((hw >> 13) & 7).astype(np.int64); xcc = ((d[:, 2] >> 28) & 15).astype(np.int64)
    rt0, rt1 = (d[:, 3] & 0xFFFFFFFF).astype(np.int64), (d[:, 3] >> 32).astype(np.int64)
    span_us = max(float((rt1[xcc == x].max() - rt0[xcc == x].min()) % (1 << 32)) / 100.0 for x in np.unique(xcc))
    unit = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    life = end - start
    span = end.max()
    per_unit = np.bincount(np.unique(unit, return_inverse=True)[1], weights=rounds)
    q = lambda a, p: float(np.percentile(a, p))   # noqa: E731
    # shader clock the kernel ran at (s_memtime cycles of a wave over its s_memrealtime span, 100 MHz) and, per SIMD, when its
    # waves finished in the order they finished (the arbiter serves a SIMD's waves oldest first)
    dur = ((rt1 - rt0) % (1 << 32)).astype(np.float64)
    clock_ghz = float(np.median(life[dur > 500] / dur[dur > 500])) * 0.1
    t_end = ((rt1 - rt0.min()) % (1 << 32)).astype(np.float64) / 100.0
    order = []
    for uu in np.unique(unit):
        m = unit == uu
        if m.sum() == 5:
            o = np.argsort(t_end[m])
            order.append(np.concatenate([t_end[m][o], rounds[m][o] / 64.0]))
    order = np.array(order).mean(0) if order else np.zeros(10)
    res.append({"working_waves": int(len(d)), "kernel_span_us_by_s_memrealtime": span_us, "kernel_span_ticks": span, "ticks_are": "s_memtime = shader-clock cycles, one counter per XCD (not comparable across XCDs); the wall span is from s_memrealtime",
                "start_p50": q(start, 50), "start_max": float(start.max()),
                "end_p10": q(end, 10), "end_p50": q(end, 50), "end_p90": q(end, 90), "end_p99": q(end, 99), "end_max": float(end.max()),
                "life_mean": float(life.mean()), "life_p50": q(life, 50), "life_p99": q(life, 99), "life_max": float(life.max()),
                "rounds_per_wave_mean": float(rounds.mean()) / 64.0, "rounds_per_wave_hist(ceil(vertex_rounds/64))": np.bincount(((rounds + 63) // 64).astype(np.int64)).tolist(),
                "waves_alive_frac": float(life.sum() / (len(d) * span)),
                "simd_units_used": int(len(per_unit)), "vertex_rounds_per_simd_mean": float(per_unit.mean()), "vertex_rounds_per_simd_max": float(per_unit.max()),
                "end_hist_10_bins": np.histogram(end, bins=10, range=(0, span))[0].tolist(),
                "shader_clock_ghz": round(clock_ghz, 3),
                "simds_with_5_waves__mean_end_us_by_finishing_order": [round(float(v), 2) for v in order[:5]],
                "simds_with_5_waves__mean_rounds_per_vertex_slot_by_finishing_order": [round(float(v), 2) for v in order[5:]]})
print(json.dumps(res[-1], indent=1))
if len(sys.argv) > 3:
    json.dump(res, open(sys.argv[3], "w"), indent=1)
