#!/usr/bin/env python3
"""From the kernel trace of `bench.py` under rocprofv3: per-launch durations of the streaming kernels, per launch grid (bench.py runs P1 at
256^3, 512^3 and 1024^3): average over all launches, over the first 20 (clocks still settling) and over the last 3 x `reps` launches = the
three windows bench.py times with HIP events (reps = 600 / 50 / 20 at the three grids, 50 for the other kernels).
usage: streaming_window.py TRACE_DIR OUT.csv"""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
w = csv.writer(open(sys.argv[2], 'w', newline=''))
w.writerow(["kernel", "grid_threads", "launches", "avg_us_all_launches", "avg_us_first_20", "timed_launches", "avg_us_timed_windows_(what_bench.py_times)"])
for key in ("k_sample_dense_q1<1, true>", "k_sample_dense<1, true>", "k_sample_dense_rows_aligned<0", "k_sample_dense_rows<0", "k_stream_fill", "k_sign_bits"):
    sel = [r for r in rows if key in r['Kernel_Name']]
    sizes = {}
    for r in sel:
        sizes.setdefault(int(r['Grid_Size_X']), []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    order = sorted(sizes)
    for g in order:
        d = sizes[g]
        if len(d) < 60:
            continue
        # the three grids of P1 in ascending size: 600 / 50 / 20 launches per window; one grid only (fill, sign bits): 50
        reps = {0: 600, 1: 50, 2: 20}[order.index(g)] if len(order) == 3 else 50
        timed = min(len(d), 3 * reps)
        w.writerow([key, g, len(d), "%.2f" % (sum(d) / len(d)), "%.2f" % (sum(d[:20]) / 20), timed, "%.2f" % (sum(d[-timed:]) / timed)])
        print(key, "grid", g, len(d), "all %.2f first20 %.2f timed(%d) %.2f" % (sum(d) / len(d), sum(d[:20]) / 20, timed, sum(d[-timed:]) / timed))
