#!/usr/bin/env python3
"""From the kernel trace of `bench.py` under rocprofv3: per-launch durations of the three streaming kernels, averaged over
all launches, over the first 20 (clocks still settling) and over the last 50 (= the window bench.py times with HIP events).
usage: streaming_window.py TRACE_DIR OUT.csv"""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
w = csv.writer(open(sys.argv[2], 'w', newline=''))
w.writerow(["kernel", "launches", "avg_us_all_launches", "avg_us_first_20", "avg_us_last_50_(the_window_bench.py_times)"])
for key in ("k_sample_dense_q1<1, true>", "k_sample_dense<1, true>", "k_sample_dense_rows_aligned<0", "k_sample_dense_rows<0", "k_stream_fill", "k_sign_bits"):
    sel = [r for r in rows if key in r['Kernel_Name']]
    if not sel:
        continue
    # bench.py also launches P1 over a volume of twice the grid: keep the launches of the most frequent grid size (the 512^3 ones)
    sizes = {}
    for r in sel:
        sizes[r['Grid_Size_X']] = sizes.get(r['Grid_Size_X'], 0) + 1
    main = max(sizes, key=sizes.get)
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in sel if r['Grid_Size_X'] == main]
    if len(d) >= 50:
        w.writerow([key, len(d), "%.2f" % (sum(d) / len(d)), "%.2f" % (sum(d[:20]) / 20), "%.2f" % (sum(d[-50:]) / 50)])
        print(key, len(d), "all %.2f first20 %.2f last50 %.2f" % (sum(d) / len(d), sum(d[:20]) / 20, sum(d[-50:]) / 50))
    for g, n in sizes.items():
        if g != main and n >= 10:
            e = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in sel if r['Grid_Size_X'] == g]
            w.writerow([key + " (grid %s threads: the volume of twice the grid)" % g, len(e), "%.2f" % (sum(e) / len(e)), "", "%.2f" % (sum(e[-20:]) / 20)])
            print(key, "grid", g, len(e), "all %.2f last20 %.2f" % (sum(e) / len(e), sum(e[-20:]) / 20))
