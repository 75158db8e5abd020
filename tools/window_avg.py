#!/usr/bin/env python3
"""Windowed per-launch averages of a kernel from a rocprofv3 kernel trace.  usage: window_avg.py DIR KERNEL_SUBSTRING [window]"""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True), key=os.path.getmtime)
win = int(sys.argv[3]) if len(sys.argv) > 3 else 25
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if sys.argv[2] in r['Kernel_Name']]
print(sys.argv[2], len(d), "launches; avg per window of", win, ":", " ".join("%.1f" % (sum(d[i:i + win]) / len(d[i:i + win])) for i in range(0, len(d), win)))
