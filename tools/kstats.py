#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats CSV directory.  usage: kstats.py DIR [regens]"""
import csv, glob, sys
d = sys.argv[1]; regens = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
import os
f = max(glob.glob(d + '/**/*_kernel_stats.csv', recursive=True), key=os.path.getmtime)   # newest: gpurun merges, never deletes
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("%-58s %6s %10s %7s" % ("kernel", "calls", "avg_us", "tot%"))
for r in rows:
    print("%-58s %6s %10.2f %7.1f" % (r['Name'][:58], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
print("sum of kernel time per regen: %.1f us" % (tot / regens / 1e3))
