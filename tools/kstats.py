#!/usr/bin/env python3
"""Per-kernel average / minimum duration from a rocprofv3 --stats directory.  usage: kstats.py DIR [substring ...]"""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True), key=os.path.getmtime)
pats = sys.argv[2:]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "selfcheck" in n or (pats and not any(p in n for p in pats)):
        continue
    print("%-34s calls %5s avg %9.2f us  min %9.2f" % (n.split("(")[0].replace("void ", "").replace("sdm::", "")[:34], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
