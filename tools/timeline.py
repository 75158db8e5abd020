#!/usr/bin/env python3
"""Print the kernel timeline of the LAST regen in a rocprofv3 --kernel-trace CSV directory: start offset, duration and
the gap to the previous kernel's end (us).  usage: timeline.py DIR"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# last regen = from the last k_create_field on
idx = max(i for i, r in enumerate(rows) if 'k_create_field' in r['Kernel_Name'])
sel = rows[idx:]
t0 = int(sel[0]['Start_Timestamp']); prev_end = t0
print("%-44s %9s %8s %8s" % ("kernel", "start_us", "dur_us", "gap_us"))
for r in sel:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('void ', '').replace('sdm::', '')[:44]
    print("%-44s %9.2f %8.2f %8.2f" % (name, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = max(prev_end, e)
print("total %.2f us" % ((prev_end - t0) / 1e3))
