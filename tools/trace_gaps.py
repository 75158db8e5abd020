#!/usr/bin/env python3
"""From a rocprofv3 kernel trace: per regen the sum of kernel durations and of the gaps between consecutive kernels (the chain's idle time),
median over the regens.  A regen = the run of kernels from one k_col_refine to the next.  usage: trace_gaps.py <dir with *_kernel_trace.csv>"""
import csv, glob, statistics, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
chains, cur = [], []
for r in rows:
    if "k_col_refine" in r[2] and cur:
        chains.append(cur); cur = []
    cur.append(r)
chains = [c for c in chains[3:] if len(c) >= 8]
busy = [sum(e - s for s, e, _ in c) / 1e3 for c in chains]
span = [(c[-1][1] - c[0][0]) / 1e3 for c in chains]
gaps = [sp - b for sp, b in zip(span, busy)]
print(f"{len(chains)} regens, kernels per regen {statistics.median(len(c) for c in chains)}: busy {statistics.median(busy):.1f} us, "
      f"first start -> last end {statistics.median(span):.1f} us, gaps {statistics.median(gaps):.1f} us")
names = {}
for c in chains:
    for i, (s, e, n) in enumerate(c):
        key = n.split("(")[0][-28:]
        names.setdefault((i, key), []).append(((e - s) / 1e3, (s - c[i - 1][1]) / 1e3 if i else 0.0))
for (i, key), v in sorted(names.items()):
    if len(v) > len(chains) // 2:
        print(f"  {i:2d} {key:28s} dur {statistics.median(x[0] for x in v):7.2f} us   gap before {statistics.median(x[1] for x in v):6.2f} us")
