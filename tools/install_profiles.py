#!/usr/bin/env python3
"""Copy the summaries tools/capture_profiles.sh left in gpurun_out/prof/ into profiles/ (round tag as argv[1])."""
import collections, csv, glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
P = "gpurun_out/prof"
one = lambda pat: max(glob.glob(pat, recursive=True), key=os.path.getmtime)   # noqa: E731 -- newest (gpurun merges, never deletes)
shutil.copy(one(f"{P}/regen_trace/**/*_kernel_stats.csv"), f"profiles/{tag}_regen_sdobj512_kernel_stats.csv")
shutil.copy(one(f"{P}/bench_trace/**/*_kernel_stats.csv"), f"profiles/{tag}_bench_kernel_stats.csv")
shutil.copy(one(f"{P}/dense_write/**/*_counter_collection.csv"), f"profiles/{tag}_dense_pmc_write_size.csv")
shutil.copy(one(f"{P}/dense_fetch/**/*_counter_collection.csv"), f"profiles/{tag}_dense_pmc_fetch_size.csv")
open(f"profiles/{tag}_bench_under_rocprof.json", "w").write(
    "".join(l for l in open(f"{P}/bench_trace.stdout") if l.startswith("{")))
shutil.copy(f"{P}/bench_plain.json", f"profiles/{tag}_bench_plain.json")
if os.path.exists(f"{P}/streaming_window.csv"):
    shutil.copy(f"{P}/streaming_window.csv", f"profiles/{tag}_bench_streaming_kernels_timed_window.csv")
rows = list(csv.DictReader(open(one(f"{P}/regen_pmc_sq/**/*_counter_collection.csv"))))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = ["SQ_INSTS_VALU", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE"]
with open(f"profiles/{tag}_regen_sdobj512_pmc_sq.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launches"] + cols + ["valu_issue_frac"])
    for k, c in agg.items():
        n = len(next(iter(c.values())))
        m = {x: (sum(c[x]) / len(c[x]) if x in c else 0.0) for x in cols}
        # SQ_ACTIVE_INST_VALU counts quad-cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs
        frac = 4.0 * m["SQ_ACTIVE_INST_VALU"] / (m["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0) if m["GRBM_GUI_ACTIVE"] else 0.0
        w.writerow([k, n] + ["%.0f" % m[x] for x in cols] + ["%.3f" % frac])
        if "project" in k or "flip" in k: print(k, n, {x: "%.3g" % m[x] for x in cols}, "valu_issue_frac %.3f" % frac)
for name in ("bench_plain.json", ):
    j = json.loads(open(f"{P}/{name}").read())
    print(name, j["value"], j["ms_per_step"], j["device_ms_per_step"], j["roofline"]["frac"], j["roofline"].get("write_ceiling"), j["cpu_baseline"]["value"])
