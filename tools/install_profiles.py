#!/usr/bin/env python3
"""Copy the summaries tools/capture_profiles.sh left in gpurun_out/prof/ into profiles/ (round tag as argv[1]) and write
profiles/constants.json: the measured constants bench.py cannot collect from inside its own process (PMC passes)."""
import collections, csv, glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
P = "gpurun_out/prof"
one = lambda pat: max(glob.glob(pat, recursive=True), key=os.path.getmtime)   # noqa: E731 -- newest (gpurun merges, never deletes)
shutil.copy(one(f"{P}/regen_trace/**/*_kernel_stats.csv"), f"profiles/{tag}_regen_sdobj512_kernel_stats.csv")
shutil.copy(one(f"{P}/bench_trace/**/*_kernel_stats.csv"), f"profiles/{tag}_bench_kernel_stats.csv")
shutil.copy(one(f"{P}/dense_write/**/*_counter_collection.csv"), f"profiles/{tag}_dense_pmc_write_size.csv")
shutil.copy(one(f"{P}/dense_fetch/**/*_counter_collection.csv"), f"profiles/{tag}_dense_pmc_fetch_size.csv")
open(f"profiles/{tag}_bench_under_rocprof.json", "w").write(
    "".join(l for l in open(f"{P}/bench_trace.stdout") if l.startswith("{")))
shutil.copy(f"{P}/bench_plain.json", f"profiles/{tag}_bench_plain.json")
for k in (4, 5):                                                    # round 4: the one-mesh steps on a one-rank communicator
    for form in ("zslab", "xslab"):
        g = glob.glob(f"{P}/{form}_trace_{k}/**/*_kernel_stats.csv", recursive=True)
        if g:
            shutil.copy(max(g, key=os.path.getmtime), f"profiles/{tag}_{form}_{32 << k}_one_rank_kernel_stats.csv")
if os.path.exists(f"{P}/one_mesh_one_rank.txt"):
    shutil.copy(f"{P}/one_mesh_one_rank.txt", f"profiles/{tag}_one_mesh_one_rank_wall.txt")
for src, dst in (("streaming_window.csv", f"{tag}_bench_streaming_kernels_timed_window.csv"), ("valu_rate.jsonl", f"{tag}_ubench_valu_rate.jsonl"),
                 ("valu_issue.jsonl", f"{tag}_ubench_valu_issue.jsonl"),
                 ("project_timeline.json", f"{tag}_project1_wave_timeline.json"), ("tile_phases.txt", f"{tag}_tile_dedupe_phases.txt")):
    if os.path.exists(f"{P}/{src}"):
        shutil.copy(f"{P}/{src}", f"profiles/{dst}")


def per_kernel(path, counters):
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for r in rows:
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = {"vgpr": int(r.get("VGPR_Count", 0) or 0), "sgpr": int(r.get("SGPR_Count", 0) or 0), "lds": int(r.get("LDS_Block_Size", 0) or 0)}
    return agg, meta


# ---- SQ pass: per-kernel averages over the launches
agg, meta = per_kernel(one(f"{P}/regen_pmc_sq/**/*_counter_collection.csv"), None)
cols = ["SQ_INSTS_VALU", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE"]
sq = {}
with open(f"profiles/{tag}_regen_sdobj512_pmc_sq.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launches", "vgpr"] + cols)
    for k, c in agg.items():
        n = len(next(iter(c.values())))
        m = {x: (sum(c[x]) / len(c[x]) if x in c else 0.0) for x in cols}
        sq[k] = m
        w.writerow([k[:60], n, meta[k]["vgpr"]] + ["%.0f" % m[x] for x in cols])

# ---- HBM traffic of the regen, per kernel (last regen of the pass), FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md, HBM)
traffic = {}
for name, ctr in (("regen_fetch", "FETCH_SIZE"), ("regen_write", "WRITE_SIZE")):
    a, _ = per_kernel(one(f"{P}/{name}/**/*_counter_collection.csv"), None)
    for k, c in a.items():
        traffic.setdefault(k, {})[ctr] = c[ctr][-1] * 1024.0          # the counters report KiB
with open(f"profiles/{tag}_regen_sdobj512_pmc_hbm.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "FETCH_SIZE_bytes_as_reported", "FETCH_bytes_x2_gfx950", "WRITE_SIZE_bytes"])
    tf = tw = 0.0
    for k, c in traffic.items():
        fs, ws = c.get("FETCH_SIZE", 0.0), c.get("WRITE_SIZE", 0.0)
        if "rocclr" in k:
            continue
        w.writerow([k[:60], "%.0f" % fs, "%.0f" % (2 * fs), "%.0f" % ws])
        tf += 2 * fs; tw += ws
    w.writerow(["TOTAL (kernels of one regen)", "", "%.0f" % tf, "%.0f" % tw])

# ---- kernel-trace averages of the planned chain
trace = {r["Name"]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(f"profiles/{tag}_regen_sdobj512_kernel_stats.csv")) if int(r["Calls"]) >= 10}


def find(d, key):
    for k, v in d.items():
        if key in k:
            return k, v
    return None, None


# registers / occupancy as the compiler reports them (tools/kernel_resources.py; the trace's VGPR_Count column is not the
# allocation on gfx950's unified file)
res = {}
rpath = f"profiles/{tag}_kernel_resource_usage.csv"
if os.path.exists(rpath):
    for r in csv.DictReader(open(rpath)):
        res[r["kernel"]] = r
step = {}
for stage, key in (("project1", "k_project1<0>"), ("flip_claim1", "k_flip_claim<0, 1>"), ("tile_first_flip", "k_tile_first<0>")):
    k, m = find(sq, key)
    tk, us = find(trace, key)
    if m is None:
        continue
    rr = res.get("sdm::" + key, {})
    step[stage] = {"kernel": k[:80], "wave_instr": round(m["SQ_INSTS_VALU"]), "avg_us": round(us, 2) if us else None,
                   "vgpr": int(rr["VGPRs"]) if rr else None, "sgpr": int(rr["SGPRs"]) if rr else None,
                   "waves_per_simd": int(rr["occupancy_waves_per_SIMD"]) if rr else None, "waves_launched": round(m["SQ_WAVES"]),
                   "wave_lifetime_quad_cycles": round(m["SQ_WAVE_CYCLES"])}

# ---- P1 traffic (dense sample at N = 512): per scene, WRITE_SIZE + 2 x FETCH_SIZE of one launch
dense = {}
for name, ctr in (("dense_write", "WRITE_SIZE"), ("dense_fetch", "FETCH_SIZE")):
    a, _ = per_kernel(one(f"{P}/{name}/**/*_counter_collection.csv"), None)
    for k, c in a.items():
        if "k_sample_dense" in k:
            dense.setdefault(k, {})[ctr] = c[ctr][-1] * 1024.0
p1 = {}
for k, c in dense.items():
    scene = "sphere" if "<1," in k else ("sdobj" if "<0," in k else None)
    if scene:
        p1[scene] = round(c.get("WRITE_SIZE", 0.0) + 2.0 * c.get("FETCH_SIZE", 0.0))

ub = [json.loads(l) for l in open(f"{P}/valu_rate.jsonl") if l.startswith("{")] if os.path.exists(f"{P}/valu_rate.jsonl") else []
ceiling = max((u["wave_instr_per_s"] for u in ub), default=None)
tl_path = f"profiles/{tag}_project1_wave_timeline.json"
clock = None
if os.path.exists(tl_path):
    clock = json.load(open(tl_path))[-1].get("shader_clock_ghz")
const = {"_what": "measured constants for bench.py, written by tools/install_profiles.py from the rocprofv3 passes committed beside this file",
         "round": tag,
         "p1_traffic_bytes": {"512": p1},
         "step_kernels": {"sdobj_512": step},
         "shader_clock_ghz_under_valu_load": clock or 2.2,
         "valu_ubench_wave_instr_per_s": ceiling,
         "valu_ubench_note": "valu_rate.hip counts SOURCE operations: hipcc packs its mul/add chains into v_pk_* (two per machine instruction), so this is "
                             "not a machine-instruction ceiling; see issue_cost (tools/ubench/valu_issue.hip, tools/isa_mix.py)",
         "valu_ubench_rows": ub,
         "regen_hbm_bytes": {"fetch_x2": round(tf), "write": round(tw)}}
json.dump(const, open("profiles/constants.json", "w"), indent=1)
# issue-cost bound of the VALU-bound kernels from their instruction mix (needs constants.json; adds to it)
if os.path.exists(f"profiles/{tag}_ubench_valu_issue.jsonl"):
    import subprocess
    subprocess.run([sys.executable, "tools/isa_mix.py", tag], check=True, stdout=subprocess.DEVNULL)
    ib = json.load(open(f"profiles/{tag}_issue_bound.json"))
    const["issue_cost"] = {"what": "mean issue cost of one VALU machine instruction of the kernel's hot block, from the per-kind costs of "
                                   "tools/ubench/valu_issue.hip at 5 waves per SIMD (tools/isa_mix.py)",
                           "cycles_per_instr_by_kind": ib["cost_cycles_per_instr_per_simd"],
                           "kernels": {k: {"mean_cycles_per_instr": v["mean_cycles_per_instr"], "by_class": v["by_class"]} for k, v in ib["kernels"].items()}}
    json.dump(const, open("profiles/constants.json", "w"), indent=1)
print(json.dumps(const["step_kernels"], indent=1)); print("p1", p1, "ceiling", ceiling, "regen traffic", tf / 1e6, tw / 1e6)
j = json.loads(open(f"{P}/bench_plain.json").read())
print("bench_plain", j["value"], j["ms_per_step"], j["device_ms_per_step"], j["roofline"]["frac"], j["cpu_baseline"]["value"])
