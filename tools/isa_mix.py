#!/usr/bin/env python3
"""Issue-cost bound of a VALU-bound kernel from its own instruction mix.

tools/ubench/valu_issue.hip measures what one MACHINE instruction of each kind costs a gfx950 SIMD when it has enough waves to
pick from (cycles per instruction per SIMD; profiles/<tag>_ubench_valu_issue.jsonl).  This tool compiles kernels.hip to assembly
(cross-compiles, no GPU), takes the hottest basic block of the named kernels (the Newton round of k_project1, the orientation test of
k_tile_first), classifies its VALU instructions by those kinds and adds the costs up:
    bound = SQ_INSTS_VALU (PMC, profiles/constants.json) x mean cost per instruction of the block / (1024 SIMDs x clock)
= the time the launch would take if every SIMD issued back to back from start to end.  Classes overlap a little in hardware (the
ubench's mixed stream runs ~10 % faster than the sum of its parts), so the sum is a slightly generous bound.
usage: isa_mix.py [tag] -> profiles/<tag>_issue_bound.json"""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
WAVES = 5                                     # waves per SIMD the costs are taken at (k_project1 runs at 4.9)
KERNELS = {"project1": "_ZN3sdm10k_project1ILi0EE", "tile_first_flip": "_ZN3sdm12k_tile_firstILi0EE"}

# instruction -> ubench kind
FAST = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32",
        "v_xor_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_add_co_u32", "v_addc_co_u32", "v_not_b32", "v_bfe_u32", "v_lshl_add_u32",
        "v_add3_u32", "v_lshl_or_b32", "v_and_or_b32", "v_accvgpr", "v_mbcnt", "v_readfirstlane", "v_readlane", "v_writelane", "v_cvt")
HALF = ("v_min_f32", "v_max_f32", "v_med3_f32", "v_min3_f32", "v_max3_f32", "v_fma_f32", "v_fmac_f32", "v_mad", "v_pk_", "v_div_scale", "v_div_fmas",
        "v_div_fixup", "v_mul_lo_u32", "v_mul_hi_u32", "v_ldexp", "v_min_u32", "v_max_u32", "v_min_i32", "v_max_i32", "v_rndne", "v_trunc", "v_floor", "v_fract")
TRANS = ("v_sqrt_f32", "v_rcp_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32")
CMP = ("v_cmp", "v_cndmask")


def costs():
    path = os.path.join(ROOT, "profiles", f"{tag}_ubench_valu_issue.jsonl")
    c = {}
    for line in open(path):
        if not line.startswith("{"):
            continue
        j = json.loads(line)
        if j["waves_per_simd"] == WAVES and j["chains"] == 8:
            c[j["instr"]] = j["cycles_per_machine_instr_per_simd"]
    return {"fast": (c["v_add_f32"] + c["v_mul_f32"] + c["v_sub_f32"] + c["v_add_u32"] + c["v_mov_b32"]) / 5.0,
            "half": (c["v_min_f32"] + c["v_max_f32"] + c["v_med3_f32"] + c["v_min3_f32"] + c["v_fma_f32"] + c["v_pk_add_f32"] + c["v_pk_mul_f32"] + c["v_pk_fma_f32"]) / 8.0,
            "trans": (c["v_sqrt_f32"] + c["v_rcp_f32"]) / 2.0,
            "cmp": c["v_cmp_lt_f32 vcc + v_cndmask_b32 vcc (2 instr)"]}


def classify(op):
    if op.startswith(TRANS):
        return "trans"
    if op.startswith(CMP):
        return "cmp"
    if op.startswith(HALF):
        return "half"
    if op.startswith(FAST):
        return "fast"
    return "fast_unlisted"


def main():
    src = os.path.join(ROOT, "bevy-signed-distance-mesh-generation_amd", "csrc", "kernels.hip")
    asm = "/tmp/_kernels_mix.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-w", "-S",
                    "--cuda-device-only", "-o", asm, src], check=True)
    lines = open(asm).read().splitlines()
    cost = costs()
    consts = json.load(open(os.path.join(ROOT, "profiles", "constants.json")))
    pmc = (consts.get("step_kernels") or {}).get("sdobj_512", {})
    clock = consts.get("shader_clock_ghz_under_valu_load", 2.2)
    out = {"what": __doc__.split("\n\n")[0], "waves_per_simd_of_the_costs": WAVES, "cost_cycles_per_instr_per_simd": {k: round(v, 3) for k, v in cost.items()},
           "clock_ghz": clock, "kernels": {}}
    for name, sym in KERNELS.items():
        start = next(i for i, l in enumerate(lines) if l.startswith(sym))
        end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
        blocks, cur = [], ["entry", collections.Counter()]
        for l in lines[start:end]:
            s = l.strip()
            if re.match(r"^\.LBB\d+_\d+:", s):
                blocks.append(cur)
                cur = [s.split(":")[0], collections.Counter()]
                continue
            if not s or s[0] in ";.":
                continue
            cur[1][s.split()[0]] += 1
        blocks.append(cur)
        hot = max(blocks, key=lambda b: sum(v for k, v in b[1].items() if k.startswith("v_")))
        valu = {k: v for k, v in hot[1].items() if k.startswith("v_")}
        by_class = collections.Counter()
        for op, n in valu.items():
            by_class[classify(op)] += n
        n_valu = sum(valu.values())
        cyc = sum(n * cost["fast" if c == "fast_unlisted" else c] for c, n in by_class.items())
        row = {"hot_block": hot[0], "valu_instr": n_valu, "s_nop": hot[1].get("s_nop", 0), "by_class": dict(by_class),
               "unlisted_as_fast": sorted(op for op in valu if classify(op) == "fast_unlisted"),
               "cycles_per_pass_per_wave": round(cyc, 1), "mean_cycles_per_instr": round(cyc / n_valu, 3)}
        k = pmc.get(name)
        if k and k.get("wave_instr"):
            bound_us = k["wave_instr"] * cyc / n_valu / (1024 * clock * 1e9) * 1e6
            row.update({"wave_instr_pmc": k["wave_instr"], "issue_bound_us": round(bound_us, 2), "trace_avg_us": k.get("avg_us"),
                        "frac_of_issue_bound": round(bound_us / k["avg_us"], 3) if k.get("avg_us") else None})
        out["kernels"][name] = row
    path = os.path.join(ROOT, "profiles", f"{tag}_issue_bound.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
