#!/usr/bin/env python3
"""Register / LDS / occupancy of every kernel as hipcc reports them (-Rpass-analysis=kernel-resource-usage; cross-compiles
without a GPU) -> profiles/<tag>_kernel_resource_usage.csv.  usage: kernel_resources.py [tag]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "bevy-signed-distance-mesh-generation_amd", "csrc", "kernels.hip")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/_kres.o"]
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in txt.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass-analysis", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
out = os.path.join(ROOT, "profiles", f"{tag}_kernel_resource_usage.csv")
with open(out, "w") as f:
    f.write("kernel,VGPRs,AGPRs,SGPRs,scratch_bytes_per_lane,LDS_bytes_per_block,occupancy_waves_per_SIMD\n")
    for r, n in zip(rows, names):
        short = re.sub(r"^void ", "", re.sub(r"\(.*", "", n))
        f.write('"%s",%s,%s,%s,%s,%s,%s\n' % (short, r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"),
                                              r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
print(out, len(rows), "kernels")
