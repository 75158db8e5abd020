#!/usr/bin/env python3
"""Golden rows for the full-size GPU parity test: the oracle at 2048^3 and 4096^3 (about 10 minutes on 8 cores).
Writes tests/golden/oracle_golden_big.json (see its _provenance)."""
import os
import sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O
O.build(); O.set_threads(8)
out = {}
for k in (6, 7):
    t = time.time()
    r = O.regen("sdobj", k)
    row = {"k": k, "voxels": int(len(r["voxels"])), "tris": int(len(r["idx"])), "verts": int(len(r["pos"])),
           "vox_fnv": "%016x" % O.fnv1a64(r["voxels"]), "idx_fnv": "%016x" % O.fnv1a64(r["idx"]),
           "pos_fnv": "%016x" % O.fnv1a64(r["pos"]), "nrm_fnv": "%016x" % O.fnv1a64(r["nrm"]), "seconds": round(time.time() - t, 1)}
    print(row, flush=True)
    out[k] = row


prov = ("oracle/sdmesh_oracle.c (CPU restatement of the reference algorithm) run by tools/gen_golden_big.py: sd_obj at 2048^3 "
        "(k=6) and 4096^3 (k=7).  Hashes are FNV-1a-64 of the raw little-endian arrays.")
json.dump({"_provenance": prov, "sdobj": {"rows": [out[k] for k in sorted(out)]}},
          open(os.path.join(ROOT, "tests", "golden", "oracle_golden_big.json"), "w"), indent=1)
