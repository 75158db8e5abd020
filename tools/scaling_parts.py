#!/usr/bin/env python3
"""The measurable PARTS of the one-mesh multi-GPU step, on the one GPU there is: for R = 2, 4, 8 balanced slabs of the workload scene at
512^3 .. 4096^3, the wall time of every rank's own slab step (its regen + weld: what runs concurrently on R GPUs), the whole-mesh step
beside it, the protocol + merge cost measured on a one-rank RCCL communicator (the root merges the WHOLE mesh whatever R is, so this part
does not shrink with R), and the bytes the largest peer message would put on its link.  From those: the compute-side bound
t1 / max_r(slab_r), and a modelled step  max_r(slab_r) + protocol_and_merge + bytes / link_rate  (z-slabs)  or  max_r(slab_r) +
protocol (x-slabs: nothing proportional to the mesh crosses a link).  A MODEL built from measured parts, not a measurement of R GPUs.
usage: scaling_parts.py [scene] [out.json]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sdmesh_pkg  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "sdobj"
out_path = sys.argv[2] if len(sys.argv) > 2 else None
LINK_GBPS = 50.0                                   # effective per xGMI link (153 GB/s raw, 7 links into the root)
pkg = sdmesh_pkg.load()


def wall(fn, reps):
    for _ in range(3):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    return (time.perf_counter() - t0) / reps * 1e3, r


out = {"scene": scene, "link_gbps_assumed": LINK_GBPS, "what": __doc__.split("usage")[0].strip(), "sizes": {}}
for k in (4, 5, 6, 7):
    n = 32 << k
    reps = {4: 200, 5: 100, 6: 30, 7: 10}[k]
    h = pkg.CudaHandler(0, scene)
    t1, info = wall(lambda: h.regen(k), reps)
    h.comm_init(pkg.CudaHandler.comm_unique_id(), 1, 0)
    tz, _ = wall(lambda: h.sharded_regen(k, 0), reps)
    tx, _ = wall(lambda: h.distributed_regen(k)[0], reps)
    h.comm_destroy()
    proto = {"zslab": max(0.0, tz - t1), "xslab": max(0.0, tx - t1)}
    row = {"t_1gpu_ms": round(t1, 4), "triangles": int(info.triangles), "vertices": int(info.vertices),
           "one_rank_step_ms": {"zslab": round(tz, 4), "xslab": round(tx, 4)}, "ranks": {}}
    for R in (2, 4, 8):
        per = {}
        for form, mode in (("zslab", pkg.PART_ZSLAB), ("xslab", pkg.PART_XSLAB)):
            bounds = h.balance_slabs(mode, R, min(k, 3))
            slabs, tris, verts = [], [], []
            for r in range(R):
                ms, si = wall(lambda: h.regen(k, mode, r, R), max(5, reps // 4))
                slabs.append(round(ms, 4)); tris.append(int(si.triangles)); verts.append(int(si.vertices))
            crit = max(slabs)
            xfer = max(24 * v + 12 * t for v, t in zip(verts[1:], tris[1:])) / (LINK_GBPS * 1e6) if form == "zslab" else 0.0
            # z-slabs: the root merges the whole mesh, the one-rank cost stands as it is; x-slabs: a rank numbers and rewrites its OWN part, so
            # the size-proportional share of the one-rank cost shrinks with the largest slab's share of the triangles (0.03 ms of it is fixed)
            cost = proto[form] if form == "zslab" else min(proto[form], 0.03) + max(0.0, proto[form] - 0.03) * max(tris) / max(1, sum(tris))
            step = crit + cost + xfer
            per[form] = {"slab_bounds": bounds, "slab_ms": slabs, "slab_triangles": tris, "compute_bound_speedup": round(t1 / crit, 2),
                         "protocol_and_merge_ms": round(cost, 4), "modelled_transfer_ms": round(xfer, 4),
                         "modelled_step_ms": round(step, 4), "modelled_speedup": round(t1 / step, 2)}
        row["ranks"][str(R)] = per
        print(n, R, {f: (p["compute_bound_speedup"], p["modelled_speedup"]) for f, p in per.items()}, flush=True)
    h.close()
    out["sizes"][f"{n}^3"] = row
if out_path:
    json.dump(out, open(out_path, "w"), indent=1)
