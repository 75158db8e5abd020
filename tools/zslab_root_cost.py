#!/usr/bin/env python3
"""The root's share of the one-mesh z-slab step, measured on ONE GPU with the slabs meshed one after the other:
  indexed form  (sdmesh_sharded_regen_rehearsal): R welded slab regens + concatenation + the indexed merge
  raw form      R unwelded slab regens + sdmesh_merge_by_column + sdmesh_weld_device_triangles (the first-seen weld of every corner)
against the slab regens alone.  usage: zslab_root_cost.py [nranks] [levels ...]"""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sdmesh_pkg
pkg = sdmesh_pkg.load()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
levels = [int(a) for a in sys.argv[2:]] or [4, 5, 6]
med = lambda f, n=7: statistics.median([f() for _ in range(n + 2)][2:]) * 1e3      # noqa: E731


def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return time.perf_counter() - t0


for k in levels:
    hs = [pkg.CudaHandler(0, "sdobj") for _ in range(R)]
    for h in hs:
        h.balance_slabs(pkg.PART_ZSLAB, R, 3)
    info = pkg.CudaHandler.sharded_regen_rehearsal(hs, k, 0)
    t_idx = med(lambda: timed(lambda: pkg.CudaHandler.sharded_regen_rehearsal(hs, k, 0)))
    t_merge = statistics.median([pkg.CudaHandler.sharded_regen_rehearsal(hs, k, 0).device_ms for _ in range(7)])
    t_slabs_w = med(lambda: timed(lambda: [hs[r].regen(k, pkg.PART_ZSLAB, r, R) for r in range(R)]))
    # raw form
    for h in hs:
        h.set_option("skip_weld", 1)
    T = info.triangles
    counts = torch.zeros((R, 1024), dtype=torch.int32, device="cuda:0")
    lists = []
    for r in range(R):
        i = hs[r].regen(k, pkg.PART_ZSLAB, r, R)
        buf = torch.empty((max(i.triangles, 1), 18), dtype=torch.float32, device="cuda:0")
        hs[r].device_triangles_into(buf.data_ptr(), buf.shape[0]); hs[r].column_triangle_counts_into(counts[r].data_ptr())
        lists.append(buf)
    merged = torch.empty((T, 18), dtype=torch.float32, device="cuda:0")
    hs[0].set_option("skip_weld", 0)

    def raw_root():
        n = hs[0].merge_by_column([b.data_ptr() for b in lists], counts.data_ptr(), merged.data_ptr(), T)
        hs[0].weld_device_triangles(merged.data_ptr(), n)
    raw_root()
    t_raw_root = med(lambda: timed(raw_root))
    one = pkg.CudaHandler(0, "sdobj"); one.regen(k)
    t_one = med(lambda: timed(lambda: one.regen(k)))
    print(f"{32 << k}^3, {R} z-slabs, {T} triangles: single-GPU regen {t_one:.3f} ms | {R} welded slab regens one after the other {t_slabs_w:.3f} ms | "
          f"indexed form in all {t_idx:.3f} ms (the rehearsal's per-slab column counts, waits and concatenation included); the root's indexed merge alone {t_merge:.3f} ms (HIP events) | raw form's root: merge + weld of every corner {t_raw_root:.3f} ms "
          f"| gather payload {info.vertices * 24 / 1e6 + T * 12 / 1e6:.1f} MB indexed vs {T * 72 / 1e6:.1f} MB raw", flush=True)
    one.close()
    for h in hs:
        h.close()
    del lists, merged
