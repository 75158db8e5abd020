#!/usr/bin/env python3
"""bench.py -- headline benchmark: SDF mesh extraction of the reference scene at 512^3 on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full regen of the workload on the device: level-0 field -> 4 hierarchical refines
-> marching cubes -> Newton projection + normals -> orientation -> vertex weld, ending with the indexed
mesh resident in HBM (SURVEY.md 8(d) t_regen).  Nothing is uploaded: the SDF is the input.

metric   Mvoxels/s = N_grid^3 * steps / wall  (BASELINE.json: "Mvoxels/s + Mtris/s for 512^3 SDF mesh extract;
         1/2/4/8-GPU scaling"); Mtris/s rides along in "mtris_per_s".
N = 1    the step runs on one GPU.
N > 1    the step is ONE mesh over all N GPUs (`sdmesh_sharded_regen`): rank r meshes level-0 z-slab r,
         RCCL all-gather of the per-column counts, RCCL send/recv of the per-slab triangle lists to rank 0, merge kernel,
         weld on rank 0 -> "scaling": "strong".  `one_mesh` repeats that at 256^3 / 512^3 / 1024^3 (BASELINE configs[3]) /
         2048^3 / 4096^3 with the same-size 1-rank time measured in the same invocation, for both one-mesh forms (z-slab
         gather + root weld; x-slab distributed weld), plus `ranks_seen` from an RCCL all-reduce.  N independent regens
         (no exchange at all) are reported as the extra `replicas`.  A failure or hang in any multi-rank leg is a failure
         of the run: every rank exits non-zero.
roofline the north-star HBM pass P1 = dense (N+1)^3 f32 SDF sample, timed live with HIP events on the
         stream the kernel runs on: algorithmic bytes 4*(N+1)^3 per launch / average launch time, vs 8 TB/s.
roofline_step  the two throughput-bound kernels of the timed step itself (k_project1, k_tile_first with the orientation test): VALU
         machine instructions per launch (SQ_INSTS_VALU from the committed PMC pass, profiles/constants.json) / live time, vs the
         spec VALU peak and vs the issue-cost bound of the kernel's own instruction mix (tools/ubench/valu_issue.hip, tools/isa_mix.py).
cpu_baseline  the CPU oracle (oracle/, a port of the reference algorithm) on the host cores of this box,
         same workload, whole regen, all cores ("port").  Reported, not a target.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import sdmesh_pkg  # noqa: E402

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"
# VALU issue peak in wave64 instructions per second: 157.3 TFLOP/s f32 (MI355X_MICROARCH.md "Peak FP32 (vector)") / 2 flop
# per FMA lane / 64 lanes = 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction
VALU_PEAK_WAVE_INSTR_PER_S = 157.3e12 / 2 / 64


def load_constants():
    """Measured constants that cannot be collected from inside this process (PMC passes), committed under profiles/ by
    tools/install_profiles.py together with the rocprofv3 files they come from."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "constants.json")))
    except Exception:                       # noqa: BLE001 -- the fields that need them become null
        return {}


def host_cores():
    """CPU threads this job may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return min(n, 64)


def cpu_baseline(scene, levels, params):
    """The oracle (CPU port of the reference algorithm) on all host cores: one whole regen of the same
    workload.  ~20-30 s of CPU work at 512^3 sd_obj."""
    from oracle import oracle as O
    O.build()
    cores = host_cores()
    O.set_threads(cores)
    t0 = time.perf_counter()
    r = O.regen(scene, levels, params)
    dt = time.perf_counter() - t0
    n = 32 << levels
    out = {"value": round(n ** 3 / dt / 1e6, 3), "unit": "Mvoxels/s", "cores": cores, "kind": "port",
           "sample": f"one full regen of {scene} {n}^3 (refine x{levels} + triangles + weld), OpenMP over voxels, "
                     f"{dt:.2f} s wall, {len(r['idx'])} triangles",
           "mtris_per_s": round(len(r["idx"]) / dt / 1e6, 4), "seconds": round(dt, 3)}
    # the same port on ONE thread (one level coarser, so it stays a ~10 s sample), and a conventional dense
    # CPU marching cubes (sample every lattice point -> classify -> count triangles; no projection, no normals, no weld)
    lv1 = max(levels - 1, 0)
    n1 = 32 << lv1
    O.set_threads(1)
    t0 = time.perf_counter()
    O.regen(scene, lv1, params)
    d1 = time.perf_counter() - t0
    out["single_thread"] = {"value": round(n1 ** 3 / d1 / 1e6, 3), "unit": "Mvoxels/s", "cores": 1, "grid": n1, "seconds": round(d1, 3)}
    O.set_threads(cores)
    t0 = time.perf_counter()
    nt = O.dense_mc_count(scene, n, params)
    dd = time.perf_counter() - t0
    out["dense_mc"] = {"value": round(n ** 3 / dd / 1e6, 3), "unit": "Mvoxels/s", "cores": cores, "grid": n, "seconds": round(dd, 3),
                       "triangles": int(nt), "what": "dense sample + classify + triangle count only"}
    return out


class Job:
    """Rank bookkeeping, barrier-bracketed timing (max over ranks) and the hang watchdog."""

    def __init__(self, world, rank):
        self.world, self.rank = world, rank
        self.dog = None

    def sync(self):
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(self, fn, steps, warmup):
        """`warmup` untimed calls, then exactly `steps` timed ones bracketed by barrier + synchronize on both sides;
        seconds for the timed region, max over ranks.  Returns (seconds, last result)."""
        res = None
        for _ in range(warmup):
            res = fn()
        self.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            res = fn()
        self.sync()
        dt = time.perf_counter() - t0
        if self.world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, res

    def arm_watchdog(self, seconds, what):
        """A hung collective or GPU must not look like a pass: after `seconds` every rank writes the reason to stderr and
        exits with status 3."""
        def bail():
            sys.stderr.write(f"bench.py: rank {self.rank}: '{what}' did not finish within {seconds} s -- giving up (exit 3)\n")
            sys.stderr.flush()
            os._exit(3)
        self.disarm_watchdog()
        self.dog = threading.Timer(seconds, bail)
        self.dog.daemon = True
        self.dog.start()

    def disarm_watchdog(self):
        if self.dog is not None:
            self.dog.cancel()
            self.dog = None


def single_gpu_step_times(pkg, local_rank, scene, params, levels_list, graph, reps):
    """ms per regen (wall, this rank alone, no barrier inside the loop) and mesh counts per level: the 1-rank same-size
    reference of every one-mesh figure, measured in the same invocation on the same GPU."""
    out = {}
    for lv in levels_list:
        h = pkg.CudaHandler(local_rank, scene, params)
        h.regen(lv)
        h.set_option("graph", 1 if graph else 0)
        for _ in range(3):
            info = h.regen(lv)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            info = h.regen(lv)
        torch.cuda.synchronize()
        out[lv] = {"ms": (time.perf_counter() - t0) / reps * 1e3, "triangles": int(info.triangles), "vertices": int(info.vertices)}
        h.close()
    return out


def one_mesh_legs(pkg, job, local_rank, scene, params, levels_list, base_levels, steps, warmup, graph, rehearse=False):
    """ONE mesh over all ranks, both forms, every size; plus the primary timing (the faster form at the workload size)."""
    from bevy_signed_distance_mesh_generation_amd import partition as P
    world, rank = job.world, job.rank
    dev = torch.device("cuda", local_rank)
    t1 = single_gpu_step_times(pkg, local_rank, scene, params, levels_list, graph, 20)
    if world > 1:                                    # rank 0's single-GPU times are the reference for everyone
        box = [t1]
        dist.broadcast_object_list(box, src=0)
        t1 = box[0]
    # --- z-slabs welded per rank, RCCL gather of the vertex / index buffers to rank 0, indexed merge there: all inside sdmesh_sharded_regen
    h = pkg.CudaHandler(local_rank, scene, params)
    mesher = P.ShardedMesher(pkg, h, base_levels, rank, world, dev, transport="collectives" if rehearse else "rccl")
    ranks_seen = mesher.ranks_seen
    try:                                             # every kind of message the steps use, once, each rank its own peer (never fatal here)
        selftest = h.comm_selftest()
    except Exception as e:                           # noqa: BLE001
        selftest = f"failed: {e}"
    zbounds, xbounds = mesher.bounds, None
    zslab, primary = {}, None
    for lv in levels_list:
        mesher.levels = lv
        mesher.step()                                # capacity discovery (synchronous), untimed
        k, w = (steps, warmup) if lv == base_levels else (max(3, min(20, steps // 4)), 2)
        dt, info = job.timed(mesher.step, k, w)
        n = 32 << lv
        tri = torch.tensor([int(info.triangles) if info is not None else 0, int(info.vertices) if info is not None else 0],
                           dtype=torch.int64, device=dev)
        if world > 1:
            dist.all_reduce(tri, op=dist.ReduceOp.MAX)
        ms = dt / k * 1e3
        if (int(tri[0]), int(tri[1])) != (t1[lv]["triangles"], t1[lv]["vertices"]):
            raise RuntimeError(f"sharded mesh at {n}^3 has {int(tri[0])} triangles / {int(tri[1])} vertices, the single-GPU mesh "
                               f"{t1[lv]['triangles']} / {t1[lv]['vertices']}")
        zslab[f"{n}^3"] = {"ms_per_regen": round(ms, 4), "t_1rank_ms": round(t1[lv]["ms"], 4),
                           "speedup_vs_1rank": round(t1[lv]["ms"] / ms, 3), "mvoxels_per_s": round(n ** 3 / ms / 1e3, 1),
                           "mtris_per_s": round(int(tri[0]) / ms / 1e3, 1), "triangles": int(tri[0]), "vertices": int(tri[1]), "steps": k,
                           "route": {1: "planned (one host synchronisation)", 2: "synchronous", 3: "planned, repeated synchronously"}.get(h.sharded_route())}
        if lv == base_levels:
            primary = (dt, k, int(tri[0]), int(tri[1]), "zslab_indexed_gather")
    mesher.close()
    h.close()
    # --- x-slabs, every rank welds its slab, neighbours unify the shared faces, the mesh stays distributed
    h = pkg.CudaHandler(local_rank, scene, params)
    xslab = {}
    for lv in levels_list:
        dm = P.DistributedMesher(pkg, h, lv, rank, world, dev, collectives=rehearse)
        xbounds = dm.bounds
        dm.step()
        k, w = (steps, warmup) if lv == base_levels else (max(3, min(20, steps // 4)), 2)
        dt, dinfo = job.timed(lambda: h.distributed_regen(lv)[0], k, w)     # (the library call alone: dm.step also wraps the part in torch tensors)
        cnt = torch.tensor([int(dinfo.vertices), int(dinfo.triangles)], dtype=torch.int64, device=dev)
        if world > 1:
            dist.all_reduce(cnt)
        n = 32 << lv
        ms = dt / k * 1e3
        if (int(cnt[1]), int(cnt[0])) != (t1[lv]["triangles"], t1[lv]["vertices"]):
            raise RuntimeError(f"distributed mesh at {n}^3 has {int(cnt[1])} triangles / {int(cnt[0])} vertices, expected "
                               f"{t1[lv]['triangles']} / {t1[lv]['vertices']}")
        xslab[f"{n}^3"] = {"ms_per_regen": round(ms, 4), "t_1rank_ms": round(t1[lv]["ms"], 4),
                           "speedup_vs_1rank": round(t1[lv]["ms"] / ms, 3), "mvoxels_per_s": round(n ** 3 / ms / 1e3, 1),
                           "triangles": int(cnt[1]), "vertices": int(cnt[0]), "steps": k,
                           "route": {1: "planned (one host synchronisation)", 2: "synchronous", 3: "planned, repeated synchronously"}.get(h.distributed_route())}
    h.close()
    return primary, {"n_gpus": world, "ranks_seen": ranks_seen, "comm_selftest_mismatches": selftest, "slab_bounds": {"z": zbounds, "x": xbounds},
                     "zslab_indexed_gather": {"what": "sdmesh_sharded_regen (indexed form): level-0 z-slab per rank, meshed AND welded there; ncclAllGather of "
                                                      "1024 column counts + sizes + status; ncclSend/ncclRecv of the slabs' vertex and index buffers "
                                                      "(24 B per vertex + 12 B per triangle) to rank 0; there the buffers are merged by (column, rank) "
                                                      "segments without welding again; mesh identical to 1 GPU (counts checked here, bits in tests)",
                                              **zslab},
                     "xslab_distributed_weld": {"what": "sdmesh_distributed_regen: level-0 x-slab per rank, every rank welds its slab, x-neighbours unify "
                                                        "the vertices of their shared face by weld key (two ncclSend/ncclRecv of the face band + two 4-byte "
                                                        "ncclAllGather), indices become global, the mesh stays distributed",
                                                **xslab}}


def dense_p1_sharded_leg(pkg, job, local_rank, scene, params, grids=(256, 512, 1024), launches=50):
    """P1 (the north star's HBM pass) sharded over the ranks: rank r samples the x planes [r (N+1) / R, (r+1) (N+1) / R) of the same
    dense lattice into its own HBM, no exchange of any kind (the parts are bit for bit the ranges of the whole volume: -m gpu
    test_dense_slabs_concatenate_to_the_whole_volume).  Timed like a step: `launches` back-to-back launches per rank between
    barrier + synchronize, max over ranks; aggregate = 4 (N+1)^3 bytes / that time, against R x 8 TB/s.  The 1-rank figure beside it
    is the whole volume on rank 0 alone, same invocation."""
    from bevy_signed_distance_mesh_generation_amd import partition as P
    world, rank = job.world, job.rank
    h = pkg.CudaHandler(local_rank, "sphere")
    out = {"what": "dense (N+1)^3 f32 SDF-sample pass, one x-slab of lattice planes per GPU, no data-path collective; aggregate over all "
                   "ranks, frac = aggregate / (n_gpus x 8 TB/s); t_1rank = the whole volume on rank 0 alone in the same invocation",
           "n_gpus": world, "launches": launches}
    for n in grids:
        S = n + 1
        x0, x1 = P.dense_slab_planes(n, rank, world)
        alg = 4.0 * S ** 3
        whole = torch.empty(S ** 3 if rank == 0 else 4, dtype=torch.float32, device="cuda")
        mine = torch.empty(max(4, (x1 - x0) * S * S), dtype=torch.float32, device="cuda")
        for sc in dict.fromkeys(("sphere", scene)):
            h.set_scene(sc, params if sc == scene else None)
            t1 = None
            if rank == 0:
                h.time_kernel(0, n, whole.data_ptr(), 200)                       # clocks settle under the new load (see roofline_legs)
                t1 = h.time_kernel(0, n, whole.data_ptr(), launches)
            job.sync()
            h.sample_dense_slab(n, x0, x1, mine.data_ptr(), iters=200, timed=False)
            dt, _ = job.timed(lambda: h.sample_dense_slab(n, x0, x1, mine.data_ptr(), iters=1, timed=False), launches, 20)
            ms = dt / launches * 1e3
            gbps = alg / (ms * 1e-3) / 1e9
            row = {"ms_per_pass": round(ms, 5), "achieved": round(gbps, 1), "unit": "GB/s", "peak": HBM_PEAK_GBPS * world,
                   "frac": round(gbps / (HBM_PEAK_GBPS * world), 4), "gsamples_per_s": round(S ** 3 / (ms * 1e-3) / 1e9, 1),
                   "mvoxels_per_s": round(n ** 3 / ms / 1e3, 1), "bytes_per_pass": alg, "planes_of_rank0": [int(x0), int(x1)] if rank == 0 else None}
            if world > 1:
                box = [t1]
                dist.broadcast_object_list(box, src=0)
                t1 = box[0]
            row["t_1rank_ms"] = round(t1, 5)
            row["speedup_vs_1rank"] = round(t1 / ms, 3)
            out.setdefault(f"{n}^3", {})[sc] = row
        del whole, mine
    h.close()
    return out


def reference_abi_leg(pkg, local_rank, scene, params, levels):
    """The reference's own launch structure on this GPU: its two kernels -- one thread per voxel, raw 8n / 5n slot layouts, by-value
    VoxelField kernargs (src/cuda/mod.rs:149-177,226-250) -- as the B1 code object exports them, launched through the HIP module API
    on voxel lists that are already on the device.  Kernel time only (HIP events): no upload / download of the 8n / 5n slot buffers,
    no host-side retain, no host weld, all of which the reference pays on top.  Reported beside the native chain, never `value`."""
    import ctypes
    hip = None
    for name in ("libamdhip64.so", "libamdhip64.so.7", os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")):
        try:
            hip = ctypes.CDLL(name)
            break
        except OSError:
            continue
    if hip is None:
        return {"error": "HIP runtime library not found for the module API"}
    A = pkg._abi
    mod, f_ref, f_tri = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    blob = open(pkg.HSACO_PATH, "rb").read()
    if hip.hipModuleLoadData(ctypes.byref(mod), blob) != 0:
        return {"error": "hipModuleLoadData failed"}
    hip.hipModuleGetFunction(ctypes.byref(f_ref), mod, b"compute_refine_voxel_field_by_sdf")
    hip.hipModuleGetFunction(ctypes.byref(f_tri), mod, b"compute_surface_triangles_from_voxel_field_by_sdf")
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    hip.hipEventCreate(ctypes.byref(e0)); hip.hipEventCreate(ctypes.byref(e1))

    def timed_launch(fn, n, *args):
        arr = (ctypes.c_void_p * len(args))(*[ctypes.cast(ctypes.pointer(a), ctypes.c_void_p) for a in args])
        grid = (n + A.BLOCK_SIZE - 1) // A.BLOCK_SIZE
        ms = []
        for _ in range(4):
            hip.hipEventRecord(e0, None)
            if hip.hipModuleLaunchKernel(fn, grid, 1, 1, A.BLOCK_SIZE, 1, 1, 0, None, arr, None) != 0:
                raise RuntimeError("hipModuleLaunchKernel failed")
            hip.hipEventRecord(e1, None)
            hip.hipEventSynchronize(e1)
            t = ctypes.c_float(0)
            hip.hipEventElapsedTime(ctypes.byref(t), e0, e1)
            ms.append(t.value)
        return min(ms[1:])

    h = pkg.CudaHandler(local_rank, scene, params)
    refine_ms, rows = 0.0, []
    tri_ms, n_last = None, 0
    for lv in range(levels + 1):
        h.regen(lv)
        f = h.download_voxels()                                     # level-lv voxel list (the reference keeps it on the host between steps)
        n = len(f.voxels)
        d_in = torch.from_numpy(np.ascontiguousarray(f.voxels)).to("cuda")
        fin = A.VoxelField(A.Point(*map(float, f.voxel_size)), d_in.data_ptr(), n)
        if lv < levels:
            d_out = torch.empty((8 * n, 3), dtype=torch.float32, device="cuda")
            fout = A.VoxelField(A.Point(0.0, 0.0, 0.0), d_out.data_ptr(), 8 * n)
            torch.cuda.synchronize()
            ms = timed_launch(f_ref, n, fin, fout)
            refine_ms += ms
            rows.append({"level": lv, "voxels": n, "refine_kernel_ms": round(ms, 4)})
            del d_out
        else:
            d_tri = torch.empty((5 * n, 18), dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            tri_ms = timed_launch(f_tri, n, fin, ctypes.c_void_p(d_tri.data_ptr()))
            n_last = n
            del d_tri
        del d_in
    h.close()
    hip.hipEventDestroy(e0); hip.hipEventDestroy(e1); hip.hipModuleUnload(mod)
    return {"what": "the reference's launch structure on this GPU: its two kernels as this build's reference-ABI code object restates them (B1: one thread per voxel, "
                    "every voxel projects all its own triangle corners, raw 8n / 5n slot layouts, by-value kernargs), launched as src/cuda/mod.rs launches them, "
                    "on device-resident voxel lists: kernel time only (HIP events) -- the reference also pays the upload / download of the slot buffers, "
                    "the host-side retain and the host HashMap weld on top; beside it the native chain's device time for the same work incl. the weld",
            "refine_kernels_ms": round(refine_ms, 4), "triangles_kernel_ms": round(tri_ms, 4), "kernels_total_ms": round(refine_ms + tri_ms, 4),
            "voxels_meshed": n_last, "triangle_slots_written_MB": round(5 * n_last * 72 / 1e6, 1), "per_level": rows}


def host_api_leg(pkg, local_rank, scene, params, levels_list=(3, 4), reps=12):
    """The reference's OWN call pattern through the C ABI, PCIe-inclusive (never `value`): host `Vec<Point>` in and out at every call, as
    the renderer drives CudaHandler (src/renderer/mod.rs:168-171, src/cuda/mod.rs:105-122, 124-202, 204-346): create_cuda_voxel_field,
    k x refine_voxel_field (upload n points, download the kept children), voxel_field_to_mesh (upload, chain, counts) and the mesh
    download (vertices + indices to pageable host memory).  Median wall ms per call over `reps` runs, ctypes overhead included."""
    import ctypes
    from bevy_signed_distance_mesh_generation_amd import handler as H
    A = pkg._abi
    h = pkg.CudaHandler(local_rank, scene, params)
    lib, hh = h._lib, h._h
    out = {"what": "the reference's host-buffer API through the C ABI, PCIe-inclusive: create_cuda_voxel_field, k x refine_voxel_field (host points in, "
                   "kept children out), voxel_field_to_mesh (host points in, counts out) + sdmesh_download_mesh (24 B per vertex + 12 B per triangle "
                   "to pageable host memory); median wall ms per call; beside it the device-resident regen of the same size.  Not the metric."}
    for k in levels_list:
        rows = {"create_voxel_field": [], "refine": [[] for _ in range(k)], "voxel_field_to_mesh": [], "download_mesh": [], "total": []}
        nv, nt = ctypes.c_size_t(0), ctypes.c_size_t(0)
        sizes = []
        for rep in range(reps + 2):
            t_start = time.perf_counter()
            vox = np.empty((A.MESH_GENERATION_INIT_FACTOR ** 3, 3), np.float32)
            size = np.empty(3, np.float32)
            t0 = time.perf_counter()
            lib.sdmesh_create_voxel_field(vox.ctypes.data, size.ctypes.data)
            tc = time.perf_counter() - t0
            tr = []
            n = vox.shape[0]
            sz = H._pt(size)
            for lv in range(k):
                outb = np.zeros((8 * n, 3), np.float32); outb.fill(0.0)      # (touched, like the reference's zero-initialised Vec: src/cuda/mod.rs:179-183)
                n_out, size_out = ctypes.c_size_t(0), A.Point()
                t0 = time.perf_counter()
                rc = lib.sdmesh_refine_voxel_field(hh, vox.ctypes.data, n, sz, outb.ctypes.data, 8 * n, ctypes.byref(n_out), ctypes.byref(size_out))
                tr.append(time.perf_counter() - t0)
                if rc != 0:
                    raise RuntimeError("sdmesh_refine_voxel_field failed")
                vox, n, sz = outb, n_out.value, size_out
            t0 = time.perf_counter()
            rc = lib.sdmesh_voxel_field_to_mesh(hh, vox.ctypes.data, n, sz, None, 0, ctypes.byref(nv), None, 0, ctypes.byref(nt))
            tm = time.perf_counter() - t0
            if rc != 0:
                raise RuntimeError("sdmesh_voxel_field_to_mesh failed")
            verts, idx = np.empty((nv.value, 2, 3), np.float32), np.empty((nt.value, 3), np.uint32)
            verts.fill(0.0); idx.fill(0)                                 # (touched pages: the copy is timed, not the first-touch faults)
            t0 = time.perf_counter()
            lib.sdmesh_download_mesh(hh, verts.ctypes.data, idx.ctypes.data)
            td = time.perf_counter() - t0
            if rep >= 2:
                rows["create_voxel_field"].append(tc); rows["voxel_field_to_mesh"].append(tm); rows["download_mesh"].append(td)
                for lv in range(k):
                    rows["refine"][lv].append(tr[lv])
                rows["total"].append(time.perf_counter() - t_start)
            sizes = [n, nv.value, nt.value]
        med = lambda v: round(float(np.median(v)) * 1e3, 4)     # noqa: E731
        for _ in range(300):                                             # (the PCIe-bound calls above leave the GPU mostly idle: let the clocks come back)
            h.regen(k)
        t0 = time.perf_counter()
        for _ in range(100):
            h.regen(k)
        dres = (time.perf_counter() - t0) / 100 * 1e3
        out[f"{32 << k}^3"] = {"create_voxel_field_ms": med(rows["create_voxel_field"]), "refine_voxel_field_ms": [med(v) for v in rows["refine"]],
                               "voxel_field_to_mesh_ms": med(rows["voxel_field_to_mesh"]), "download_mesh_ms": med(rows["download_mesh"]),
                               "mesh_plus_download_ms": round(med(rows["voxel_field_to_mesh"]) + med(rows["download_mesh"]), 4),
                               "whole_sequence_ms": med(rows["total"]), "voxels": sizes[0], "vertices": sizes[1], "triangles": sizes[2],
                               "bytes_up_last_call": 12 * sizes[0], "bytes_down_mesh": 24 * sizes[1] + 12 * sizes[2],
                               "device_resident_regen_ms": round(dres, 4)}
    h.close()
    return out


def steady_ms(h, which, n, ptr, reps, warm_ms=40.0, windows=3):
    """HIP-event time per launch of one kernel in steady state: at least `warm_ms` of THAT kernel first (the chip's clock and power
    management settle over ~25 ms under a new load; a fixed number of warm-up launches is 3 ms at 256^3 and 120 ms at 1024^3), then
    `windows` windows of `reps` back-to-back launches.  Returns (median window, the windows)."""
    t = h.time_kernel(which, n, ptr, 5)
    done = 5 * t
    while done < warm_ms:
        k = int(min(4000, max(10, (warm_ms - done) / max(t, 1e-4) + 1)))
        t = h.time_kernel(which, n, ptr, k)
        done += t * k
    ws = [h.time_kernel(which, n, ptr, reps) for _ in range(windows)]
    return float(np.median(ws)), ws


def roofline_legs(h, out, n_grid, scene, params, consts):
    """P1 / P2a / write ceiling, HIP-event timed on the handle's stream (N = 1 only).  One buffer for every grid, allocated before the
    first leg; every figure = the median of three windows behind a time-based warm-up of the kernel under test (steady_ms)."""
    big = torch.empty((2 * n_grid + 1) ** 3, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ptr = big.data_ptr()
    window = "median of 3 windows of back-to-back launches (HIP events) behind >= 40 ms of the same kernel; `windows_ms` = the three"
    traffic = (consts.get("p1_traffic_bytes") or {}).get(str(n_grid), {})

    def row(sc, g, reps, burst=False):
        bb = 4.0 * (g + 1) ** 3
        r = {}
        if burst:
            h.time_kernel(0, g, ptr, 20)                                     # (instruction cache)
            ms_first = h.time_kernel(0, g, ptr, 50)                          # the first 50 of a burst, clocks not settled yet
            r["first_50_of_burst"] = {"avg_launch_ms": round(ms_first, 5), "frac": round(bb / (ms_first * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
        ms, ws = steady_ms(h, 0, g, ptr, reps)
        gbps = bb / (ms * 1e-3) / 1e9
        r.update({"bytes_per_launch": bb, "avg_launch_ms": round(ms, 5), "achieved": round(gbps, 1), "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBPS, 4),
                  "mvoxels_per_s": round(g ** 3 / ms / 1e3, 1), "gsamples_per_s": round((g + 1) ** 3 / (ms * 1e-3) / 1e9, 1),
                  "windows_ms": [round(w, 5) for w in ws], "window_spread": round((max(ws) - min(ws)) / ms, 4)})
        return r

    roofs, by_grid = {}, {}
    try:
        for sc in dict.fromkeys(("sphere", scene)):
            h.set_scene(sc, [0.25] if sc == "anim" else None)
            rows = {str(n_grid): row(sc, n_grid, 50, burst=True), str(n_grid // 2): row(sc, n_grid // 2, 600), str(2 * n_grid): row(sc, 2 * n_grid, 20)}
            by_grid[sc] = dict(sorted(rows.items(), key=lambda kv: int(kv[0])))
            roofs[sc] = dict({"bound": "hbm", "kernel": f"k_sample_dense<{sc}>", "peak": HBM_PEAK_GBPS, "traffic": traffic.get(sc), "window": window},
                             **rows[str(n_grid)])
        # P2a: sign bits of the dense lattice -- the streaming-READ pass (4 B x (N+1)^3 in, 1 bit per sample out)
        alg_bytes = 4.0 * (n_grid + 1) ** 3
        h.set_scene("sphere")
        ms2, ws2 = steady_ms(h, 1, n_grid, ptr, 50)
        gb2 = alg_bytes / (ms2 * 1e-3) / 1e9
        out["roofline_p2_sign_bits"] = {"bound": "hbm", "kernel": "k_sign_bits", "achieved": round(gb2, 1), "peak": HBM_PEAK_GBPS,
                                        "unit": "GB/s", "frac": round(gb2 / HBM_PEAK_GBPS, 4), "traffic": None,
                                        "bytes_per_launch": alg_bytes + alg_bytes / 32, "avg_launch_ms": round(ms2, 5), "windows_ms": [round(w, 5) for w in ws2]}
        # write ceiling on THIS box: P1's store stream (same bytes, same launch shape: one float4 store per thread, one 4 KiB piece
        # per workgroup) with no arithmetic in front
        ms3, ws3 = steady_ms(h, 2, n_grid, ptr, 50)
        gb3 = alg_bytes / (ms3 * 1e-3) / 1e9
    finally:
        h.set_scene(scene, params)
        del big
    out["roofline"] = roofs["sphere"]
    out["roofline"]["write_ceiling"] = {"kernel": "k_stream_fill", "achieved": round(gb3, 1), "unit": "GB/s", "avg_launch_ms": round(ms3, 5),
                                        "windows_ms": [round(w, 5) for w in ws3], "p1_frac_of_ceiling": round(roofs["sphere"]["achieved"] / gb3, 4),
                                        "note": "the same store stream with no arithmetic in front; by box 4 % faster to 3 % slower than the sphere pass"}
    out["roofline"]["note"] = ("P1 dense (N+1)^3 f32 SDF-sample pass, the HBM pass the north star names; claimed on the "
                               "sphere scene (SURVEY 8(d) caveat i); the regen itself is O(surface) and VALU/latency "
                               "bound, see roofline_step and DESIGN.md")
    if scene != "sphere":
        out["roofline_workload_scene"] = roofs[scene]
    # the same pass over a volume eight times as large (BASELINE configs[3]'s grid: a launch is 0.6 ms instead of 0.08, so its ramp and
    # drain weigh an eighth as much) and over one an eighth as large (256^3: a 0.012 ms launch, mostly ramp and drain)
    out["roofline"]["at_twice_the_grid"] = dict(by_grid["sphere"][str(2 * n_grid)], grid=2 * n_grid)
    out["roofline_by_grid"] = by_grid


def roofline_step_leg(pkg, local_rank, scene, params, levels, consts, info):
    """The throughput-bound kernels of the timed step, live: stage times from HIP events between the stages of the planned
    chain (median of 30 regens), VALU wave-instructions per launch from the committed SQ PMC pass."""
    h = pkg.CudaHandler(local_rank, scene, params)
    h.regen(levels)
    h.set_option("profile", 1)
    acc = {}
    for _ in range(32):
        h.regen(levels)
        per = {}
        for name, ms in h.profile_stages():          # (refine_mask / refine_emit appear once per level: summed per regen)
            per[name] = per.get(name, 0.0) + ms
        for name, ms in per.items():
            acc.setdefault(name, []).append(ms)
    rounds = h.projection_rounds()
    # the projection kernel's own wall span, from in-kernel s_memrealtime stamps (constant 100 MHz) of its waves: first wave start
    # to last wave end per XCD, the longest XCD -- no stage markers around it, so nothing but the kernel is inside the interval
    h.set_option("profile", 0)
    h.set_option("project_diag", 1)
    spans = []
    for _ in range(20):
        h.regen(levels)
        d = h.project_diag()
        d = d[d[:, 1] != 0]
        xcc = (d[:, 2] >> np.uint64(28)) & np.uint64(15)
        rt0, rt1 = (d[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64), (d[:, 3] >> np.uint64(32)).astype(np.int64)
        spans.append(max(float((rt1[xcc == x].max() - rt0[xcc == x].min()) % (1 << 32)) / 100.0 for x in np.unique(xcc)))
    proj_span_us = float(np.median(spans))
    h.close()
    stages = {k: round(float(np.median(v)) * 1e3, 2) for k, v in acc.items()}                              # us per regen
    pmc = (consts.get("step_kernels") or {}).get(f"{scene}_{32 << levels}", {})
    icost = consts.get("issue_cost") or {}
    clock = consts.get("shader_clock_ghz_under_valu_load") or 2.2
    out = {"what": "VALU roofline of the step's throughput-bound kernels: wave64 VALU MACHINE instructions per launch (SQ_INSTS_VALU, "
                   "profiles/constants.json) / live time.  frac = against the spec peak 157.3 TF / 2 / 64, which only a stream of packed FMAs "
                   "reaches; frac_of_issue_bound = against the time this kernel's own instruction mix needs if every SIMD issues back to "
                   "back: instructions x mean issue cost (per-kind costs measured by tools/ubench/valu_issue.hip: add/mul ~3.0, min/max/fma/"
                   "packed ~4.3, compare/select ~3.7, sqrt/rcp ~8.3 cycles per SIMD at 5 waves) / (1024 SIMDs x measured shader clock)",
           "peak": VALU_PEAK_WAVE_INSTR_PER_S, "unit": "wave-instr/s", "issue_cost_cycles_per_instr_by_kind": icost.get("cycles_per_instr_by_kind"),
           "shader_clock_ghz": clock, "stage_us": stages, "stage_sum_us": round(sum(stages.values()), 2)}
    # SURVEY 8(d)'s other passes: algorithmic bytes (its table) / live stage time -- reported, not claimed: they are latency- or
    # VALU-bound at this size, and the build never materialises the 72-byte triangle records P4 / P5 count
    A, T, V = int(info.voxel_count[levels]), int(info.triangles), int(info.vertices)
    st = lambda *names: sum(v for k, v in stages.items() if any(k.startswith(n) for n in names))   # noqa: E731
    passes = {}
    for key, what, nbytes, us in (
            ("P3_emit_active_list", "hierarchical refine to the level-k voxel list: write 12 B x A_k", 12.0 * A, st("col_", "refine_", "create_field")),
            ("P4_triangles", "marching cubes + projection + orientation: read 12 B x A_k, write 72 B x T", 12.0 * A + 72.0 * T, st("tile_dedupe", "tile_prefix", "project", "flip_claim", "tile_first")),
            ("P5_weld", "first-seen weld: read 72 B x T, write 24 B x V + 12 B x T", 72.0 * T + 24.0 * V + 12.0 * T, st("weld", "tile_first"))):
        if us > 0:
            gbps = nbytes / (us * 1e-6) / 1e9
            passes[key] = {"what": what, "bytes": nbytes, "stage_us": round(us, 2), "achieved": round(gbps, 1), "unit": "GB/s", "peak": HBM_PEAK_GBPS,
                           "frac": round(gbps / HBM_PEAK_GBPS, 4)}
    out["hbm_passes"] = {"what": "SURVEY 8(d) algorithmic bytes of the hierarchical passes / live stage time (HIP events between the stages; k_tile_first "
                                 "counts in both P4 and P5: it holds the orientation test and the first-seen pass); reported, not claimed", **passes}
    evals = {"project": 13 * rounds if rounds else None, "flip": 12 * int(info.triangles)}
    for key, stage_prefix in (("project", "project"), ("flip", "flip_claim"), ("flip", "tile_first_flip")):
        name = next((s for s in stages if s.startswith(stage_prefix)), None)
        if name is None:
            continue
        us = proj_span_us if key == "project" else stages[name]
        k = pmc.get(name, {})
        wi = k.get("wave_instr")
        row = {"kernel": name, "avg_us": round(us, 2), "avg_us_from": ("in-kernel s_memrealtime span (first wave start to last wave end)" if key == "project"
                                                                      else "HIP events around the stage (adds ~2-3 us of marker latency)"),
               "trace_avg_us": k.get("avg_us"), "wave_instr": wi,
               "vgpr": k.get("vgpr"), "waves_per_simd": k.get("waves_per_simd")}
        if wi:
            ach = wi / (us * 1e-6)
            row.update({"achieved": round(ach, 1), "frac": round(ach / VALU_PEAK_WAVE_INSTR_PER_S, 4)})
            mc = ((icost.get("kernels") or {}).get(name) or {}).get("mean_cycles_per_instr")
            if mc:
                bound_us = wi * mc / (1024.0 * clock * 1e9) * 1e6
                row.update({"mean_issue_cycles_per_instr": mc, "issue_bound_us": round(bound_us, 2), "frac_of_issue_bound": round(bound_us / us, 4)})
        if evals[key]:
            row.update({"sdf_evals": int(evals[key]), "sdf_evals_per_s": round(evals[key] / (us * 1e-6), 1)})
        out[key] = row
    return out


class StdoutGuard:
    """The contract is ONE JSON line on stdout.  Libraries below us write there too (RCCL prints a version banner to fd 1
    when its first communicator is made), so fd 1 points at stderr for the whole run and only `emit` reaches the real one."""

    def __init__(self):
        sys.stdout.flush()
        self.real = os.dup(1)
        os.dup2(2, 1)

    def emit(self, line):
        sys.stdout.flush()
        os.write(self.real, (line + "\n").encode())


def main():
    # (before anything initialises the GPU runtime: the host driver only supports dmabuf IPC, RCCL needs this for multi-process work)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    guard = StdoutGuard()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--scene", default="sdobj")
    ap.add_argument("--levels", type=int, default=4)          # N = 32 * 2^levels = 512
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ref-abi", action="store_true", help="skip the leg that times the reference-ABI code object's kernels")
    ap.add_argument("--no-anim", action="store_true")
    ap.add_argument("--no-pipelined", action="store_true")
    ap.add_argument("--no-sizes", action="store_true")
    ap.add_argument("--no-replicas", action="store_true")
    ap.add_argument("--max-levels", type=int, default=7, help="largest one-mesh size of the N > 1 legs: 32 * 2^L (7 = 4096^3)")
    ap.add_argument("--graph", dest="graph", action="store_true",
                    help="drive the single-GPU step by replaying the planned chain as a hipGraph (default: eager launches -- on "
                         "the driver's box the graph executor costs ~4 us more on the device than it saves on the host; the "
                         "other mode is reported beside it either way, and BASELINE configs[4] is reported with the graph)")
    ap.set_defaults(graph=False)
    ap.add_argument("--one-mesh-selftest", action="store_true", help="run the N > 1 one-mesh legs on 1 GPU (degenerate, for testing)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 ranks that all use GPU 0 (process group gloo, the library's collectives host-staged instead of RCCL, which refuses "
                         "duplicate devices): runs every line of the N > 1 path for testing; its timings mean nothing and the line says so")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py: --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible -- the product path has no CPU fallback")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    job = Job(world, rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        job.arm_watchdog(180, "process group initialisation")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        job.disarm_watchdog()

    pkg = sdmesh_pkg.load()
    params = [0.25] if args.scene == "anim" else None
    n_grid = 32 << args.levels
    consts = load_constants()
    multi = world > 1 or args.one_mesh_selftest

    if multi:
        # ---------------------------------------------------------------- N > 1: the step is ONE mesh over all ranks
        # 256^3 (one level below the workload) ... 4096^3: the sizes the north star names and the ones where one mesh can scale
        levels_list = [lv for lv in range(max(0, args.levels - 1), max(args.levels, args.max_levels) + 1)]
        job.arm_watchdog(900, "one-mesh legs (z-slab RCCL gather / x-slab distributed weld)")
        primary, one_mesh = one_mesh_legs(pkg, job, local_rank, args.scene, params, levels_list, args.levels, args.steps, args.warmup, args.graph,
                                          rehearse=args.rehearse_on_one_gpu)
        job.disarm_watchdog()
        dt, k, tris, verts, form = primary
        out = {
            "metric": "Mvoxels/s (512^3 SDF mesh extract: N^3 grid cells per full regen, whole job)",
            "value": round(n_grid ** 3 * k / dt / 1e6, 1), "unit": "Mvoxels/s", "n_gpus": world, "steps": k, "warmup": args.warmup,
            "ms_per_step": round(dt / k * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.scene} {n_grid}^3 (BASELINE configs[2]: reference scene sd_obj, 32^3 level-0 cells refined "
                                   f"x{args.levels}) as ONE mesh over {world} GPUs, the form the north star names: level-0 z-slabs, RCCL gather of "
                                   f"the slabs' vertex / index buffers to rank 0, whole indexed mesh there ({form}); the x-slab form that leaves "
                                   f"the mesh distributed is in one_mesh beside it, each with its same-size 1-GPU time",
                       "levels": args.levels, "grid": n_grid, "parallelism": f"{world} slabs of equal surface content, one per GPU",
                       "one_mesh_form": form},
            "mtris_per_s": round(tris * k / dt / 1e6, 2), "triangles": tris, "vertices": verts,
            "ranks_seen": one_mesh["ranks_seen"], "one_mesh": one_mesh,
        }
        if args.rehearse_on_one_gpu:
            out["rehearsal"] = "all ranks on ONE GPU over a host-staged transport: a test of the N > 1 code path, NOT a measurement"
        job.arm_watchdog(300, "sharded dense P1 leg")
        out["dense_p1_sharded"] = dense_p1_sharded_leg(pkg, job, local_rank, args.scene, params)
        job.disarm_watchdog()
        if not args.no_replicas:
            # extra: N independent regens, one per GPU, no exchange (objects / frames shard trivially) -- not the judged figure
            job.arm_watchdog(300, "replicas leg")
            h = pkg.CudaHandler(local_rank, args.scene, params)
            h.regen(args.levels)
            h.set_option("graph", 1 if args.graph else 0)
            rdt, rinfo = job.timed(lambda: h.regen(args.levels), args.steps, args.warmup)
            h.close()
            job.disarm_watchdog()
            out["replicas"] = {"what": f"{world} independent regens of the workload, one per GPU, no data-path exchange (weak scaling of "
                                       "independent objects / frames); an extra, not the scaling result",
                               "ms_per_step": round(rdt / args.steps * 1e3, 4),
                               "mvoxels_per_s": round(n_grid ** 3 * args.steps * world / rdt / 1e6, 1),
                               "mtris_per_s": round(int(rinfo.triangles) * args.steps * world / rdt / 1e6, 2)}
        if rank == 0:
            guard.emit(json.dumps(out))
        if world > 1:
            job.arm_watchdog(120, "final barrier")
            dist.barrier()
            dist.destroy_process_group()
            job.disarm_watchdog()
        return

    # -------------------------------------------------------------------- N = 1
    def device_ms(hh, levels, n=60):
        """Median HIP-event time of the device chain: a separate pass with option "timing" on (the event pair costs ~9 us of
        wall time per regen, so the timed steps run without it)."""
        hh.set_option("timing", 1)
        ms = [hh.regen(levels).device_ms for _ in range(n + 5)][5:]
        hh.set_option("timing", 0)
        return round(float(np.median(ms)), 4)

    h = pkg.CudaHandler(local_rank, args.scene, params)
    info = h.regen(args.levels)                      # capacity discovery pass (synchronous, untimed)
    h.set_option("graph", 1 if args.graph else 0)
    dt, info = job.timed(lambda: h.regen(args.levels), args.steps, args.warmup)
    ms_per_step = dt / args.steps * 1e3
    out = {
        "metric": "Mvoxels/s (512^3 SDF mesh extract: N^3 grid cells per full regen, whole job)",
        "value": round(n_grid ** 3 * args.steps / dt / 1e6, 1), "unit": "Mvoxels/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.scene} {n_grid}^3 (BASELINE configs[2]: reference scene sd_obj, 32^3 level-0 "
                               f"cells refined x{args.levels}), device-resident regen, indexed mesh left in HBM, "
                               + ("hipGraph replay" if args.graph else "eager launches"),
                   "levels": args.levels, "grid": n_grid, "parallelism": "1 GPU"},
        "mtris_per_s": round(info.triangles * args.steps / dt / 1e6, 2),
        "triangles": int(info.triangles), "vertices": int(info.vertices), "active_voxels": int(info.voxel_count[args.levels]),
        "device_ms_per_step": device_ms(h, args.levels),
        "work_actually_done": {"active_gvoxels_per_s": round(int(info.voxel_count[args.levels]) / ms_per_step / 1e6, 3),
                               "gtris_per_s": round(info.triangles / ms_per_step / 1e6, 3),
                               "note": "`value` is the N^3-equivalent rate BASELINE defines; the hierarchical path touches O(surface) cells"},
    }

    # --- the same step driven the other way (graph replay vs eager launches of the identical chain)
    h.set_option("graph", 0 if args.graph else 1)
    gdt, _ = job.timed(lambda: h.regen(args.levels), 100, 10)
    out["eager_launches" if args.graph else "hipgraph_replay"] = {
        "ms_per_step": round(gdt / 100 * 1e3, 4), "device_ms_per_step": device_ms(h, args.levels),
        "mvoxels_per_s": round(n_grid ** 3 * 100 / gdt / 1e6, 1)}
    h.set_option("graph", 1 if args.graph else 0)

    # --- the other grid sizes the north star names (256^3 / 1024^3 of the workload scene, and BASELINE configs[1]: the
    #     single sphere at 256^3); same measurement as the primary metric, fewer steps
    if not args.no_sizes:
        sizes = {}
        for label, sc, lv in ((f"{args.scene}_256", args.scene, 3), (f"{args.scene}_1024", args.scene, 5), ("sphere_256", "sphere", 3)):
            if sc == args.scene and lv == args.levels:
                continue
            hz = pkg.CudaHandler(local_rank, sc, params if sc == args.scene else None)
            hz.regen(lv)
            hz.set_option("graph", 1 if args.graph else 0)
            dz, zi = job.timed(lambda: hz.regen(lv), 50, 5)
            dz /= 50
            ng = 32 << lv
            sizes[label] = {"grid": ng, "ms_per_step": round(dz * 1e3, 4), "device_ms_per_step": device_ms(hz, lv, 30),
                            "mvoxels_per_s": round(ng ** 3 / dz / 1e6, 1), "mtris_per_s": round(zi.triangles / dz / 1e6, 1),
                            "triangles": int(zi.triangles), "vertices": int(zi.vertices)}
            hz.close()
        out["sizes"] = sizes

    # --- several regens in flight (one handle each): most kernels of the chain are latency-bound, so they overlap
    if not args.no_pipelined:
        depth = 3          # (measured in round 3, MEASUREMENTS.md: 1 / 2 / 3 / 4 / 6 / 8 in flight = 0.129 / 0.083 / 0.070 / 0.085 / 0.087 / 0.088 ms per regen at 512^3)
        hs = [pkg.CudaHandler(local_rank, args.scene, params) for _ in range(depth)]
        for hh in hs:
            hh.regen(args.levels); hh.set_option("graph", 1 if args.graph else 0); hh.regen(args.levels); hh.regen(args.levels)
        torch.cuda.synchronize()
        psteps = max(args.steps, 60)
        tp = time.perf_counter()
        for hh in hs:
            hh.regen_begin(args.levels)
        for i in range(psteps):
            hh = hs[i % depth]
            pinfo = hh.regen_end()
            if i + depth < psteps:
                hh.regen_begin(args.levels)
        torch.cuda.synchronize()
        dp = time.perf_counter() - tp
        out["pipelined"] = {"what": f"{depth} regens in flight on one GPU (one handle each, sdmesh_regen_begin/_end)",
                            "ms_per_regen": round(dp / psteps * 1e3, 4), "mvoxels_per_s": round(n_grid ** 3 * psteps / dp / 1e6, 1),
                            "mtris_per_s": round(pinfo.triangles * psteps / dp / 1e6, 1)}
        for hh in hs:
            hh.close()

    # --- the ray-march preview of the same scene (pure VALU, one thread per pixel), 1080p
    tex = torch.empty((1080, 1920, 4), dtype=torch.uint8, device="cuda")
    cam = ((0.0, 1.0, -6.0), (0.0, -0.16439898, 0.98639392), (0.0, 0.98639392, 0.16439898), (1.0, 0.0, 0.0))
    rms = [h.render(1920, 1080, *cam, 1.0, None, dev_ptr=tex.data_ptr()) for _ in range(12)][2:]
    out["render_preview"] = {"what": "sdmesh_render (compute_render.cu) of the workload scene, 1920x1080 RGBA8, kernel time",
                             "ms": round(float(np.median(rms)), 4), "mpixels_per_s": round(1920 * 1080 / float(np.median(rms)) / 1e3, 1)}
    del tex

    # --- BASELINE config 5: per-frame animated SDF (sphere centre (0, 0.5 sin t, 0), t = frame / 60), same grid
    if not args.no_anim:
        ha = pkg.CudaHandler(local_rank, "anim", [0.5])
        for cy in (0.5, -0.5, 0.0):                  # widest frames first: one capacity plan covers the animation
            ha.set_scene("anim", [cy]); ha.regen(args.levels)
        anim = {"what": "120 frames, new scene uniform + full regen per frame (BASELINE configs[4]); hipGraph replay as the "
                        "config names it, and plain eager launches of the same chain beside it; each mode is run twice (eager, graph, eager, "
                        "graph) and the better pass of a mode is the one reported, frames_in_flight_3 likewise (best of 2 passes)", "grid": n_grid}
        for mode, flag in (("eager", 0), ("hipgraph", 1), ("eager", 0), ("hipgraph", 1)):   # two passes each, the better one counts
            ha.set_option("graph", flag)
            for f in range(20):
                ha.set_animation_time(f / 60.0); ha.regen(args.levels)
            torch.cuda.synchronize()
            ta = time.perf_counter()
            atris = 0
            for f in range(120):
                ha.set_animation_time(f / 60.0)
                atris += ha.regen(args.levels).triangles
            torch.cuda.synchronize()
            da = time.perf_counter() - ta
            if mode not in anim or 120 / da > anim[mode]["frames_per_s"]:
                anim[mode] = {"frames_per_s": round(120 / da, 1), "ms_per_frame": round(da / 120 * 1e3, 4),
                              "mtris_per_s": round(atris / da / 1e6, 1)}
        anim["frames_per_s"] = anim["hipgraph"]["frames_per_s"]
        ha.close()
        # the same animation as a runtime SDF PROGRAM in time (row N3: the reference scene's two instructions, the sphere's centre moved by
        # sdmesh_set_program_time; interpreted in every kernel, so slower than the compiled scene) -- same meshes, frame for frame (tests)
        hp = pkg.CudaHandler(local_rank, "sdobj")
        hp.set_program([("box_skeleton", [0, 0, 0, 3, 1, 0.5, 0.1], "set"), ("sphere", [0, 0, 0, 1], "smooth_min", 0.5)])
        hp.set_program_motion([[0] * 7, [0, 1.0, 0, 0, 0, 0, 0]])
        for cy in (0.5, -0.5, 0.0):
            hp.set_program_time(cy); hp.regen(args.levels)
        import math
        for f in range(10):
            hp.set_program_time(0.5 * math.sin(f / 60.0)); hp.regen(args.levels)
        torch.cuda.synchronize()
        ta = time.perf_counter()
        for f in range(60):
            hp.set_program_time(0.5 * math.sin(f / 60.0))
            hp.regen(args.levels)
        torch.cuda.synchronize()
        da = time.perf_counter() - ta
        anim["as_program"] = {"frames_per_s": round(60 / da, 1), "ms_per_frame": round(da / 60 * 1e3, 4), "frames": 60}
        hp.close()
        # the same 120 frames with three frames in flight (one handle each, sdmesh_regen_begin / _end): frame f + 1 and f + 2 are
        # meshed while frame f is waited for -- throughput of a renderer that double-buffers its meshes; an extra, not `frames_per_s`
        depth = 3
        hs = [pkg.CudaHandler(local_rank, "anim", [0.5]) for _ in range(depth)]
        for hh in hs:
            for cy in (0.5, -0.5, 0.0):
                hh.set_scene("anim", [cy]); hh.regen(args.levels)
        best = None
        for _ in range(2):
            torch.cuda.synchronize()
            ta = time.perf_counter()
            atris = 0
            for f in range(min(depth, 120)):
                hs[f % depth].set_animation_time(f / 60.0); hs[f % depth].regen_begin(args.levels)
            for f in range(120):
                hh = hs[f % depth]
                atris += hh.regen_end().triangles
                if f + depth < 120:
                    hh.set_animation_time((f + depth) / 60.0); hh.regen_begin(args.levels)
            torch.cuda.synchronize()
            da = time.perf_counter() - ta
            if best is None or da < best[0]:
                best = (da, atris)
        anim["frames_in_flight_3"] = {"frames_per_s": round(120 / best[0], 1), "ms_per_frame": round(best[0] / 120 * 1e3, 4),
                                      "mtris_per_s": round(best[1] / best[0] / 1e6, 1)}
        for hh in hs:
            hh.close()
        out["animated"] = anim

    # --- the opt-in fast numeric build (libsdmesh_fast.so = the reference's own --use_fast_math mode): reported, never `value`
    hf = pkg.CudaHandler(local_rank, args.scene, params, fast_math=True)
    hf.regen(args.levels)
    fdt, finfo = job.timed(lambda: hf.regen(args.levels), 100, 10)
    out["fast_math_build"] = {"what": "same step through libsdmesh_fast.so (FMA contraction, approximate divide / sqrt: the reference's nvcc "
                                      "--use_fast_math mode); opt-in, within 1e-5 relative of the strict build, not bit-exact, never the headline",
                              "ms_per_step": round(fdt / 100 * 1e3, 4), "device_ms_per_step": device_ms(hf, args.levels, 30),
                              "mvoxels_per_s": round(n_grid ** 3 * 100 / fdt / 1e6, 1), "triangles": int(finfo.triangles), "vertices": int(finfo.vertices)}
    hf.close()

    if not args.no_ref_abi:
        ra = reference_abi_leg(pkg, local_rank, args.scene, params, args.levels)
        if "kernels_total_ms" in ra:
            ra["native_chain_device_ms"] = out["device_ms_per_step"]
            ra["native_speedup_over_reference_shaped_kernels"] = round(ra["kernels_total_ms"] / out["device_ms_per_step"], 1)
        out["reference_abi_kernels"] = ra
    out["host_api"] = host_api_leg(pkg, local_rank, args.scene, params)
    roofline_legs(h, out, n_grid, args.scene, params, consts)
    out["roofline_step"] = roofline_step_leg(pkg, local_rank, args.scene, params, args.levels, consts, info)
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.scene, args.levels, params)
    guard.emit(json.dumps(out))
    h.close()


if __name__ == "__main__":
    main()
