"""z-slab partition of one mesh across ranks and the order-restoring merge.

The reference is single-GPU (`CudaDevice::new(0)`, src/cuda/mod.rs:52); this module is the build's own
multi-GPU extension.  The only facts it relies on are the reference's global order (level-0 cells x outer /
y / z inner, src/cuda/mod.rs:105-122, children in slot order id*8 + i*4 + j*2 + k,
cuda/modules/compute_mesh_generation.cu:51) and the first-seen weld (src/cuda/mod.rs:263-296):

  * rank r owns the level-0 z layers [32 r / R, 32 (r+1) / R); its voxel / triangle lists are the global
    lists restricted to those cells, in the same relative order;
  * z0 is the least significant level-0 digit of the order key, so the global list is, for every (x0, y0)
    column in increasing order, rank 0's triangles of that column, then rank 1's, ...;
  * welding the merged list with the first-seen rule gives the single-GPU vertices and indices, bit for bit.

The data path is behind the C ABI (`sdmesh_sharded_regen`: RCCL + HIP kernels); this module is its thin Python caller,
the gloo-staged transport for rehearsals, and a CPU restatement of the merge for the no-GPU tests.
"""
import torch

INIT_FACTOR = 32
NCOLS = INIT_FACTOR * INIT_FACTOR


def slab_bounds(part, nparts):
    """Level-0 z layers [z0, z1) of rank `part` (same formula as sdmesh_regen_part)."""
    return INIT_FACTOR * part // nparts, INIT_FACTOR * (part + 1) // nparts


def dense_slab_planes(N, part, nparts):
    """x planes [x0, x1) of the dense (N+1)^3 lattice rank `part` samples (P1 shards with no exchange: x is the outermost index
    of the volume, so a slab of planes is a contiguous range of it).  Same formula as the level-0 slabs, over N + 1 planes."""
    return (N + 1) * part // nparts, (N + 1) * (part + 1) // nparts


def merge_destinations(counts):
    """CPU restatement of the merge plan (k_merge_plan / k_merge_copy in csrc/kernels.hip), for the tests that run without a
    GPU and as the kernel's checker -- the product path calls `sdmesh_merge_by_column`, never this.
    counts: int64 [R, 1024] triangles per (rank, column).  Returns a list of R int64 tensors: for rank r the
    position in the merged list of each of its triangles (rank lists are column-sorted already)."""
    counts = counts.to(torch.int64)
    R = counts.shape[0]
    col_total = counts.sum(0)                                     # [1024]
    col_base = torch.cumsum(col_total, 0) - col_total             # exclusive
    rank_off = torch.cumsum(counts, 0) - counts                   # [R, 1024] offset of rank r inside a column
    dests = []
    for r in range(R):
        c = counts[r]
        seg_start = torch.cumsum(c, 0) - c                        # start of column segment inside rank r's list
        base = col_base + rank_off[r]                             # merged start of that segment
        n = int(c.sum())
        col_of = torch.repeat_interleave(torch.arange(NCOLS, device=c.device), c, output_size=n)
        dests.append(base[col_of] + (torch.arange(n, device=c.device) - seg_start[col_of]))
    return dests


def merge_triangles(tri_lists, counts):
    """CPU restatement of `sdmesh_merge_by_column` (see merge_destinations).
    tri_lists: R tensors [T_r, 18] float32 (compact Triangle structs) on one device; counts [R, 1024].
    Returns the merged [sum T_r, 18] tensor in single-GPU emission order."""
    dests = merge_destinations(counts.to(tri_lists[0].device))
    total = int(sum(t.shape[0] for t in tri_lists))
    out = torch.empty((total, 18), dtype=torch.float32, device=tri_lists[0].device)
    for t, d in zip(tri_lists, dests):
        if t.shape[0]:
            out.index_copy_(0, d, t)
    return out


def indexed_merge_reference(parts, counts, z_faces, band):
    """CPU restatement (numpy) of the root's share of the INDEXED z-slab gather (k_zs_* in csrc/zslab.hip), for the tests that
    run without a GPU -- the product path calls `sdmesh_sharded_regen`, never this.

    parts: per rank (verts [V_r, 6] f32 = position | normal in the rank's own first-seen order, idx [T_r, 3] local indices in the
    rank's emission order); counts [R, 1024] triangles per level-0 column; z_faces: per rank (z_lo, z_hi) of its slab; band: half
    width of the face bands.  Returns (verts, idx) of the single-GPU mesh."""
    import numpy as np
    R = len(parts)
    counts = np.asarray(counts, dtype=np.int64).reshape(R, NCOLS)
    tstart = np.concatenate([np.zeros((R, 1), np.int64), np.cumsum(counts, 1)], 1)        # [R, 1025] segment starts in the rank's list
    # (1) local vertices first seen per (rank, column): prefix maximum of the segments' largest index + 1
    vstart = np.zeros((R, NCOLS + 1), np.int64)
    for r, (v, t) in enumerate(parts):
        run = 0
        for c in range(NCOLS):
            seg = t[tstart[r, c]:tstart[r, c + 1]]
            if len(seg):
                run = max(run, int(seg.max()) + 1)
            vstart[r, c + 1] = run
        assert run == len(v), "an unreferenced vertex"
    segof = [np.searchsorted(vstart[r, 1:], np.arange(len(parts[r][0])), side="right") for r in range(R)]   # first-seen column
    # (2) face-band pairs by weld key; of a pair the copy whose first-seen (column, rank) comes first stays
    key_of = lambda p: np.round(p.astype(np.float32) * np.float32(10e4)).astype(np.int64)      # noqa: E731  (src/cuda/mod.rs:268)
    redirect = [np.full(len(parts[r][0]), -1, np.int64) for r in range(R)]                    # -1 = kept, else (rank << 32 | vertex)
    seen = {}
    for r, (v, t) in enumerate(parts):
        z = v[:, 2]
        lo, hi = z_faces[r]
        near = ((hi < 2.5) & (z >= hi - band)) | ((lo > -2.5) & (z <= lo + band))
        for i in np.nonzero(near)[0]:
            k = tuple(key_of(v[i, :3]))
            if k not in seen:
                seen[k] = (r, int(i))
                continue
            ra, ia = seen[k]
            a_first = (segof[ra][ia], ra) <= (segof[r][i], r)
            if a_first:
                redirect[r][i] = (ra << 32) | ia
            else:
                redirect[ra][ia] = (r << 32) | int(i)
                seen[k] = (r, int(i))
    # (3) kept vertices numbered segment by segment in (column, rank) order, local order inside a segment
    gid = [np.full(len(parts[r][0]), -1, np.int64) for r in range(R)]
    out_v, n = [], 0
    for c in range(NCOLS):
        for r in range(R):
            a, b = vstart[r, c], vstart[r, c + 1]
            keep = np.nonzero(redirect[r][a:b] < 0)[0] + a
            gid[r][keep] = n + np.arange(len(keep))
            out_v.append(parts[r][0][keep])
            n += len(keep)
    for r in range(R):
        for i in np.nonzero(redirect[r] >= 0)[0]:
            w = int(redirect[r][i])
            gid[r][i] = gid[w >> 32][w & 0xFFFFFFFF]
    # (4) index buffers, segment by segment, into the merged triangle order
    out_t = []
    for c in range(NCOLS):
        for r in range(R):
            seg = parts[r][1][tstart[r, c]:tstart[r, c + 1]]
            if len(seg):
                out_t.append(gid[r][seg])
    verts = np.concatenate(out_v) if out_v else np.zeros((0, 6), np.float32)
    idx = np.concatenate(out_t) if out_t else np.zeros((0, 3), np.int64)
    return verts, idx


class ShardedMesher:
    """ONE mesh across all ranks (strong scaling) -- a thin caller of the C ABI.

    With RCCL available to the job (process group backend "nccl", or a single rank) the whole step is ONE library call,
    `sdmesh_sharded_regen`, by default in its INDEXED form: z-slab regen AND weld per rank, ncclAllGather of the 1024 per-column
    triangle counts and first-seen vertex counts (+ sizes and a status word), grouped ncclSend / ncclRecv of the slabs' vertex and
    index buffers to the root (24 B per vertex + 12 B per triangle, each peer on its own xGMI link), and on the root the merge of
    those buffers by (column, rank) segments -- no second weld (csrc/zslab.hip); in steady state the step is enqueued blind from a
    collective plan and synchronises once.  Option "shard_indexed" = 0 selects the raw form (unwelded slabs, 72 B per triangle,
    order-restoring merge kernel + first-seen weld of every corner on the root).  The communicator is the library's own
    (`sdmesh_comm_init`); torch.distributed only carries the 128-byte unique id to the ranks.

    Under "gloo" (the rehearsals that put several ranks on one GPU, where RCCL refuses duplicate devices) only the TRANSPORT
    differs: rows and buffers are staged through host memory by torch.distributed, everything else is the same C ABI and the same
    kernels -- indexed form: `sdmesh_zslab_local` per rank, `sdmesh_zslab_merge` on the root; raw form (`indexed=False`): slab regen
    without weld, column counts, `sdmesh_merge_by_column`, `sdmesh_weld_device_triangles`."""

    def __init__(self, pkg, handler, levels, rank, world, device, transport=None, balance=True, indexed=True):
        import torch.distributed as dist
        self.pkg, self.h, self.levels, self.rank, self.world, self.dev = pkg, handler, levels, rank, world, device
        self.dist = dist
        self.indexed = indexed
        # slabs of equal surface content instead of equal thickness (every rank derives the same table from the scene)
        self.bounds = handler.balance_slabs(pkg.PART_ZSLAB, world, min(levels, 3)) if (balance and world > 1) else None
        if transport is None:
            transport = "staged" if (world > 1 and dist.get_backend() == "gloo") else "rccl"
        self.transport = transport
        # "rccl": sdmesh_sharded_regen over the library's RCCL communicator; "collectives": the same C++ step over the caller's collectives
        # (StagedTransport: host staging through this process group -- several ranks on one GPU); "staged": the protocol in Python, below
        self.in_library = transport in ("rccl", "collectives")
        if self.in_library:
            handler.ensure_comm(dist, world, rank, staged_device=device if transport == "collectives" else None)
            self.ranks_seen = handler.comm_ranks()
            if self.ranks_seen != world:
                raise RuntimeError(f"RCCL communicator saw {self.ranks_seen} ranks, expected {world}")
        else:
            self.ranks_seen = world
            self.cap = 0
            self.mine = None
            self.cc = torch.zeros(NCOLS, dtype=torch.int32, device=device)
            self.allc = torch.zeros((world, NCOLS), dtype=torch.int32, device=device)
            self.merged = None
            self.row = torch.zeros(handler.ZSLAB_ROW_WORDS, dtype=torch.int32, device=device)

    def close(self):
        if self.in_library and self.h.comm is not None:
            self.h.comm_destroy()

    def step(self):
        """One sharded regen.  Returns the root's sdmesh_regen_info of the welded mesh (None on the other ranks)."""
        if self.in_library:
            info = self.h.sharded_regen(self.levels, 0)
            return info if self.rank == 0 else None
        return self._step_staged_indexed() if self.indexed else self._step_staged()

    # -- gloo rehearsal, indexed form: the product's kernels on every rank and on the root, host-staged transport -------------------
    def _step_staged_indexed(self):
        h, dist, R = self.h, self.dist, self.world
        h.zslab_local(self.levels, self.rank, R, self.row.data_ptr())
        verts, tris = h.device_mesh_tensors(self.dev)                 # this rank's welded slab (views of the handle's buffers)
        rows = [torch.zeros(h.ZSLAB_ROW_WORDS, dtype=torch.int32) for _ in range(R)]
        if R > 1:
            dist.all_gather(rows, self.row.cpu())
        else:
            rows = [self.row.cpu()]
        if self.rank != 0:
            if verts.shape[0]:
                dist.send(verts.cpu(), dst=0)
            if tris.shape[0]:
                dist.send(tris.cpu(), dst=0)
            return None
        allrows = torch.stack(rows).to(self.dev)
        vs, ts = [verts], [tris]
        for r in range(1, R):
            nt, nv = int(rows[r][2048 + 1]), int(rows[r][2048 + 4])          # the record: triangles, vertices (kernels.hpp ZsRec)
            bv, bt = torch.empty((nv, 6), dtype=torch.float32), torch.empty((nt, 3), dtype=torch.int32)
            if nv:
                dist.recv(bv, src=r)
            if nt:
                dist.recv(bt, src=r)
            vs.append(bv.to(self.dev)); ts.append(bt.to(self.dev))
        torch.cuda.current_stream().synchronize()                         # the staged copies above ran on torch's stream
        info = h.zslab_merge(self.levels, allrows.data_ptr(), [v.data_ptr() if v.numel() else 0 for v in vs],
                             [t.data_ptr() if t.numel() else 0 for t in ts])
        return info

    # -- gloo rehearsal, raw form: same kernels, host-staged transport ------------------------------------------------------------------
    def _step_staged(self):
        h, dist = self.h, self.dist
        h.set_option("skip_weld", 1)
        try:
            info = h.regen(self.levels, self.pkg.PART_ZSLAB, self.rank, self.world)
            if self.mine is None or info.triangles > self.cap:           # grow before any collective is entered
                self.cap = int(info.triangles * 1.25) + 1024
                self.mine = torch.empty((self.cap, 18), dtype=torch.float32, device=self.dev)
            n = h.device_triangles_into(self.mine.data_ptr(), self.cap)
            h.column_triangle_counts_into(self.cc.data_ptr())
        finally:
            h.set_option("skip_weld", 0)
        if self.world == 1:
            self.allc.copy_(self.cc.unsqueeze(0))
        else:
            mine = self.cc.cpu()
            parts = [torch.zeros_like(mine) for _ in range(self.world)]
            dist.all_gather(parts, mine)
            self.allc.copy_(torch.stack(parts))
        if self.rank != 0:
            if n:
                dist.send(self.mine[:n].cpu(), dst=0)
            return None
        sizes = self.allc.sum(1).cpu().tolist()
        lists = [self.mine]
        for r in range(1, self.world):
            b = torch.empty((int(sizes[r]), 18), dtype=torch.float32)
            if b.numel():
                dist.recv(b, src=r)
            lists.append(b.to(self.dev) if b.numel() else torch.empty((1, 18), dtype=torch.float32, device=self.dev))
        total = int(sum(sizes))
        if self.merged is None or self.merged.shape[0] < max(total, 1):
            self.merged = torch.empty((int(total * 1.25) + 1024, 18), dtype=torch.float32, device=self.dev)
        torch.cuda.current_stream().synchronize()                         # the staged copies above ran on torch's stream
        got = h.merge_by_column([t.data_ptr() for t in lists], self.allc.data_ptr(), self.merged.data_ptr(), self.merged.shape[0])
        assert got == total
        return h.weld_device_triangles(self.merged.data_ptr(), total)


def weld_keys(pos):
    """int64 weld keys of float32 positions [n, 3], exactly as the device weld forms them: q = roundf(x * 1e5f) per
    axis (round half away from zero, src/cuda/mod.rs:270), 21 bits per axis."""
    v = (pos * torch.tensor(100000.0, dtype=torch.float32, device=pos.device)).to(torch.float64)
    q = (torch.sign(v) * torch.floor(torch.abs(v) + 0.5)).to(torch.int64) + 1048576
    return q[:, 0] | (q[:, 1] << 21) | (q[:, 2] << 42)


class DistributedMesher:
    """ONE mesh across all ranks with a DISTRIBUTED weld (x-slab partition).

    x is the outermost digit of the reference's order, so rank r's voxel and triangle lists are contiguous ranges of
    the single-GPU lists and the global first-seen vertex order is: rank 0's vertices in its local first-seen order,
    then rank 1's vertices that rank 0 does not already hold, and so on.  Every rank therefore runs the complete local
    pipeline (its own weld included); the only exchange is with the x-neighbours, about the vertices within one voxel
    of the shared face:
      1. rank r sends the weld keys of its upper-face band to rank r + 1, which matches its lower-face band against them
         (equal key = the same welded vertex in the single-GPU mesh: it keeps the lower rank's copy);
      2. the counts of kept vertices are all-gathered (global offsets), rank r sends the new numbers of its upper band
         to rank r + 1, and every rank rewrites its index buffer with the global numbers;
      3. the mesh stays distributed (`step` returns this rank's kept vertices and global-index triangles); `gather`
         concatenates the parts on the root -- no root-side weld, nothing proportional to the whole mesh is re-hashed.
    Transport as in ShardedMesher: the current process group ("nccl" direct, "gloo" staged through host memory)."""

    def __init__(self, pkg, handler, levels, rank, world, device, balance=True, bounds=None, collectives=False):
        import torch.distributed as dist
        self.pkg, self.h, self.levels, self.rank, self.world, self.dev, self.dist = pkg, handler, levels, rank, world, device, dist
        self.staged = world > 1 and dist.get_backend() == "gloo"
        self.collectives = collectives           # sdmesh_distributed_regen over StagedTransport instead of RCCL (several ranks on one GPU)
        size = torch.tensor(5.0, dtype=torch.float32) / torch.tensor(float(INIT_FACTOR << levels), dtype=torch.float32)
        if bounds is None and balance and handler is not None and world > 1:
            # slabs of equal surface content instead of equal thickness (every rank derives the same table from the scene)
            bounds = handler.balance_slabs(pkg.PART_XSLAB, world, min(levels, 3))
        self.bounds = bounds
        self.native = handler is not None and (world == 1 or dist.get_backend() != "gloo" or collectives)
        lo, hi = (bounds[rank], bounds[rank + 1]) if bounds is not None else slab_bounds(rank, world)
        self.band = 2.5 * float(size)      # a projected vertex stays within ~a voxel diagonal of its lattice edge
        self.x_lo = -2.5 + 5.0 * lo / INIT_FACTOR
        self.x_hi = -2.5 + 5.0 * hi / INIT_FACTOR

    # -- point-to-point helpers: variable-length vectors (length first, then the payload in its own dtype) ----------
    def _send(self, t, dst):
        t = t.contiguous().flatten()
        n = torch.tensor([t.numel()], dtype=torch.int64, device="cpu" if self.staged else self.dev)
        self.dist.send(n, dst=dst)
        if t.numel():
            self.dist.send(t.cpu() if self.staged else t, dst=dst)

    def _recv(self, src, dtype=torch.int64):
        n = torch.zeros(1, dtype=torch.int64, device="cpu" if self.staged else self.dev)
        self.dist.recv(n, src=src)
        t = torch.empty(int(n.item()), dtype=dtype, device="cpu" if self.staged else self.dev)
        if t.numel():
            self.dist.recv(t, src=src)
        return t.to(self.dev)

    def step(self):
        """One distributed regen.  Returns (vertices [v, 6] float32 kept by this rank, triangles [t, 3] int32 with GLOBAL
        vertex indices, global offset of this rank's vertices).  Concatenated over the ranks in order they are the
        single-GPU mesh, bit for bit.

        With RCCL available (backend "nccl", or one rank) the whole step is `sdmesh_distributed_regen`: HIP kernels + two
        neighbour messages + two 4-byte all-gathers behind the C ABI, no torch arithmetic.  Under "gloo" (several ranks
        rehearsed on one GPU) the same face exchange runs as torch index arithmetic with a host-staged transport
        (`exchange` below, also what the no-GPU tests drive with the oracle standing in for the kernels)."""
        if self.native:
            self.h.ensure_comm(self.dist, self.world, self.rank, staged_device=self.dev if self.collectives else None)
            _, first = self.h.distributed_regen(self.levels)
            verts, tris = self.h.device_mesh_tensors(self.dev)
            return verts, tris, first
        self.h.regen(self.levels, self.pkg.PART_XSLAB, self.rank, self.world)
        verts, tris = self.h.device_mesh_tensors(self.dev)
        return self.exchange(verts, tris)

    def exchange(self, verts, tris):
        """The weld across the slab faces, given this rank's locally welded slab mesh (vertices [v, 6] float32, triangles
        [t, 3] with local indices) on `self.dev` -- a GPU in the product, the CPU in the gloo tests."""
        dist, r, R, dev = self.dist, self.rank, self.world, self.dev
        nv = verts.shape[0]
        x = verts[:, 0]
        up_ids = torch.nonzero(x >= self.x_hi - self.band).flatten() if r + 1 < R else torch.empty(0, dtype=torch.int64, device=dev)
        lo_ids = torch.nonzero(x <= self.x_lo + self.band).flatten() if r > 0 else torch.empty(0, dtype=torch.int64, device=dev)
        # 1. keys of my upper band -> rank r + 1; match my lower band against rank r - 1's upper band
        if r + 1 < R:
            self._send(weld_keys(verts[up_ids, :3]), r + 1)
        dup_pos = torch.full((nv,), -1, dtype=torch.int64, device=dev)      # position in the neighbour's band, or -1
        if r > 0:
            prev_keys = self._recv(r - 1)
            if prev_keys.numel() and lo_ids.numel():
                order = torch.argsort(prev_keys)
                sk = prev_keys[order]
                mine = weld_keys(verts[lo_ids, :3])
                pos = torch.searchsorted(sk, mine).clamp_(max=sk.numel() - 1)
                hit = sk[pos] == mine
                dup_pos[lo_ids[hit]] = order[pos[hit]]
        keep = dup_pos < 0
        newidx = torch.cumsum(keep.to(torch.int32), 0, dtype=torch.int32) - 1
        n_keep = keep.sum().reshape(1)
        # 2. global offsets; my upper band's new numbers -> rank r + 1
        if R > 1:
            counts = [torch.zeros(1, dtype=torch.int64, device="cpu" if self.staged else dev) for _ in range(R)]
            dist.all_gather(counts, n_keep.cpu() if self.staged else n_keep)
            counts = torch.cat(counts).to(dev)
        else:
            counts = n_keep
        offsets = torch.cumsum(counts, 0) - counts
        if r == 0:
            # rank 0 keeps every vertex at its local number: nothing to rewrite, nothing to copy
            if R > 1:
                self._send(up_ids.to(torch.int64), 1)
            return verts, tris.to(torch.int32), offsets[0]
        gidx = offsets[r].to(torch.int32) + newidx
        if r + 1 < R:
            # (an upper-band vertex is never itself a duplicate: the two faces of a slab are at least one level-0 cell apart)
            self._send(gidx[up_ids].to(torch.int64), r + 1)
        prev_g = self._recv(r - 1)
        dups = torch.nonzero(~keep).flatten()
        if dups.numel():
            gidx[dups] = prev_g[dup_pos[dups]].to(torch.int32)
        out_tris = torch.index_select(gidx, 0, tris.reshape(-1).to(torch.int32)).reshape(-1, 3)
        return verts[keep], out_tris, offsets[r]

    def close(self):
        if self.native and self.h.comm is not None:
            self.h.comm_destroy()

    def gather(self, part):
        """Concatenate the ranks' parts on rank 0: (vertices [V, 6], triangles [T, 3] int64) there, None elsewhere."""
        verts, tris, _ = part
        dist, R = self.dist, self.world
        if R == 1:
            return verts, tris
        if self.rank != 0:
            self._send(verts, 0)
            self._send(tris, 0)
            return None
        vs, ts = [verts], [tris]
        for src in range(1, R):
            vs.append(self._recv(src, torch.float32).reshape(-1, 6))
            ts.append(self._recv(src, torch.int32).reshape(-1, 3))
        return torch.cat(vs), torch.cat(ts)


class _RawDevice:
    """`nbytes` of device memory at `ptr` for torch.as_tensor (no copy, no ownership)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


class StagedTransport:
    """An sdmesh_transport (include/sdmesh.h, CudaHandler.comm_init_transport) over a torch.distributed group with HOST staging: every
    operation synchronises the step's stream, moves the bytes through pageable host memory and the group's own backend (gloo), and copies
    them back before it returns.  It is what lets sdmesh_sharded_regen / sdmesh_distributed_regen -- the very C++ steps that run over RCCL --
    run between processes that share one GPU, where RCCL refuses duplicate devices: tests use it; a real job uses RCCL (comm_init)."""

    def __init__(self, dist, device, group=None):
        self.dist, self.device, self.group = dist, device, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.queue = None
        self.calls = {"all_gather": 0, "all_reduce_u32": 0, "send": 0, "recv": 0, "groups": 0, "bytes_sent": 0}

    def _view(self, ptr, nbytes):
        return torch.as_tensor(_RawDevice(ptr, nbytes), device=self.device)

    def _stream(self, stream):
        return torch.cuda.ExternalStream(stream, device=self.device) if stream else torch.cuda.default_stream(self.device)

    def _down(self, ptr, nbytes, stream):
        st = self._stream(stream)
        with torch.cuda.stream(st):
            host = self._view(ptr, nbytes).cpu()
        st.synchronize()
        return host

    def _up(self, ptr, host, stream):
        st = self._stream(stream)
        with torch.cuda.stream(st):
            self._view(ptr, host.numel()).copy_(host)
        st.synchronize()

    def all_gather(self, send, recv, nbytes, stream):
        self.calls["all_gather"] += 1
        mine = self._down(send, nbytes, stream)
        parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(self.world)]
        self.dist.all_gather(parts, mine, group=self.group)
        self._up(recv, torch.cat(parts), stream)

    def all_reduce_u32(self, send, recv, count, op, stream):
        self.calls["all_reduce_u32"] += 1
        v = self._down(send, 4 * count, stream).view(torch.int32).to(torch.int64) & 0xFFFFFFFF
        self.dist.all_reduce(v, op=self.dist.ReduceOp.MAX if op == 1 else self.dist.ReduceOp.SUM, group=self.group)
        self._up(recv, (v & 0xFFFFFFFF).to(torch.int32).view(torch.uint8), stream)

    def group_start(self):
        if self.queue is not None:
            raise RuntimeError("nested group")
        self.queue = []

    def group_end(self):
        q, self.queue = self.queue, None
        self.calls["groups"] += 1
        self._run(q)

    def send(self, buf, nbytes, peer, stream):
        self.calls["send"] += 1
        self.calls["bytes_sent"] += nbytes
        op = ("send", buf, nbytes, peer, stream)
        if self.queue is None:
            self._run([op])
        else:
            self.queue.append(op)

    def recv(self, buf, nbytes, peer, stream):
        self.calls["recv"] += 1
        op = ("recv", buf, nbytes, peer, stream)
        if self.queue is None:
            self._run([op])
        else:
            self.queue.append(op)

    def _run(self, ops):
        reqs, landed, sent = [], [], []
        mine = [op for op in ops if op[3] == self.rank]         # a rank that is its own peer (sdmesh_comm_selftest): sends meet receives in order
        for snd, rcv in zip([o for o in mine if o[0] == "send"], [o for o in mine if o[0] == "recv"]):
            landed.append((rcv[1], self._down(snd[1], snd[2], snd[4]), rcv[4]))
        for kind, buf, nbytes, peer, stream in ops:             # every message of a group is posted before any is waited for
            if peer == self.rank:
                continue
            if kind == "send":
                sent.append(self._down(buf, nbytes, stream))    # (kept alive until the request has completed)
                reqs.append(self.dist.isend(sent[-1], peer, group=self.group))
            else:
                host = torch.empty(nbytes, dtype=torch.uint8)
                reqs.append(self.dist.irecv(host, peer, group=self.group))
                landed.append((buf, host, stream))
        for r in reqs:
            r.wait()
        for buf, host, stream in landed:
            self._up(buf, host, stream)
