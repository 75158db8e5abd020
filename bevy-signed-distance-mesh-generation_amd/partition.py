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

Pure index arithmetic on torch tensors (CPU or GPU): PyTorch is the plumbing for buffers and collectives.
"""
import torch

INIT_FACTOR = 32
NCOLS = INIT_FACTOR * INIT_FACTOR


def slab_bounds(part, nparts):
    """Level-0 z layers [z0, z1) of rank `part` (same formula as sdmesh_regen_part)."""
    return INIT_FACTOR * part // nparts, INIT_FACTOR * (part + 1) // nparts


def merge_destinations(counts):
    """counts: int64 [R, 1024] triangles per (rank, column).  Returns a list of R int64 tensors: for rank r the
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
    """tri_lists: R tensors [T_r, 18] float32 (compact Triangle structs) on one device; counts [R, 1024].
    Returns the merged [sum T_r, 18] tensor in single-GPU emission order."""
    dests = merge_destinations(counts.to(tri_lists[0].device))
    total = int(sum(t.shape[0] for t in tri_lists))
    out = torch.empty((total, 18), dtype=torch.float32, device=tri_lists[0].device)
    for t, d in zip(tri_lists, dests):
        if t.shape[0]:
            out.index_copy_(0, d, t)
    return out


class ShardedMesher:
    """ONE mesh across all ranks (strong scaling): rank r meshes level-0 z-slab r with the weld
    skipped, the 1024 per-column triangle counts are all-gathered (4 KB), the compact per-slab triangle lists go
    straight to the root with point-to-point send/recv (RCCL: each peer on its own xGMI link), the root restores the
    single-GPU order and welds.  The handler must run on torch's current stream (`set_stream`).

    The transport is whatever process group is current: with "nccl" (= RCCL) the device buffers are exchanged
    directly; with "gloo" (tests, CPU rehearsals) they are staged through host memory -- same code otherwise."""

    def __init__(self, pkg, handler, levels, rank, world, device):
        import torch.distributed as dist
        self.pkg, self.h, self.levels, self.rank, self.world, self.dev = pkg, handler, levels, rank, world, device
        self.dist = dist
        self.staged = world > 1 and dist.get_backend() == "gloo"
        handler.set_option("skip_weld", 1)
        info = handler.regen(levels, pkg.PART_ZSLAB, rank, world)        # capacity discovery
        handler.set_option("skip_weld", 0)
        self.cap = int(info.triangles * 1.25) + 1024
        self.mine = torch.empty((self.cap, 18), dtype=torch.float32, device=device)
        self.cc = torch.zeros(NCOLS, dtype=torch.int32, device=device)
        self.allc = torch.zeros((world, NCOLS), dtype=torch.int32, device=device)

    def _all_gather_counts(self):
        if self.world == 1:
            self.allc.copy_(self.cc.unsqueeze(0))
        elif self.staged:
            mine = self.cc.cpu()
            parts = [torch.zeros_like(mine) for _ in range(self.world)]
            self.dist.all_gather(parts, mine)
            self.allc.copy_(torch.stack(parts))
        else:
            self.dist.all_gather_into_tensor(self.allc, self.cc)

    def step(self):
        """One sharded regen.  Returns the root's sdmesh_regen_info of the welded mesh (None on the other ranks)."""
        h, dist = self.h, self.dist
        h.set_option("skip_weld", 1)
        try:
            h.regen(self.levels, self.pkg.PART_ZSLAB, self.rank, self.world)
            n = h.device_triangles_into(self.mine.data_ptr(), self.cap)
            h.column_triangle_counts_into(self.cc.data_ptr())
        finally:
            h.set_option("skip_weld", 0)
        self._all_gather_counts()
        if self.rank == 0:
            sizes = self.allc.sum(1).cpu().tolist()
            bufs, ops = [], []
            for r in range(1, self.world):
                b = torch.empty((int(sizes[r]), 18), dtype=torch.float32, device="cpu" if self.staged else self.dev)
                bufs.append(b)
                if b.numel():
                    if self.staged:
                        dist.recv(b, src=r)
                    else:
                        ops.append(dist.P2POp(dist.irecv, b, r))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            lists = [self.mine[:n]] + [b.to(self.dev) for b in bufs]
            merged = merge_triangles(lists, self.allc.to(torch.int64))
            torch.cuda.current_stream().synchronize()
            return h.weld_device_triangles(merged.data_ptr(), merged.shape[0])
        if n:
            if self.staged:
                dist.send(self.mine[:n].cpu(), dst=0)
            else:
                for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, self.mine[:n], 0)]):
                    w.wait()
        return None
