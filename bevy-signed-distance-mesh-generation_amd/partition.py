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
