// zslab.hip -- multi-GPU, z-slabs, INDEXED gather (the north star's "RCCL gather of per-slab vertex/index buffers").
//
// The reference is single-device (src/cuda/mod.rs:52); this is the build's own extension.  Every rank meshes AND welds its own
// level-0 z-slab; the root receives the slabs' vertex and index buffers (24 B per vertex + 12 B per triangle, against 72 B per
// triangle for raw lists) and restores the single-GPU mesh from them WITHOUT welding anything again -- no hash claim per corner,
// only streaming passes over the gathered buffers:
//
//   * a rank's welded vertices are numbered in its own first-seen order (src/cuda/mod.rs:276-296), so the number of distinct
//     vertices in its first n triangles is 1 + the largest index among them: the largest index per column segment, prefix-maximised
//     over the columns, gives the number of local vertices FIRST SEEN in each level-0 column -- equivalently, the vertices first seen
//     in a column are a contiguous range of the rank's vertex buffer, whose start the rank's own weld publishes on the way (k_tile_rank,
//     k_tile_finish in kernels.hip; decoded by k_zs_rowfinish) and which rides to the root in the all-gather beside the column triangle counts;
//   * in the single-GPU order (x outer, y, z inner: src/cuda/mod.rs:105-122) a level-0 column holds rank 0's triangles, then rank
//     1's, ...: so the single-GPU first-seen vertex order is, column by column and rank by rank, each (rank, column) segment's
//     first-seen vertices in their local order -- vertex buffers merge by segments exactly like the triangle lists;
//   * the only vertices two slabs can share lie within `band` of a slab face: those go through one small hash (key -> smallest
//     slot vertex id) and look themselves up; of a pair the copy whose first-seen segment comes first in the (column, rank) order
//     is the single-GPU representative, the other one is dropped from its segment's count and redirected;
//   * a kept vertex's merged place is its segment's merged start + its position in the segment - the dropped copies in front of it
//     inside the segment (counted per block of 256 or 1024 entries: no global scan); the dropped ones inherit their partner's number; every
//     triangle finds its segment by bisection over a few segment starts and is rewritten with the merged numbers.  All of it as flat
//     streaming kernels, one or four items per thread (k_zs_vertex_out, k_zs_loser_map, k_zs_index_out).
//
// Inputs lie in SLOTS (kernels.hpp, ZsRanks): slot capacities come from the host (exact counts on the synchronous path, planned
// sizes on the planned one, where the host does not know the counts when it enqueues the kernels); the ACTUAL counts are read from
// the gathered rows on the device.  Every kernel first checks the rows against the slots (zs_rows_ok) and does nothing if a rank
// reported a failure or a count exceeds its slot: the host sees the same records and repeats the step synchronously.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "kernels.hpp"
#include "prims.hpp"

namespace sdm {

static constexpr uint32_t NCOL = MESH_GENERATION_INIT_FACTOR * MESH_GENERATION_INIT_FACTOR;     // 1024 level-0 columns
static constexpr uint32_t ZS_GAP = 0xfffffffeu;        // redirect[] of a slot entry behind the rank's last vertex
// k_zs_vertex_out / k_zs_index_out take 256 * IPT slot entries per workgroup (IPT = 1 or 4 items per thread, every load issued in front):
// a workgroup's chain of dependent steps (hints, checks, segment starts, gathers, staging) is the same whatever it carries, so large
// meshes want the big block (MEASUREMENTS.md) and small ones the many small blocks.  zr.blk = 256 * IPT, chosen by the host;
// blockdrop[] and the block hints are per that many entries.
static constexpr uint32_t ZS_FCOLS = 4;                // columns per workgroup of k_zs_front's per-vertex pass
static_assert(SHARD_REC == 2 * NCOL, "a row is 1024 triangle counts, 1024 vertex counts, the record");

__device__ __forceinline__ uint32_t row_rec(const uint32_t *__restrict__ counts, uint32_t cstride, uint32_t r, uint32_t w) {
    return counts[(size_t) r * cstride + SHARD_REC + w];
}

// every rank healthy and inside its slot?  (uniform over the grid: every workgroup evaluates the same words)  All threads must call.
__device__ __forceinline__ bool zs_rows_ok(const uint32_t *__restrict__ counts, uint32_t cstride, const ZsRanks &zr) {
    bool bad = false;
    if (threadIdx.x < zr.nranks) {
        const uint32_t r = threadIdx.x;
        bad = row_rec(counts, cstride, r, ZR_STATUS) != 0u || row_rec(counts, cstride, r, ZR_FORM) != 1u ||
              row_rec(counts, cstride, r, ZR_VERTS) > zr.vbase[r + 1u] - zr.vbase[r] || row_rec(counts, cstride, r, ZR_TRIS) > zr.tbase[r + 1u] - zr.tbase[r];
    }
    return __syncthreads_or(bad ? 1 : 0) == 0;
}

// src_start[r][c] = sum_{c' < c} counts[r][c'];  dst_start[r][c] = sum_{c' < c} sum_r' counts[r'][c'] + sum_{r' < r} counts[r'][c]
// -> plan[(r * NCOL + c) * 2 + {0, 1}]; returns the grand total.  One 1024-thread workgroup, thread = column.
__device__ __forceinline__ uint32_t zs_plan_body(const uint32_t *__restrict__ counts, uint32_t cstride, uint32_t nranks, uint32_t *__restrict__ plan,
                                                 uint32_t *__restrict__ lds) {
    const uint32_t c = threadIdx.x;
    uint32_t col_total = 0;
    for (uint32_t r = 0; r < nranks; r++) col_total += counts[(size_t) r * cstride + c];
    uint32_t total;
    const uint32_t col_base = block_excl_scan(col_total, total, lds);
    uint32_t within = 0;
    for (uint32_t r = 0; r < nranks; r++) {
        const uint32_t n = counts[(size_t) r * cstride + c];
        uint32_t rank_total;
        const uint32_t src = block_excl_scan(n, rank_total, lds);
        plan[(r * NCOL + c) * 2u] = src;
        plan[(r * NCOL + c) * 2u + 1u] = col_base + within;
        within += n;
    }
    return total;
}

// the triangle plan with the block hints of k_zs_index_out: thint[zr.tblk[r] + j] = the column that holds triangle j * zr.blk of rank r
// (so a workgroup of that kernel stages only the few segment starts its 256 triangles can fall into, not the rank's 1024)
__device__ __forceinline__ uint32_t zs_col_of(const uint32_t *__restrict__ s_start, uint32_t t);
__device__ __forceinline__ uint32_t zs_tplan_with_hints(const uint32_t *__restrict__ counts, uint32_t cstride, const ZsRanks &zr, uint32_t *__restrict__ plan,
                                                        uint32_t *__restrict__ thint, uint32_t *__restrict__ s_src /* NCOL + 1 */, uint32_t *__restrict__ lds) {
    const uint32_t c = threadIdx.x, nranks = zr.nranks;
    uint32_t col_total = 0;
    for (uint32_t r = 0; r < nranks; r++) col_total += counts[(size_t) r * cstride + c];
    uint32_t total;
    const uint32_t col_base = block_excl_scan(col_total, total, lds);
    uint32_t within = 0;
    for (uint32_t r = 0; r < nranks; r++) {
        const uint32_t n = counts[(size_t) r * cstride + c];
        uint32_t rank_total;
        const uint32_t src = block_excl_scan(n, rank_total, lds);
        plan[(r * NCOL + c) * 2u] = src;
        plan[(r * NCOL + c) * 2u + 1u] = col_base + within;
        within += n;
        __syncthreads();
        s_src[c] = src;
        __syncthreads();
        for (uint32_t j = c; j < zr.tblk[r + 1u] - zr.tblk[r]; j += 1024u) thint[zr.tblk[r] + j] = zs_col_of(s_src, j * zr.blk);
    }
    return total;
}

// ------------------------------------------------------------------------------------------
// local phase (every rank)
// ------------------------------------------------------------------------------------------
// the largest c in [0, NCOL) with s_start[c] <= t (s_start non-decreasing, s_start[0] == 0): the column whose segment holds item t
__device__ __forceinline__ uint32_t zs_col_of(const uint32_t *__restrict__ s_start, uint32_t t) {
    uint32_t lo = 0, hi = NCOL;
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (s_start[mid] <= t) lo = mid; else hi = mid;
    }
    return lo;
}

// published starts -> counts: enc[c] = 1 + the first item (triangle / welded vertex) of column c for the columns that hold voxels, 0 for
// the others; an empty column starts where the next one does (suffix minimum), `end` behind the last.  One 1024-thread workgroup, thread =
// column; s: >= 16 words of LDS.  Returns the column's count; `bad` is raised for starts that do not ascend from 0.
__device__ __forceinline__ uint32_t zs_decode_starts(uint32_t enc, uint32_t end, uint32_t *__restrict__ s, bool &bad) {
    const uint32_t c = threadIdx.x, lane = c & 63u, wave = c >> 6;
    // inclusive suffix minimum over the 1024 columns: inside the wave by shuffles, across the 16 waves through LDS
    uint32_t x = enc ? enc - 1u : U32_NONE;
#pragma unroll
    for (uint32_t o = 1; o < 64u; o <<= 1) {
        const uint32_t y = (uint32_t) __shfl_down((int) x, o, 64);
        if (lane + o < 64u) x = y < x ? y : x;
    }
    __syncthreads();                                                 // (s: the previous use is over)
    if (lane == 0) s[wave] = x;
    __syncthreads();
    uint32_t after = U32_NONE;                                       // the waves behind this one
    for (uint32_t w = wave + 1u; w < 16u; w++) after = s[w] < after ? s[w] : after;
    x = after < x ? after : x;
    // the next column's start: the next lane's value, or for the wave's last lane the minimum of the waves behind
    uint32_t nx = (uint32_t) __shfl_down((int) x, 1u, 64);
    if (lane == 63u) nx = after;
    const uint32_t start = x < end ? x : end, next = c + 1u < NCOL ? (nx < end ? nx : end) : end;
    bad = next < start || (c == 0 && start != 0u && end != 0u) || (x != U32_NONE && x > end);
    return next >= start ? next - start : 0u;
}

// One workgroup: the row the chain published -- row[c] = column starts in triangles (k_tile_dedupe), row[NCOL + c] = column starts in
// welded vertices (k_tile_finish; welded vertices are numbered in first-seen order, so the vertices first seen in a column are a
// contiguous range) -- becomes the row the merge reads: triangle counts and first-seen vertex counts per column; the checks that make
// it trustworthy; its record; and the clear of the NEXT step's row (the next chain publishes into it; the buffers alternate).
__global__ __launch_bounds__(1024) void k_zs_rowfinish(uint32_t *__restrict__ row, uint32_t *__restrict__ next_row, const uint32_t *__restrict__ ctr_tris,
                                                        const uint32_t *__restrict__ ctr_verts, const uint32_t *__restrict__ status, uint32_t host_status,
                                                        uint32_t cap_v, uint32_t cap_t, uint32_t inject, uint32_t sig) {
    __shared__ uint32_t s[NCOL + 1];
    __shared__ uint32_t lds[17];
    const uint32_t c = threadIdx.x;
    const bool failed = host_status != 0u;                             // the host already knows this rank's local phase failed: an empty row
    const uint32_t T = failed ? 0u : *ctr_tris, V = failed ? 0u : *ctr_verts;
    const uint32_t enc_t = failed ? 0u : row[c], enc_v = failed ? 0u : row[NCOL + c];
    bool bad_t, bad_v;
    const uint32_t tc = zs_decode_starts(enc_t, T, s, bad_t);
    const uint32_t vc = zs_decode_starts(enc_v, V, s, bad_v);
    const uint32_t disorder = __syncthreads_or((bad_t || bad_v || ((enc_t != 0u) != (enc_v != 0u))) ? 1 : 0);
    uint32_t tsum, vsum;
    (void) block_excl_scan(tc, tsum, lds);
    (void) block_excl_scan(vc, vsum, lds);
    row[c] = tc;
    row[NCOL + c] = vc;
    if (next_row) { next_row[c] = 0u; next_row[NCOL + c] = 0u; }
    if (c == NCOL - 1u) {
        uint32_t st = host_status;
        if (!failed) {
            // what makes the local chain's result unusable: a capacity of the plan exceeded, a field the lattice path cannot take, a
            // vertex outside the packed key range, an invariant -- and a row that does not describe the buffers
            const uint32_t dev = (status[ST_OVERFLOW] ? 1u : 0u) | (status[ST_NOT_LATTICE] ? 2u : 0u) | (status[ST_WELD_RANGE] ? 4u : 0u) |
                                 (status[ST_INTERNAL] ? 8u : 0u) | (status[ST_TILE_OVERFLOW] ? 16u : 0u) | (status[ST_NONFINITE] ? 32u : 0u) |
                                 ((tsum != T || disorder) ? 64u : 0u) | (vsum != V ? 128u : 0u);
            if (dev) st = 0x80000000u | dev;
        }
        uint32_t *rec = row + SHARD_REC;
        rec[ZR_STATUS] = st; rec[ZR_TRIS] = T; rec[ZR_CAP_V] = cap_v; rec[ZR_CAP_T] = cap_t;
        rec[ZR_VERTS] = V; rec[ZR_FORM] = 1u; rec[ZR_INJECT] = inject; rec[ZR_SIG] = sig;
    }
}

// ------------------------------------------------------------------------------------------
// the merge (root)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t zs_rank_of(const ZsRanks &zr, uint32_t i) {
    uint32_t r = 0;
    while (r + 1u < zr.nranks && i >= zr.vbase[r + 1u]) r++;
    return r;
}

// One launch in front of everything else.  Workgroup (r, g) of the first nranks * 256: the exclusive prefix of rank r's first-seen
// vertex counts (one block scan of the row), then its four columns: vstart / newcnt of the segments, every vertex of theirs
// learns its segment and starts as "kept" (the vertices of consecutive columns are one contiguous range of the slot).  The last
// group of a rank checks the row's sum and marks the slot's unused tail.  Two extra workgroups plan the triangle segments
// (resetting the merge's counters) and, with one rank, the vertex segments (nothing can be dropped).  With more than one rank the
// whole grid also clears the band hash and the per-block drop counts.
__global__ __launch_bounds__(1024) void k_zs_front(const uint32_t *__restrict__ counts, uint32_t cstride, const ZsRanks *__restrict__ zrp,
                                                    uint32_t *__restrict__ tplan, uint32_t *__restrict__ vplan, uint32_t *__restrict__ vstart,
                                                    uint32_t *__restrict__ newcnt, uint32_t *__restrict__ segof, uint32_t *__restrict__ redirect,
                                                    uint32_t *__restrict__ hclear, uint32_t hwords, uint32_t *__restrict__ blockdrop, uint32_t *__restrict__ thint,
                                                    uint32_t cap_verts, uint32_t cap_tris, uint32_t *__restrict__ zctl) {
    const ZsRanks &zr = *zrp;
    __shared__ uint32_t lds[17];
    __shared__ uint32_t s_b[ZS_FCOLS + 1];
    __shared__ uint32_t s_src[NCOL + 1];
    const uint32_t tid = threadIdx.x, R = zr.nranks, groups = R * (NCOL / ZS_FCOLS);
    const bool ok = zs_rows_ok(counts, cstride, zr);
    if (blockIdx.x == groups) {                  // the triangle segments' plan (with k_zs_index_out's block hints), the merge's counters
        uint32_t flags = ok ? 0u : 1u, T = 0;
        if (ok) {
            T = zs_tplan_with_hints(counts, cstride, zr, tplan, thint, s_src, lds);
            if (T > cap_tris) flags |= 2u;
        }
        if (tid == 0) { zctl[ZC_TRIS] = T; zctl[ZC_NBAND] = 0u; zctl[ZC_BANDFLAGS] = 0u; zctl[ZC_PLANFLAGS] = flags; if (R > 1u || !ok) zctl[ZC_VERTS] = 0u; }
        return;
    }
    if (blockIdx.x == groups + 1u) {             // one rank: nothing can be dropped, the vertex segments' plan is final now
        if (ok && R == 1u) {
            const uint32_t V = zs_plan_body(counts + NCOL, cstride, 1u, vplan, lds);
            if (tid == 0) { zctl[ZC_VERTS] = V; zctl[ZC_VFLAGS] = V > cap_verts ? 4u : 0u; }
        } else if (tid == 0) {
            zctl[ZC_VFLAGS] = 0u;
        }
        return;
    }
    if (!ok) return;
    for (uint32_t i = blockIdx.x * 1024u + tid; i < hwords; i += groups * 1024u) hclear[i] = 0xffffffffu;      // (all-ones = empty key = no vertex yet)
    if (blockdrop)
        for (uint32_t i = blockIdx.x * 1024u + tid; i < zr.vbase[R] / zr.blk + 1u; i += groups * 1024u) blockdrop[i] = 0u;
    const uint32_t r = blockIdx.x / (NCOL / ZS_FCOLS), c0 = (blockIdx.x % (NCOL / ZS_FCOLS)) * ZS_FCOLS;
    const uint32_t vc = counts[(size_t) r * cstride + NCOL + tid];
    uint32_t total;
    const uint32_t ex = block_excl_scan(vc, total, lds);
    if (tid >= c0 && tid < c0 + ZS_FCOLS) { vstart[r * NCOL + tid] = ex; newcnt[r * NCOL + tid] = vc; s_b[tid - c0] = ex; }
    if (tid == c0 + ZS_FCOLS) s_b[ZS_FCOLS] = ex;
    if (c0 + ZS_FCOLS == NCOL && tid == 0) s_b[ZS_FCOLS] = total;
    __syncthreads();
    const uint32_t Vr = row_rec(counts, cstride, r, ZR_VERTS), vb = zr.vbase[r];
    const uint32_t a = s_b[0], b = s_b[ZS_FCOLS] < Vr ? s_b[ZS_FCOLS] : Vr;
    for (uint32_t i = a + tid; i < b; i += 1024u) {
        uint32_t k = 0;
#pragma unroll
        for (uint32_t j = 1; j < ZS_FCOLS; j++) k += s_b[j] <= i ? 1u : 0u;
        segof[vb + i] = r * NCOL + c0 + k;
        redirect[vb + i] = U32_NONE;
    }
    if (c0 + ZS_FCOLS == NCOL) {
        if (tid == 0) zctl[ZC_RANKBAD + r] = total != Vr ? 1u : 0u;       // the counts do not add up to the rank's vertices
        const uint32_t cap = zr.vbase[r + 1u] - vb;
        for (uint32_t i = Vr + tid; i < cap; i += 1024u) redirect[vb + i] = ZS_GAP;
    }
}

__device__ __forceinline__ bool zs_in_band(const ZsRanks &zr, uint32_t r, float z) {
    return (zr.zhi[r] < zr.ztop && z >= zr.zhi[r] - zr.band) || (zr.zlo[r] > zr.zbot && z <= zr.zlo[r] + zr.band);
}

__global__ __launch_bounds__(256) void k_zs_band_insert(const Vertex *__restrict__ v, const uint32_t *__restrict__ counts, uint32_t cstride,
                                                         const ZsRanks *__restrict__ zrp, const uint32_t *__restrict__ redirect, uint64_t *__restrict__ hkeys,
                                                         uint32_t *__restrict__ hmin, uint32_t hsize, uint32_t *__restrict__ zctl) {
    // (the band is counted per workgroup at the END of its grid-stride loop: one counter add per wave that holds a band vertex --
    // nearly every wave does, every column crosses the slab faces -- would queue tens of thousands of atomics on one word.  The table
    // has room for twice the keys it was planned for; a probe sequence that gets long anyway means the plan was too small.)
    const ZsRanks &zr = *zrp;
    __shared__ uint32_t s_n;
    if (!zs_rows_ok(counts, cstride, zr)) return;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const uint32_t V = zr.vbase[zr.nranks];
    uint32_t mine = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < V; i += gridDim.x * blockDim.x) {
        if (redirect[i] == ZS_GAP) continue;
        const Point p = v[i].position;
        if (!zs_in_band(zr, zs_rank_of(zr, i), p.z)) continue;
        mine++;
        uint64_t key;
        if (!packed_weld_key(p, key)) { atomicOr(&zctl[ZC_BANDFLAGS], 1u); continue; }
        uint32_t s = hash_home(key, hsize);
        bool placed = false;
        for (uint32_t probe = 0; probe < 512u && probe < hsize; probe++) {
            const uint64_t prev = atomicCAS((unsigned long long *) &hkeys[s], (unsigned long long) KEY_EMPTY, (unsigned long long) key);
            if (prev == KEY_EMPTY || prev == key) { atomicMin(&hmin[s], i); placed = true; break; }
            s = hash_next(s, hsize);
        }
        if (!placed) atomicOr(&zctl[ZC_BANDFLAGS], 64u);
    }
    const uint32_t ws = wave_sum(mine);
    if (lane_id() == 0 && ws) atomicAdd(&s_n, ws);
    __syncthreads();
    if (threadIdx.x == 0 && s_n) atomicAdd(&zctl[ZC_NBAND], s_n);
}

__global__ __launch_bounds__(256) void k_zs_band_resolve(const Vertex *__restrict__ v, const uint32_t *__restrict__ counts, uint32_t cstride,
                                                          const ZsRanks *__restrict__ zrp, const uint64_t *__restrict__ hkeys,
                                                          const uint32_t *__restrict__ hmin, uint32_t hsize, const uint32_t *__restrict__ segof,
                                                          uint32_t *__restrict__ redirect, uint32_t *__restrict__ newcnt, uint32_t *__restrict__ blockdrop) {
    const ZsRanks &zr = *zrp;
    if (!zs_rows_ok(counts, cstride, zr)) return;
    const uint32_t V = zr.vbase[zr.nranks];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < V; i += gridDim.x * blockDim.x) {
        if (redirect[i] == ZS_GAP) continue;          // (a slot entry behind the rank's vertices; only `lose` entries are rewritten below, never to ZS_GAP)
        const Point p = v[i].position;
        if (!zs_in_band(zr, zs_rank_of(zr, i), p.z)) continue;
        uint64_t key;
        if (!packed_weld_key(p, key)) continue;
        uint32_t s = hash_home(key, hsize);
        for (uint32_t probe = 0; probe < 512u && probe < hsize; probe++) {
            const uint64_t k = hkeys[s];
            if (k == KEY_EMPTY) break;
            if (k == key) {
                const uint32_t a = hmin[s];
                if (a != i) {                                            // (a < i: the copy of a lower rank -- a rank's own keys are distinct)
                    const uint32_t sa = segof[a], si = segof[i];
                    // single-GPU order of the two first occurrences: by column, then by rank
                    const bool a_first = (sa % NCOL) < (si % NCOL) || ((sa % NCOL) == (si % NCOL) && sa < si);
                    const uint32_t win = a_first ? a : i, lose = a_first ? i : a;
                    redirect[lose] = win;
                    atomicSub(&newcnt[a_first ? si : sa], 1u);
                    atomicAdd(&blockdrop[lose / zr.blk], 1u);        // (k_zs_vertex_out numbers the kept vertices of a segment from these)
                }
                break;
            }
            s = hash_next(s, hsize);
        }
    }
}

// more than one rank: the vertex segments' merged places, from the counts the band pairs have decremented (one workgroup)
__global__ __launch_bounds__(1024) void k_zs_vplan(const uint32_t *__restrict__ counts, uint32_t cstride, const ZsRanks *__restrict__ zrp,
                                                    const uint32_t *__restrict__ newcnt, uint32_t *__restrict__ vplan, uint32_t cap_verts,
                                                    uint32_t *__restrict__ zctl) {
    const ZsRanks &zr = *zrp;
    __shared__ uint32_t lds[17];
    if (!zs_rows_ok(counts, cstride, zr)) return;
    const uint32_t V = zs_plan_body(newcnt, NCOL, zr.nranks, vplan, lds);
    if (threadIdx.x == 0) {
        zctl[ZC_VERTS] = V;
        zctl[ZC_VFLAGS] = V > cap_verts ? 4u : 0u;
    }
}

// kept vertices to their merged places; map[i] = merged number.  Flat over the slot space, 256 IPT entries per workgroup, IPT consecutive ones
// per thread, every load that does not depend on another one issued in front.  A kept vertex's place = its segment's merged start + its rank
// among the segment's kept vertices = (i - first) - (dropped copies in [first, i)).  Only band vertices are ever dropped, and only with
// several ranks: the drops in front of i are counted inside the workgroup's own entries by a block scan; a segment that began in an earlier
// block (only the block's first segment can) adds the drops between its first vertex and the block -- the tail of the first vertex's
// block, read here, and the whole blocks in between from blockdrop[] (k_zs_band_resolve).  No global scan.  The vertices themselves move
// as dwords through registers with their destinations staged in LDS: a 24-byte vertex per lane would touch every line six times.
template <uint32_t IPT>
__global__ __launch_bounds__(256) void k_zs_vertex_out(const Vertex *__restrict__ v, const uint32_t *__restrict__ counts, uint32_t cstride,
                                                        const ZsRanks *__restrict__ zrp, const uint32_t *__restrict__ redirect, const uint32_t *__restrict__ segof,
                                                        const uint32_t *__restrict__ vstart, const uint32_t *__restrict__ vplan, const uint32_t *__restrict__ blockdrop,
                                                        Vertex *__restrict__ out, uint32_t cap, uint32_t *__restrict__ map, uint32_t flags) {
    constexpr uint32_t BLK = 256u * IPT;
    const ZsRanks &zr = *zrp;
    __shared__ uint32_t lds[17];
    __shared__ uint32_t s_e[BLK];
    __shared__ uint32_t s_x;
    const uint32_t tid = threadIdx.x, V = zr.vbase[zr.nranks], b0 = blockIdx.x * BLK;
    if (b0 >= V) return;
    // (speculative loads: entries behind V or in a slot's tail read allocated memory, their values are not used)
    uint32_t red[IPT], seg_i[IPT];
#pragma unroll
    for (uint32_t j = 0; j < IPT; j++) {
        const uint32_t i = b0 + IPT * tid + j;
        red[j] = i < V ? redirect[i] : ZS_GAP;
        seg_i[j] = i < V ? segof[i] : 0u;
    }
    const uint32_t nd = (V - b0 < BLK ? V - b0 : BLK) * 6u;
    const uint32_t *__restrict__ v32 = reinterpret_cast<const uint32_t *>(v) + 6ull * b0;
    uint32_t w[6u * IPT];
#pragma unroll
    for (uint32_t j = 0; j < 6u * IPT; j++) w[j] = tid + 256u * j < nd ? v32[tid + 256u * j] : 0u;
    if (!zs_rows_ok(counts, cstride, zr)) return;
    const bool drops = zr.nranks > 1u;
    uint32_t first[IPT], dstart[IPT];
#pragma unroll
    for (uint32_t j = 0; j < IPT; j++) {
        first[j] = 0; dstart[j] = 0;
        if (red[j] == U32_NONE) { first[j] = zr.vbase[seg_i[j] / NCOL] + vstart[seg_i[j]]; dstart[j] = vplan[2u * seg_i[j] + 1u]; }
    }
    uint32_t ex = 0, xt = 0;
    if (drops) {
        uint32_t d = 0;
#pragma unroll
        for (uint32_t j = 0; j < IPT; j++) d += (red[j] != U32_NONE && red[j] != ZS_GAP) ? 1u : 0u;
        uint32_t total;
        ex = block_excl_scan(d, total, lds);
        uint32_t e = ex;
#pragma unroll
        for (uint32_t j = 0; j < IPT; j++) { s_e[IPT * tid + j] = e; e += (red[j] != U32_NONE && red[j] != ZS_GAP) ? 1u : 0u; }
        // drops of the block's first segment in front of the block (thread 0 holds the block's first entry)
        if (tid == 0) s_x = red[0] != ZS_GAP ? zr.vbase[seg_i[0] / NCOL] + vstart[seg_i[0]] : b0;
        __syncthreads();
        const uint32_t first0 = s_x;
        uint32_t x = 0;
        if (first0 < b0) {
            const uint32_t fb = first0 / BLK, fend = (fb + 1u) * BLK;              // (fend <= b0)
            for (uint32_t j = first0 + tid; j < fend; j += 256u) { const uint32_t q = redirect[j]; x += (q != U32_NONE && q != ZS_GAP) ? 1u : 0u; }
            for (uint32_t bb = fb + 1u + tid; bb < blockIdx.x; bb += 256u) x += blockdrop[bb];
        }
        (void) block_excl_scan(x, xt, lds);                                  // (synchronises: every reader of s_e below is behind it)
    }
    uint32_t g[IPT];
#pragma unroll
    for (uint32_t j = 0; j < IPT; j++) {
        g[j] = U32_NONE;
        if (red[j] == U32_NONE) {
            const uint32_t i = b0 + IPT * tid + j;
            uint32_t before = 0;
            if (drops) before = first[j] >= b0 ? s_e[i - b0] - s_e[first[j] - b0] : s_e[i - b0] + xt;
            g[j] = dstart[j] + (i - first[j]) - before;
            map[i] = g[j];
            if (g[j] >= cap) g[j] = U32_NONE;                              // flagged by the plan
        }
    }
    __syncthreads();                                                         // (s_e: its last readers are behind this barrier)
#pragma unroll
    for (uint32_t j = 0; j < IPT; j++) s_e[IPT * tid + j] = g[j];
    __syncthreads();
    uint32_t *__restrict__ out32 = reinterpret_cast<uint32_t *>(out);
#pragma unroll
    for (uint32_t j = 0; j < 6u * IPT; j++) {
        const uint32_t k = tid + 256u * j;
        if (k < nd) { const uint32_t gg = s_e[k / 6u]; if (gg != U32_NONE) out32[6ull * gg + k % 6u] = w[j]; }
    }
    (void) flags;
}

// a dropped copy takes its partner's number (the partner is kept: its number is final).  Its own pass over the vertices: folded into the
// index rewrite it would cost every one of the 3T index lookups a second gather (measured: +40 us at 2048^3, +0.2 ms at 4096^3)
__global__ __launch_bounds__(256) void k_zs_loser_map(const uint32_t *__restrict__ counts, uint32_t cstride, const ZsRanks *__restrict__ zrp,
                                                       const uint32_t *__restrict__ redirect, uint32_t *__restrict__ map) {
    const ZsRanks &zr = *zrp;
    if (!zs_rows_ok(counts, cstride, zr)) return;
    const uint32_t V = zr.vbase[zr.nranks];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < V; i += gridDim.x * blockDim.x) {
        const uint32_t w = redirect[i];
        if (w != U32_NONE && w != ZS_GAP) map[i] = map[w];
    }
}

__device__ __forceinline__ void zs_publish_body(const uint32_t *__restrict__ counts, uint32_t cstride, uint32_t nranks, const uint32_t *__restrict__ zctl,
                                                uint32_t seq, uint32_t *__restrict__ pub) {
    const uint32_t tid = threadIdx.x;
    if (tid < ZC_WORDS) pub[tid] = tid == ZC_SEQ ? seq : (zctl ? zctl[tid] : 0u);
    if (tid < nranks * 8u) pub[ZC_WORDS + tid] = counts[(size_t) (tid >> 3) * cstride + SHARD_REC + (tid & 7u)];
}

// the index buffers into the merged triangle order with merged vertex numbers.  Flat over the triangle slots: workgroup = 256 IPT slot
// triangles of one rank (blocks per rank from the slot capacities: zr.tblk), IPT triangles per thread; a triangle finds its column segment
// by bisection over the few segment starts the block can fall into (tplan's source offsets between the block's hint and the next one's).
// The indices move as dwords (three coalesced loads per triangle-slot of the thread, their destinations staged in LDS).  Workgroup 0 also
// publishes the step's control block and the ranks' records to pinned host memory: every word of them was final before this kernel
// started (no read-back copy behind the chain).
template <uint32_t IPT>
__global__ __launch_bounds__(256) void k_zs_index_out(const uint32_t *__restrict__ idx_all, const uint32_t *__restrict__ counts, uint32_t cstride,
                                                       const uint32_t *__restrict__ tplan, const uint32_t *__restrict__ thint, const ZsRanks *__restrict__ zrp,
                                                       const uint32_t *__restrict__ map, uint32_t *__restrict__ out, uint32_t cap_tris,
                                                       const uint32_t *__restrict__ zctl, uint32_t seq, uint32_t *__restrict__ pub) {
    constexpr uint32_t BLK = 256u * IPT;
    const ZsRanks &zr = *zrp;
    __shared__ uint32_t s_start[NCOL + 1];
    __shared__ uint32_t s_dst[BLK];
    if (pub && blockIdx.x == 0) zs_publish_body(counts, cstride, zr.nranks, zctl, seq, pub);
    uint32_t r = 0;
    while (r + 1u < zr.nranks && blockIdx.x >= zr.tblk[r + 1u]) r++;
    const uint32_t tid = threadIdx.x, t0 = (blockIdx.x - zr.tblk[r]) * BLK;
    const uint32_t slot = zr.tbase[r + 1u] - zr.tbase[r];
    if (t0 >= slot) return;
    // the block's indices as coalesced dword loads (speculative: in front of the hints and the checks)
    const uint32_t ne_cap = (slot - t0 < BLK ? slot - t0 : BLK) * 3u;
    const uint32_t *__restrict__ s = idx_all + 3ull * (zr.tbase[r] + t0);
    uint32_t e[3u * IPT];
#pragma unroll
    for (uint32_t j = 0; j < 3u * IPT; j++) e[j] = tid + 256u * j < ne_cap ? s[tid + 256u * j] : 0u;
    // the columns this block's triangles can fall into: from its own first triangle's to the next block's (k_zs_front's hints)
    uint32_t c_lo = thint[blockIdx.x], c_hi = blockIdx.x + 1u < zr.tblk[r + 1u] ? thint[blockIdx.x + 1u] : NCOL - 1u;
    if (!zs_rows_ok(counts, cstride, zr)) return;
    if (c_lo >= NCOL || c_hi >= NCOL || c_hi < c_lo) { c_lo = 0; c_hi = NCOL - 1u; }          // (never of a checked row)
    const uint32_t w = c_hi - c_lo + 1u;
    for (uint32_t k = tid; k < w; k += 256u) s_start[k] = tplan[2u * (r * NCOL + c_lo + k)];
    const uint32_t Tr = row_rec(counts, cstride, r, ZR_TRIS), Vr = row_rec(counts, cstride, r, ZR_VERTS), vb = zr.vbase[r];
    if (t0 >= Tr) return;
    const uint32_t ne = (Tr - t0 < BLK ? Tr - t0 : BLK) * 3u;
    uint32_t m[3u * IPT];
#pragma unroll
    for (uint32_t j = 0; j < 3u * IPT; j++) m[j] = (tid + 256u * j < ne && e[j] < Vr) ? map[vb + e[j]] : 0u;
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < IPT; j++) {
        const uint32_t t = t0 + tid + 256u * j;
        uint32_t dst = U32_NONE;
        if (t < Tr) {
            uint32_t lo = 0, hi = w;                                    // the largest k with s_start[k] <= t
            while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (s_start[mid] <= t) lo = mid; else hi = mid; }
            dst = tplan[2u * (r * NCOL + c_lo + lo) + 1u] + (t - s_start[lo]);
            if (dst >= cap_tris) dst = U32_NONE;                        // flagged by k_zs_front
        }
        s_dst[tid + 256u * j] = dst;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < 3u * IPT; j++) {
        const uint32_t k = tid + 256u * j;
        if (k < ne) { const uint32_t dd = s_dst[k / 3u]; if (dd != U32_NONE) out[3ull * dd + k % 3u] = m[j]; }
    }
}

__global__ __launch_bounds__(256) void k_zs_publish(const uint32_t *__restrict__ counts, uint32_t cstride, uint32_t nranks, const uint32_t *__restrict__ zctl,
                                                     uint32_t seq, uint32_t *__restrict__ pub) {
    zs_publish_body(counts, cstride, nranks, zctl, seq, pub);
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
hipError_t launch_zs_local(hipStream_t st, uint32_t *row, uint32_t *next_row, const uint32_t *ctr_tris, const uint32_t *ctr_verts, const uint32_t *status,
                           uint32_t host_status, uint32_t cap_v, uint32_t cap_t, uint32_t inject, uint32_t sig) {
    hipLaunchKernelGGL(k_zs_rowfinish, dim3(1), dim3(1024), 0, st, row, next_row, ctr_tris, ctr_verts, status, host_status, cap_v, cap_t, inject, sig);
    return hipGetLastError();
}

hipError_t launch_zs_merge(hipStream_t st, int grid, const Vertex *v_all, const uint32_t *idx_all, const uint32_t *counts, uint32_t cstride,
                           const ZsRanks &zr, const ZsRanks *d_zr, uint32_t *tplan, uint32_t *vplan, uint32_t *vstart, uint32_t *newcnt,
                           uint32_t *segof, uint32_t *redirect, uint32_t *map, uint32_t *blockdrop, uint32_t *thint, uint64_t *hkeys, uint32_t hsize,
                           Vertex *v_out, uint32_t cap_verts, uint32_t *i_out, uint32_t cap_tris, uint32_t *zctl, uint32_t seq, uint32_t *pub) {
    const uint32_t R = zr.nranks;
    const bool band = R > 1;
    hipLaunchKernelGGL(k_zs_front, dim3(R * (NCOL / ZS_FCOLS) + 2u), dim3(1024), 0, st, counts, cstride, d_zr, tplan, vplan, vstart, newcnt, segof, redirect,
                       band ? (uint32_t *) hkeys : nullptr, band ? 3u * hsize : 0u, band ? blockdrop : nullptr, thint, cap_verts, cap_tris, zctl);
    if (band) {
        uint32_t *hmin = (uint32_t *) (hkeys + hsize);
        hipLaunchKernelGGL(k_zs_band_insert, dim3(grid), dim3(256), 0, st, v_all, counts, cstride, d_zr, (const uint32_t *) redirect, hkeys, hmin, hsize, zctl);
        hipLaunchKernelGGL(k_zs_band_resolve, dim3(grid), dim3(256), 0, st, v_all, counts, cstride, d_zr, (const uint64_t *) hkeys, (const uint32_t *) hmin,
                           hsize, (const uint32_t *) segof, redirect, newcnt, blockdrop);
        hipLaunchKernelGGL(k_zs_vplan, dim3(1), dim3(1024), 0, st, counts, cstride, d_zr, (const uint32_t *) newcnt, vplan, cap_verts, zctl);
    }
    const bool big = zr.blk == 1024u;
    const dim3 vgrid((zr.vbase[R] + zr.blk - 1u) / zr.blk + 1u);
    if (big) hipLaunchKernelGGL(k_zs_vertex_out<4>, vgrid, dim3(256), 0, st, v_all, counts, cstride, d_zr, (const uint32_t *) redirect, (const uint32_t *) segof,
                                (const uint32_t *) vstart, (const uint32_t *) vplan, (const uint32_t *) blockdrop, v_out, cap_verts, map, 0u);
    else hipLaunchKernelGGL(k_zs_vertex_out<1>, vgrid, dim3(256), 0, st, v_all, counts, cstride, d_zr, (const uint32_t *) redirect, (const uint32_t *) segof,
                            (const uint32_t *) vstart, (const uint32_t *) vplan, (const uint32_t *) blockdrop, v_out, cap_verts, map, 0u);
    if (band) hipLaunchKernelGGL(k_zs_loser_map, dim3(grid), dim3(256), 0, st, counts, cstride, d_zr, (const uint32_t *) redirect, map);
    const dim3 tgrid(std::max<uint32_t>(1u, zr.tblk[R]));
    if (big) hipLaunchKernelGGL(k_zs_index_out<4>, tgrid, dim3(256), 0, st, idx_all, counts, cstride, (const uint32_t *) tplan, (const uint32_t *) thint, d_zr,
                                (const uint32_t *) map, i_out, cap_tris, (const uint32_t *) zctl, seq, pub);
    else hipLaunchKernelGGL(k_zs_index_out<1>, tgrid, dim3(256), 0, st, idx_all, counts, cstride, (const uint32_t *) tplan, (const uint32_t *) thint, d_zr,
                            (const uint32_t *) map, i_out, cap_tris, (const uint32_t *) zctl, seq, pub);
    return hipGetLastError();
}

hipError_t launch_zs_publish(hipStream_t st, const uint32_t *counts, uint32_t cstride, uint32_t nranks, const uint32_t *zctl, uint32_t seq, uint32_t *pub) {
    hipLaunchKernelGGL(k_zs_publish, dim3(1), dim3(256), 0, st, counts, cstride, nranks, zctl, seq, pub);
    return hipGetLastError();
}

}  // namespace sdm
