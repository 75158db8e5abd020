// kernels.hip -- gfx950 kernels of the SDF mesh-generation path.
//
// Reference behaviour being reproduced (results, not structure):
//   refine      cuda/modules/compute_mesh_generation.cu:12-62  + host retain src/cuda/mod.rs:192-193
//   triangles   cuda/modules/compute_mesh_generation.cu:64-120 + cuda/includes/marching_cubes.cu:13-43
//   weld        src/cuda/mod.rs:263-296
//
// MI355X design:
//   * every kernel is a fixed-size grid of 256-thread workgroups (4 wave64) that grid-strides
//     over 256-item tiles and reads its item count from device memory, so a whole regen is a
//     fixed launch sequence (hipGraph-capturable, no host round trip per level);
//   * order is semantics: compaction is count -> scan -> emit with wave64 shuffles/ballots, never
//     an atomic append, so voxel and triangle order equal the reference's key order;
//   * the expensive part (Newton projection + normals) runs ONCE per unique lattice edge instead
//     of once per triangle corner (~6x fewer evaluations, bit-identical because the projection
//     is a pure function of the midpoint), with in-wave lane refill to hide the divergent
//     iteration counts;
//   * atomics are used only where the result is order independent (hash insert, atomicMin of
//     first-seen ordinals, bitmap OR).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sdmesh/bindings.h"
#include "kernels.hpp"
#include "prims.hpp"
#include "mc_tables.hpp"
#include "mesh_device.hpp"
#include "sdf.hpp"
#include "render_device.hpp"

#if !defined(SDM_FAST_MATH)
#pragma clang fp contract(off)       // strict build: no FMA contraction (DESIGN.md "Numerics")
#endif

namespace sdm {

// ------------------------------------------------------------------------------------------
// lattice addressing
// A voxel field is a *lattice field* when every base coordinate equals L(m) = (float)m * h - 2.5f
// for an integer m in [0, 2^20) AND base + h == L(m + 1), bit for bit.  Then a lattice corner has
// the same float position whichever voxel computes it, an edge midpoint is identified by the
// integer sum of its two corner coordinates, and (same key) => (same midpoint float) => (same
// projection).  Fields made by create_voxel_field + refine are lattice fields by construction
// (all values are exact dyadic multiples); anything else takes the direct path.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float lat(int m, float h) { return (float) m * h - 2.5f; }

__device__ __forceinline__ bool lattice_coord(float b, float h, int &m) {
    float mf = rintf((b + 2.5f) / h);
    bool ok = (mf >= 0.0f) && (mf < 1048575.0f);
    m = ok ? (int) mf : 0;
    return ok && (lat(m, h) == b) && (lat(m + 1, h) == b + h);
}

// key of cube edge e of the voxel at lattice (mx, my, mz): the doubled-lattice coordinate of the edge midpoint, i.e.
// 2 m + (corner offsets of the edge's two ends).  The offset sums (0, 1 or 2 per axis) of the 12 edges are packed two
// bits each into immediates (derived from MC_EDGE_CORNERS: cox/coy/coz of both ends) -- no table load per corner.
__device__ __forceinline__ uint64_t edge_key(int mx, int my, int mz, int e) {
    const uint32_t ox = (0x281919u >> (2 * e)) & 3u, oy = (0xa06464u >> (2 * e)) & 3u, oz = (0x55aa00u >> (2 * e)) & 3u;
    const uint64_t kx = (uint64_t) (2 * mx) + ox, ky = (uint64_t) (2 * my) + oy, kz = (uint64_t) (2 * mz) + oz;
    return kx | (ky << 21) | (kz << 42);
}

// ------------------------------------------------------------------------------------------
// A2 refine, pass 1: child masks.  One lane per (parent, lattice sample): 32 lanes per parent
// (27 active), sign bits gathered with one wave64 ballot (2 parents per wave).  A workgroup owns
// chunks of 32 consecutive parents (all inside one 256-parent tile) and adds the chunk's kept-child
// count to tilecnt[tile] with a single global atomic (order independent; tilecnt is pre-zeroed).
// ------------------------------------------------------------------------------------------
static constexpr uint32_t MASK_CHUNK = 32;

// child (i,j,k) of a parent is a border voxel iff its 8 corner signs (bits of the 27-sample field) are not all equal
__device__ __forceinline__ uint32_t child_mask_from_bits(uint32_t bits) {
    uint32_t m = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const int i = c >> 2, j = (c >> 1) & 1, k = c & 1;
        const uint32_t cnt = __popc(bits & (0x361Bu << (i * 9 + j * 3 + k)));
        if (cnt != 0 && cnt != 8) m |= 1u << c;
    }
    return m;
}

// One lane per parent: the 27 lattice samples of a parent in one lane, z innermost, so the (x, y) terms of a column
// are computed once for its three samples and the x terms once per plane (sd_obj: ~65 instead of 93 instructions per
// sample); no ballot, no idle lanes (the 32-lane layout leaves 5 of 32 idle), and a workgroup is exactly one 256-parent
// tile, so the tile count is a plain store.  Same sample positions, same `sd <= 0` test, same outputs as
// k_refine_mask.  Used for the compiled scenes; the program interpreter keeps the 32-lane kernel.
template <int SCENE>
__global__ __launch_bounds__(256) void k_refine_mask1(const Point *__restrict__ vox, const uint32_t *__restrict__ n_ptr, Point cs,
                                                       const SceneParams *__restrict__ spp, uint8_t *__restrict__ mask,
                                                       uint32_t *__restrict__ tilecnt, uint32_t *__restrict__ signs) {
    const SceneParams &sp = *spp;
    __shared__ uint32_t s_cnt;
    const uint32_t n = *n_ptr;
    const uint32_t ntiles = (n + TILE - 1) / TILE;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        const uint32_t p = tile * TILE + threadIdx.x;
        uint32_t cnt = 0;
        if (p < n) {
            const Point b = vox[p];
            uint32_t bits = 0;
#pragma unroll 1
            for (int ix = 0; ix < 3; ix++) {
                const float x = b.x + (float) ix * cs.x;
#pragma unroll 1
                for (int iy = 0; iy < 3; iy++) {
                    const float y = b.y + (float) iy * cs.y;
#pragma unroll
                    for (int iz = 0; iz < 3; iz++) {
                        const float z = b.z + (float) iz * cs.z;
                        if (sdf<SCENE>(x, y, z, sp) <= 0.0f) bits |= 1u << (ix * 9 + iy * 3 + iz);
                    }
                }
            }
            uint32_t m = 0;
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const int i = c >> 2, j = (c >> 1) & 1, k = c & 1;
                const uint32_t pc = __popc(bits & (0x361Bu << (i * 9 + j * 3 + k)));
                if (pc != 0 && pc != 8) m |= 1u << c;
            }
            mask[p] = (uint8_t) m;
            if (signs) signs[p] = bits;
            cnt = __popc(m);
        }
        const uint32_t ws = wave_sum(cnt);
        if (lane_id() == 0 && ws) atomicAdd(&s_cnt, ws);
        __syncthreads();
        if (threadIdx.x == 0) tilecnt[tile] = s_cnt;
        __syncthreads();
    }
}

template <int SCENE, uint32_t CH>
__global__ __launch_bounds__(256) void k_refine_mask(const Point *__restrict__ vox, const uint32_t *__restrict__ n_ptr,
                                                      Point cs /* child size = parent size / 2 */, const SceneParams *__restrict__ spp,
                                                      uint8_t *__restrict__ mask, uint32_t *__restrict__ tilecnt,
                                                      uint32_t *__restrict__ signs /* optional: the 27 sign bits per parent */) {
    const SceneParams &sp = *spp;  // scene uniform in device memory (a captured graph replays with new values)
    __shared__ uint32_t s_cnt;
    const uint32_t n = *n_ptr;
    const uint32_t nchunks = (n + CH - 1) / CH;
    const uint32_t grp = threadIdx.x >> 5, s = threadIdx.x & 31u;     // 8 parents per pass, sample id
    const int ix = (int) (s / 9u), iy = (int) ((s / 3u) % 3u), iz = (int) (s % 3u);
    for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        uint32_t local = 0;
        Point pb[CH / 8];
#pragma unroll
        for (uint32_t it = 0; it < CH / 8; it++) {        // all loads in flight before the first eval
            const uint32_t p = chunk * CH + it * 8 + grp;
            pb[it] = vox[p < n ? p : (n - 1)];
        }
#pragma unroll
        for (uint32_t it = 0; it < CH / 8; it++) {
            const uint32_t p = chunk * CH + it * 8 + grp;
            bool in = false;
            if (p < n && s < 27) {
                const Point b = pb[it];
                in = sdf<SCENE>(b.x + (float) ix * cs.x, b.y + (float) iy * cs.y, b.z + (float) iz * cs.z, sp) <= 0.0f;
            }
            const uint64_t bal = __ballot(in);
            if (s == 0 && p < n) {
                const uint32_t bits = (uint32_t) (bal >> (threadIdx.x & 32u)) & 0x7FFFFFFu;   // bit ix*9 + iy*3 + iz
                uint32_t m = 0;
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    const int i = c >> 2, j = (c >> 1) & 1, k = c & 1;
                    const uint32_t cnt = __popc(bits & (0x361Bu << (i * 9 + j * 3 + k)));
                    if (cnt != 0 && cnt != 8) m |= 1u << c;
                }
                mask[p] = (uint8_t) m;
                if (signs) signs[p] = bits;
                local += __popc(m);
            }
        }
        if (local) atomicAdd(&s_cnt, local);
        __syncthreads();
        if (threadIdx.x == 0 && s_cnt) atomicAdd(&tilecnt[(chunk * CH) / TILE], s_cnt);
        __syncthreads();
    }
}

// Exclusive scan of per-tile counts (single workgroup), total -> *total_out (clamped to cap).
__global__ __launch_bounds__(1024) void k_scan_tiles(uint32_t *__restrict__ cnt, const uint32_t *__restrict__ n_ptr,
                                                      uint32_t items_per_tile, uint32_t *__restrict__ total_out,
                                                      uint32_t cap, uint32_t *__restrict__ status) {
    __shared__ uint32_t lds[17];
    const uint32_t n = *n_ptr;
    const uint32_t ntiles = (n + items_per_tile - 1) / items_per_tile;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < ntiles; base += blockDim.x) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < ntiles ? cnt[i] : 0;
        uint32_t total;
        const uint32_t ex = block_excl_scan(v, total, lds);
        if (i < ntiles) cnt[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) {
        if (carry > cap) { atomicOr(&status[ST_OVERFLOW], 1u); carry = cap; }
        *total_out = carry;
    }
}

// A2 refine, pass 2: ordered emit of the border children (slot order id*8 + i*4 + j*2 + k).
// Each workgroup derives its tile's output offset from the per-tile counts itself (no scan kernel);
// the last tile publishes the total.
__global__ __launch_bounds__(256) void k_refine_emit(const Point *__restrict__ vox, const uint32_t *__restrict__ n_ptr,
                                                      Point cs, const uint8_t *__restrict__ mask,
                                                      const uint32_t *__restrict__ tilecnt, Point *__restrict__ out,
                                                      uint32_t cap, uint32_t *__restrict__ n_out, uint32_t *__restrict__ status,
                                                      const uint32_t *__restrict__ signs, uint8_t *__restrict__ cubeidx,
                                                      uint32_t *__restrict__ mtilecnt, uint32_t prescanned) {
    // prescanned: tilecnt already holds the EXCLUSIVE prefix of the tile counts and *n_out the total (k_scan_tiles ran in front:
    // above a few thousand tiles the per-workgroup re-summing below is quadratic in the tile count -- 2 G loads at 4096^3)
    // signs != nullptr (last level of a device-resident regen): the parent's 27 lattice signs already hold the 8 corner
    // signs of every child (same sample positions on an exact lattice, same `sd <= 0` test as k_classify), so the
    // marching-cubes index of each child and the triangle count of each 256-voxel MESH tile (pre-zeroed mtilecnt) are
    // written here and k_classify is not launched.
    __shared__ uint32_t lds[17];
    __shared__ uint32_t s_mt[12];       // <= 2048 children per pass -> at most 9 mesh tiles touched
    const uint32_t n = *n_ptr;
    const uint32_t ntiles = (n + TILE - 1) / TILE;
    if (n == 0 && blockIdx.x == 0 && threadIdx.x == 0 && !prescanned) *n_out = 0;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint32_t tbase;
        if (prescanned) {
            tbase = tilecnt[tile];
        } else {
            uint32_t part = 0;
            for (uint32_t i = threadIdx.x; i < tile; i += 256) part += tilecnt[i];
            (void) block_excl_scan(part, tbase, lds);
        }
        if (signs && threadIdx.x < 12) s_mt[threadIdx.x] = 0;      // (block_excl_scan below synchronises)
        const uint32_t p = tile * TILE + threadIdx.x;
        const uint32_t m = p < n ? mask[p] : 0u;
        uint32_t total;
        uint32_t o = tbase + block_excl_scan(__popc(m), total, lds);
        if (tile == ntiles - 1 && threadIdx.x == 0 && !prescanned) {
            uint32_t tot = tbase + total;
            if (tot > cap) { atomicOr(&status[ST_OVERFLOW], 1u); tot = cap; }
            *n_out = tot;
        }
        if (m) {
            const Point b = vox[p];
            const uint32_t sg = signs ? signs[p] : 0u;
            uint32_t tri_lo = 0, tri_hi = 0;                        // a parent's <= 8 children span at most 2 mesh tiles
            const uint32_t t_lo = o / TILE;
#pragma unroll
            for (int c = 0; c < 8; c++) {
                if (m & (1u << c)) {
                    if (o < cap) {
                        Point q;
                        q.x = b.x + (float) (c >> 2) * cs.x;
                        q.y = b.y + (float) ((c >> 1) & 1) * cs.y;
                        q.z = b.z + (float) (c & 1) * cs.z;
                        out[o] = q;
                        if (signs) {
                            const uint32_t w = sg >> ((c >> 2) * 9 + ((c >> 1) & 1) * 3 + (c & 1));   // child's 2x2x2 corner of the 3^3
                            uint32_t ci = 0;
#pragma unroll
                            for (int cz = 0; cz < 2; cz++)
#pragma unroll
                                for (int cq = 0; cq < 4; cq++)
                                    ci |= ((w >> (cox(cq) * 9 + coy(cq) * 3 + cz)) & 1u) << (cz * 4 + cq);
                            cubeidx[o] = (uint8_t) ci;
                            const uint32_t nt = MC_TRI_COUNT[ci];
                            if (o / TILE == t_lo) tri_lo += nt; else tri_hi += nt;
                        }
                    }
                    o++;
                }
            }
            if (signs) {
                const uint32_t t0 = tbase / TILE;
                if (tri_lo) atomicAdd(&s_mt[t_lo - t0], tri_lo);
                if (tri_hi) atomicAdd(&s_mt[t_lo + 1 - t0], tri_hi);
            }
        }
        if (signs) {
            __syncthreads();
            if (threadIdx.x < 12 && s_mt[threadIdx.x]) atomicAdd(&mtilecnt[tbase / TILE + threadIdx.x], s_mt[threadIdx.x]);
            __syncthreads();
        }
    }
}

// total kept children = sum of the per-tile counts (single workgroup; host-API sizing pass)
__global__ __launch_bounds__(1024) void k_sum_tiles(const uint32_t *__restrict__ tilecnt, const uint32_t *__restrict__ n_ptr,
                                                     uint32_t *__restrict__ total_out) {
    __shared__ uint32_t lds[17];
    const uint32_t ntiles = (*n_ptr + TILE - 1) / TILE;
    uint32_t part = 0;
    for (uint32_t i = threadIdx.x; i < ntiles; i += blockDim.x) part += tilecnt[i];
    uint32_t total;
    (void) block_excl_scan(part, total, lds);
    if (threadIdx.x == 0) *total_out = total;
}

// ------------------------------------------------------------------------------------------
// A2 refine, two levels per step (planned path).  The refine chain is latency-bound (every tiny kernel sits on a
// ~5 us floor), so the level L+1 list is never materialised: one mask kernel evaluates, per level-L parent, its 8
// children (27 samples) and then the 8 children of every KEPT child (27 samples each, at the kept child's own float
// position, exactly as a second refine launch would), one emit kernel writes the level-(L+2) voxels in the
// reference's order (parent-major, child slot, grandchild slot).  mask2[p] packs the 8 grandchild masks of parent p.
// ------------------------------------------------------------------------------------------
template <int SCENE>
__global__ __launch_bounds__(256) void k_refine2_mask(const Point *__restrict__ vox, const uint32_t *__restrict__ n_ptr, Point cs1,
                                                       const SceneParams *__restrict__ spp, uint8_t *__restrict__ mask1,
                                                       uint64_t *__restrict__ mask2, uint32_t *__restrict__ tilecnt,
                                                       uint32_t *__restrict__ n_mid) {
    const SceneParams &sp = *spp;
    __shared__ uint32_t s_cnt1, s_cnt2;
    const uint32_t n = *n_ptr;
    const uint32_t nchunks = (n + MASK_CHUNK - 1) / MASK_CHUNK;
    const uint32_t grp = threadIdx.x >> 5, s = threadIdx.x & 31u;
    const int ix = (int) (s / 9u), iy = (int) ((s / 3u) % 3u), iz = (int) (s % 3u);
    const Point cs2 = Point{cs1.x / 2.0f, cs1.y / 2.0f, cs1.z / 2.0f};      // src/cuda/mod.rs:126-130
    for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        if (threadIdx.x == 0) { s_cnt1 = 0; s_cnt2 = 0; }
        __syncthreads();
        uint32_t local1 = 0, local2 = 0;
#pragma unroll 1
        for (uint32_t it = 0; it < MASK_CHUNK / 8; it++) {
            const uint32_t p = chunk * MASK_CHUNK + it * 8 + grp;
            const Point b = vox[p < n ? p : (n - 1)];
            bool in = false;
            if (p < n && s < 27)
                in = sdf<SCENE>(b.x + (float) ix * cs1.x, b.y + (float) iy * cs1.y, b.z + (float) iz * cs1.z, sp) <= 0.0f;
            const uint64_t bal = __ballot(in);
            const uint32_t m1 = p < n ? child_mask_from_bits((uint32_t) (bal >> (threadIdx.x & 32u)) & 0x7FFFFFFu) : 0u;
            const uint32_t mu = m1 | (uint32_t) __shfl_xor((int) m1, 32, 64);          // children either half-wave needs
            uint64_t packed = 0;
#pragma unroll 1
            for (uint32_t c = 0; c < 8; c++) {
                if (!((mu >> c) & 1u)) continue;                                        // wave-uniform
                const bool kept = (m1 >> c) & 1u;
                const float cx = b.x + (float) (c >> 2) * cs1.x, cy = b.y + (float) ((c >> 1) & 1u) * cs1.y, cz = b.z + (float) (c & 1u) * cs1.z;
                bool in2 = false;
                if (kept && s < 27)
                    in2 = sdf<SCENE>(cx + (float) ix * cs2.x, cy + (float) iy * cs2.y, cz + (float) iz * cs2.z, sp) <= 0.0f;
                const uint64_t bal2 = __ballot(in2);
                if (kept) {
                    const uint32_t m2 = child_mask_from_bits((uint32_t) (bal2 >> (threadIdx.x & 32u)) & 0x7FFFFFFu);
                    packed |= (uint64_t) m2 << (8u * c);
                }
            }
            if (s == 0 && p < n) {
                mask1[p] = (uint8_t) m1;
                mask2[p] = packed;
                local1 += __popc(m1);
                local2 += __popcll(packed);
            }
        }
        if (local1) atomicAdd(&s_cnt1, local1);
        if (local2) atomicAdd(&s_cnt2, local2);
        __syncthreads();
        if (threadIdx.x == 0) {
            if (s_cnt2) atomicAdd(&tilecnt[(chunk * MASK_CHUNK) / TILE], s_cnt2);
            if (s_cnt1) atomicAdd(n_mid, s_cnt1);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_refine2_emit(const Point *__restrict__ vox, const uint32_t *__restrict__ n_ptr, Point cs1,
                                                       const uint8_t *__restrict__ mask1, const uint64_t *__restrict__ mask2,
                                                       const uint32_t *__restrict__ tilecnt, Point *__restrict__ out, uint32_t cap,
                                                       uint32_t *__restrict__ n_out, uint32_t *__restrict__ status) {
    __shared__ uint32_t lds[17];
    const uint32_t n = *n_ptr;
    const uint32_t ntiles = (n + TILE - 1) / TILE;
    const Point cs2 = Point{cs1.x / 2.0f, cs1.y / 2.0f, cs1.z / 2.0f};
    if (n == 0 && blockIdx.x == 0 && threadIdx.x == 0) *n_out = 0;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t p = tile * TILE + threadIdx.x;
        const uint32_t m1 = p < n ? mask1[p] : 0u;
        const uint64_t m2 = p < n ? mask2[p] : 0ull;
        const Point b = vox[p < n ? p : (n - 1)];
        uint32_t part = 0;
        for (uint32_t i = threadIdx.x; i < tile; i += 256) part += tilecnt[i];
        uint32_t tbase;
        (void) block_excl_scan(part, tbase, lds);
        uint32_t total;
        uint32_t o = tbase + block_excl_scan((uint32_t) __popcll(m2), total, lds);
        if (tile == ntiles - 1 && threadIdx.x == 0) {
            uint32_t tot = tbase + total;
            if (tot > cap) { atomicOr(&status[ST_OVERFLOW], 1u); tot = cap; }
            *n_out = tot;
        }
        if (m2) {
#pragma unroll 1
            for (uint32_t c = 0; c < 8; c++) {
                const uint32_t mc = (uint32_t) (m2 >> (8u * c)) & 0xFFu;
                if (!mc) continue;
                const float cx = b.x + (float) (c >> 2) * cs1.x, cy = b.y + (float) ((c >> 1) & 1u) * cs1.y, cz = b.z + (float) (c & 1u) * cs1.z;
#pragma unroll
                for (uint32_t g = 0; g < 8; g++) {
                    if (mc & (1u << g)) {
                        if (o < cap) {
                            Point q;
                            q.x = cx + (float) (g >> 2) * cs2.x;
                            q.y = cy + (float) ((g >> 1) & 1u) * cs2.y;
                            q.z = cz + (float) (g & 1u) * cs2.z;
                            out[o] = q;
                        }
                        o++;
                    }
                }
            }
        }
        (void) m1;
    }
}


// ------------------------------------------------------------------------------------------
// A1 + A2, levels 0 .. J in ONE workgroup per level-0 (x0, y0) column ("column refine", J = 1..3).
//
// The refine chain is latency-bound: at 512^3 the first three levels hold 32 768 / 4 136 / 16 532 parents -- 17 to 128
// workgroups on 256 CUs, two dependent launches per level.  A level-0 column (32 cells, z innermost) is a CONTIGUOUS
// range of the reference's order at every level (src/cuda/mod.rs:105-122: x outer, y, z inner; children in slot order
// compute_mesh_generation.cu:51), and all its descendants down to level 3 fit LDS, so one workgroup walks the levels
// itself: level-1 lattice signs of the column -> kept level-1 children (ordered) -> their 27 level-2 samples each ->
// kept level-2 cells (ordered) -> their 27 level-3 samples each -> per-cell child masks.  Same sample positions
// (lattice point m of level l is (float) m * h_l - 2.5f, the value b + i * cs of the reference's float adds: all
// dyadic, exact), same `sd <= 0` test, same "not all 8 corner signs equal" criterion as k_refine_mask.
// k_col_refine writes the ordered lists / masks per column, k_col_emit (next launch: it needs every column's count)
// writes the level-J voxel list in the reference's order.  The workgroups of k_col_refine also clear the regen's
// control block and the weld table (what k_create_field does in the level-by-level chain).
// ------------------------------------------------------------------------------------------
static constexpr uint32_t COL_L1 = 256;      // level-1 children of a column (32 cells x 8)
static constexpr uint32_t COL_L2 = 2048;     // level-2 cells of a column
static constexpr uint32_t COL_S1 = 9 * 65;   // level-1 lattice samples of a column: 3 x 3 x (2 * 32 + 1)

// lattice index (at the level of the entry) of list1 entry t = cell * 8 + slot of column (X0, Y0), slab start Z0
__device__ __forceinline__ void col_l1_index(uint32_t t, uint32_t X0, uint32_t Y0, uint32_t Z0, uint32_t &lx, uint32_t &ly, uint32_t &lz) {
    const uint32_t zc = t >> 3, c = t & 7u;
    lx = 2u * X0 + (c >> 2); ly = 2u * Y0 + ((c >> 1) & 1u); lz = 2u * (Z0 + zc) + (c & 1u);
}

// 27 samples of the cell at lattice (lx, ly, lz) of spacing 2 * hc (children spacing hc): 32 lanes per cell, lane s < 27
// evaluates sample (s / 9, (s / 3) % 3, s % 3); returns the 8-bit kept-children mask in the lane with s == 0
template <int SCENE>
__device__ __forceinline__ uint32_t col_cell_mask(bool valid, uint32_t lx, uint32_t ly, uint32_t lz, float hc, uint32_t s, const SceneParams &sp) {
    bool in = false;
    if (valid && s < 27u) {
        const uint32_t ix = s / 9u, iy = (s / 3u) % 3u, iz = s % 3u;
        in = sdf<SCENE>(lat((int) (2u * lx + ix), hc), lat((int) (2u * ly + iy), hc), lat((int) (2u * lz + iz), hc), sp) <= 0.0f;
    }
    const uint64_t bal = __ballot(in);
    return child_mask_from_bits((uint32_t) (bal >> (threadIdx.x & 32u)) & 0x7FFFFFFu);
}

template <int SCENE>
__global__ __launch_bounds__(256) void k_col_refine(uint32_t x0, uint32_t z0, uint32_t zs, uint32_t J, const SceneParams *__restrict__ spp,
                                                     uint8_t *__restrict__ g_list1, uint8_t *__restrict__ g_mask2, uint8_t *__restrict__ g_mask3,
                                                     uint4 *__restrict__ colinfo, uint32_t *__restrict__ zero, uint32_t nzero,
                                                     uint4 *__restrict__ ff, uint32_t nff16, uint32_t segs) {
    // segs (option "col_segs", default 1): a column cut into `segs` equal z segments, one workgroup ("unit") each -- a segment is a
    // contiguous range of the order too (z is the innermost digit).  Measured slower (2 / 4 / 8 segments: col_refine 20.5 -> 22.8 /
    // 26.6 / 26.9 us, col_emit 8.5 -> 10 / 16 / 37 us): the kernels are bound by the fixed phase chain of a workgroup, not by
    // the heaviest column.
    const SceneParams &sp = *spp;
    __shared__ uint8_t s_sign[COL_S1];
    __shared__ uint8_t s_list1[COL_L1];
    __shared__ uint8_t s_mask2[COL_L1];
    __shared__ uint16_t s_list2[COL_L2];
    __shared__ uint32_t s_scan[17];
    __shared__ uint32_t s_n3;
    const uint32_t tid = threadIdx.x;
    const uint32_t F = MESH_GENERATION_INIT_FACTOR;
    const float h0 = MESH_GENERATION_BB_SIZE / (float) MESH_GENERATION_INIT_FACTOR;
    const float h1 = h0 / 2.0f, h2 = h1 / 2.0f, h3 = h2 / 2.0f;                 // src/cuda/mod.rs:126-130, exact halvings
    const uint32_t unit = blockIdx.x, col = unit / segs;
    zs /= segs;                                                                 // (from here on: the unit's own z extent)
    const uint32_t X0 = x0 + col / F, Y0 = col % F, Z0 = z0 + (unit % segs) * zs;
    const uint32_t nz1 = 2u * zs + 1u;                                          // level-1 lattice points along z
    if (tid == 0) s_n3 = 0;
    // A. signs of the column's level-1 lattice: (ix, iy) in 0..2, iz in 0..2 zs
    for (uint32_t i = tid; i < 9u * nz1; i += 256u) {
        const uint32_t ix = i / (3u * nz1), iy = (i / nz1) % 3u, iz = i % nz1;
        s_sign[i] = sdf<SCENE>(lat((int) (2u * X0 + ix), h1), lat((int) (2u * Y0 + iy), h1), lat((int) (2u * Z0 + iz), h1), sp) <= 0.0f ? 1 : 0;
    }
    __syncthreads();
    // B. level-1 children in order: thread t = cell * 8 + slot; kept iff its 8 corner signs differ
    bool act1 = false;
    if (tid < 8u * zs) {
        const uint32_t zc = tid >> 3, c = tid & 7u, i = c >> 2, j = (c >> 1) & 1u, k = c & 1u;
        uint32_t cnt = 0;
#pragma unroll
        for (uint32_t q = 0; q < 8; q++)
            cnt += s_sign[((i + (q >> 2)) * 3u + (j + ((q >> 1) & 1u))) * nz1 + 2u * zc + k + (q & 1u)];
        act1 = cnt != 0 && cnt != 8;
    }
    uint32_t n1;
    const uint32_t p1 = block_excl_scan(act1 ? 1u : 0u, n1, s_scan);
    if (act1) s_list1[p1] = (uint8_t) tid;
    __syncthreads();
    uint32_t n2 = 0, n3 = 0;
    const uint32_t grp = tid >> 5, s = tid & 31u;
    if (J >= 2 && n1) {
        // C. level-2: 27 samples per kept level-1 child, 8 children per pass
        for (uint32_t a0 = 0; a0 < n1; a0 += 8u) {
            const uint32_t a = a0 + grp;
            uint32_t lx = 0, ly = 0, lz = 0;
            if (a < n1) col_l1_index(s_list1[a], X0, Y0, Z0, lx, ly, lz);
            const uint32_t m = col_cell_mask<SCENE>(a < n1, lx, ly, lz, h2, s, sp);
            if (s == 0 && a < n1) s_mask2[a] = (uint8_t) m;
        }
        __syncthreads();
        const uint32_t m2 = tid < n1 ? s_mask2[tid] : 0u;
        uint32_t o2 = block_excl_scan(__popc(m2), n2, s_scan);
        if (J >= 3) {
            for (uint32_t c = 0; c < 8; c++)
                if (m2 & (1u << c)) { if (o2 < COL_L2) s_list2[o2] = (uint16_t) (tid * 8u + c); o2++; }
            __syncthreads();
            // D. level-3: 27 samples per kept level-2 cell.  Few cells: 32 lanes per cell, 8 cells per pass (short dependent
            // chain).  Many cells (the heaviest columns set the kernel's duration): one lane per cell, its 27 samples z
            // innermost so the (x, y) terms are shared -- a flat ~1 700 instructions for up to 256 cells instead of 140 per 8.
            uint32_t local3 = 0;
            if (n2 > 96u) {
                for (uint32_t b0 = 0; b0 < n2; b0 += 256u) {
                    const uint32_t b = b0 + tid;
                    if (b < n2) {
                        const uint32_t e = s_list2[b], c2 = e & 7u;
                        uint32_t lx, ly, lz;
                        col_l1_index(s_list1[e >> 3], X0, Y0, Z0, lx, ly, lz);
                        lx = 2u * lx + (c2 >> 2); ly = 2u * ly + ((c2 >> 1) & 1u); lz = 2u * lz + (c2 & 1u);
                        uint32_t bits = 0;
#pragma unroll 1
                        for (uint32_t ix = 0; ix < 3; ix++) {
                            const float x = lat((int) (2u * lx + ix), h3);
#pragma unroll 1
                            for (uint32_t iy = 0; iy < 3; iy++) {
                                const float y = lat((int) (2u * ly + iy), h3);
#pragma unroll
                                for (uint32_t iz = 0; iz < 3; iz++)
                                    if (sdf<SCENE>(x, y, lat((int) (2u * lz + iz), h3), sp) <= 0.0f) bits |= 1u << (ix * 9u + iy * 3u + iz);
                            }
                        }
                        const uint32_t m = child_mask_from_bits(bits);
                        g_mask3[(size_t) unit * COL_L2 + b] = (uint8_t) m;
                        local3 += __popc(m);
                    }
                }
            } else {
                for (uint32_t b0 = 0; b0 < n2; b0 += 8u) {
                    const uint32_t b = b0 + grp;
                    uint32_t lx = 0, ly = 0, lz = 0;
                    if (b < n2) {
                        const uint32_t e = s_list2[b], c2 = e & 7u;
                        col_l1_index(s_list1[e >> 3], X0, Y0, Z0, lx, ly, lz);
                        lx = 2u * lx + (c2 >> 2); ly = 2u * ly + ((c2 >> 1) & 1u); lz = 2u * lz + (c2 & 1u);
                    }
                    const uint32_t m = col_cell_mask<SCENE>(b < n2, lx, ly, lz, h3, s, sp);
                    if (s == 0 && b < n2) { g_mask3[(size_t) unit * COL_L2 + b] = (uint8_t) m; local3 += __popc(m); }
                }
            }
            if (local3) atomicAdd(&s_n3, local3);
            __syncthreads();
            n3 = s_n3;
        }
    }
    if (tid < n1) {
        g_list1[(size_t) unit * COL_L1 + tid] = s_list1[tid];
        if (J >= 2) g_mask2[(size_t) unit * COL_L1 + tid] = s_mask2[tid];
    }
    if (tid == 0) colinfo[unit] = make_uint4(n1, n2, n3, 0u);
    // the regen's clears (control block, weld table), spread over the grid -- behind the column's own work, so the few
    // heavy columns that set the kernel's duration start at once and the many light ones do most of the clearing
    for (uint32_t i = blockIdx.x * 256u + tid; i < nzero; i += gridDim.x * 256u) zero[i] = 0u;
    const uint4 ones = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    for (uint32_t i = blockIdx.x * 256u + tid; i < nff16; i += gridDim.x * 256u) ff[i] = ones;
}

// Ordered emission of the level-J voxel list from the per-column lists (J = 1..3).  Every workgroup sums the counts of
// the columns in front of it (<= 1024 words); block 0 also publishes the level totals.  out == nullptr: totals only.
__global__ __launch_bounds__(256) void k_col_emit(uint32_t x0, uint32_t z0, uint32_t zs, uint32_t ncols, uint32_t J,
                                                   const uint8_t *__restrict__ g_list1, const uint8_t *__restrict__ g_mask2,
                                                   const uint8_t *__restrict__ g_mask3, const uint4 *__restrict__ colinfo,
                                                   Point *__restrict__ out, uint32_t cap, uint32_t *__restrict__ level_ctr /* [0..J] */,
                                                   uint32_t *__restrict__ status, uint32_t segs /* units per column; ncols = units */) {
    __shared__ uint8_t s_list1[COL_L1];
    __shared__ uint16_t s_list2[COL_L2];
    __shared__ uint32_t s_scan[17];
    const uint32_t F = MESH_GENERATION_INIT_FACTOR, tid = threadIdx.x;
    const float h0 = MESH_GENERATION_BB_SIZE / (float) MESH_GENERATION_INIT_FACTOR;
    const float hJ = J == 1 ? h0 / 2.0f : (J == 2 ? h0 / 4.0f : h0 / 8.0f);
    const uint32_t col = blockIdx.x /* = unit */, X0 = x0 + (col / segs) / F, Y0 = (col / segs) % F, Z0 = z0 + (col % segs) * (zs / segs);
    // offset of this unit in the level-J list; block 0 sums everything for the totals
    uint32_t before = 0, t1 = 0, t2 = 0, t3 = 0;
    for (uint32_t c = tid; c < ncols; c += 256u) {
        const uint4 ci = colinfo[c];
        const uint32_t nj = J == 1 ? ci.x : (J == 2 ? ci.y : ci.z);
        if (c < col) before += nj;
        t1 += ci.x; t2 += ci.y; t3 += ci.z;
    }
    uint32_t base, tot;
    (void) block_excl_scan(before, base, s_scan);
    if (blockIdx.x == 0) {
        (void) block_excl_scan(t1, tot, s_scan); const uint32_t T1 = tot;
        (void) block_excl_scan(t2, tot, s_scan); const uint32_t T2 = tot;
        (void) block_excl_scan(t3, tot, s_scan); const uint32_t T3 = tot;
        if (tid == 0) {
            uint32_t tj = J == 1 ? T1 : (J == 2 ? T2 : T3);
            if (out && tj > cap) { atomicOr(&status[ST_OVERFLOW], 1u); tj = cap; }
            level_ctr[0] = (ncols / segs) * zs;
            level_ctr[1] = J == 1 ? tj : T1;
            if (J >= 2) level_ctr[2] = J == 2 ? tj : T2;
            if (J >= 3) level_ctr[3] = tj;
        }
    }
    if (!out) return;
    const uint4 me = colinfo[col];
    const uint32_t n1 = me.x, n2 = me.y;
    if (n1 == 0) return;
    if (tid < n1) s_list1[tid] = g_list1[(size_t) col * COL_L1 + tid];
    __syncthreads();
    if (J == 1) {
        if (tid < n1 && base + tid < cap) {
            uint32_t lx, ly, lz;
            col_l1_index(s_list1[tid], X0, Y0, Z0, lx, ly, lz);
            out[base + tid] = Point{lat((int) lx, hJ), lat((int) ly, hJ), lat((int) lz, hJ)};
        }
        return;
    }
    const uint32_t m2 = tid < n1 ? g_mask2[(size_t) col * COL_L1 + tid] : 0u;
    uint32_t tot2;
    uint32_t o2 = block_excl_scan(__popc(m2), tot2, s_scan);
    if (J == 2) {
        if (m2) {
            uint32_t lx, ly, lz;
            col_l1_index(s_list1[tid], X0, Y0, Z0, lx, ly, lz);
            for (uint32_t c = 0; c < 8; c++)
                if (m2 & (1u << c)) {
                    if (base + o2 < cap) out[base + o2] = Point{lat((int) (2u * lx + (c >> 2)), hJ), lat((int) (2u * ly + ((c >> 1) & 1u)), hJ), lat((int) (2u * lz + (c & 1u)), hJ)};
                    o2++;
                }
        }
        return;
    }
    for (uint32_t c = 0; c < 8; c++)
        if (m2 & (1u << c)) { if (o2 < COL_L2) s_list2[o2] = (uint16_t) (tid * 8u + c); o2++; }
    __syncthreads();
    uint32_t run = base;
    for (uint32_t b0 = 0; b0 < n2; b0 += 256u) {
        const uint32_t b = b0 + tid;
        const uint32_t m3 = b < n2 ? g_mask3[(size_t) col * COL_L2 + b] : 0u;
        uint32_t tot3;
        uint32_t o3 = run + block_excl_scan(__popc(m3), tot3, s_scan);
        if (m3) {
            const uint32_t e = s_list2[b], c2 = e & 7u;
            uint32_t lx, ly, lz;
            col_l1_index(s_list1[e >> 3], X0, Y0, Z0, lx, ly, lz);
            lx = 2u * lx + (c2 >> 2); ly = 2u * ly + ((c2 >> 1) & 1u); lz = 2u * lz + (c2 & 1u);
            for (uint32_t c = 0; c < 8; c++)
                if (m3 & (1u << c)) {
                    if (o3 < cap) out[o3] = Point{lat((int) (2u * lx + (c >> 2)), hJ), lat((int) (2u * ly + ((c >> 1) & 1u)), hJ), lat((int) (2u * lz + (c & 1u)), hJ)};
                    o3++;
                }
        }
        run += tot3;
    }
}

// Raw 8n layout of the reference kernel (INFINITY for non-border children)
__global__ __launch_bounds__(256) void k_refine_expand_raw(const Point *__restrict__ vox, uint32_t n, Point cs,
                                                            const uint8_t *__restrict__ mask, Point *__restrict__ out) {
    const uint32_t total = n * 8u;
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < total; s += gridDim.x * blockDim.x) {
        const uint32_t p = s >> 3, c = s & 7u;
        Point q;
        if (mask[p] & (1u << c)) {
            const Point b = vox[p];
            q.x = b.x + (float) (c >> 2) * cs.x;
            q.y = b.y + (float) ((c >> 1) & 1) * cs.y;
            q.z = b.z + (float) (c & 1) * cs.z;
        } else {
            q.x = q.y = q.z = __builtin_inff();
        }
        out[s] = q;
    }
}

// ------------------------------------------------------------------------------------------
// A4 mesh stage 1: cube index per voxel + triangle count per 256-voxel tile (no atomics)
// ------------------------------------------------------------------------------------------
template <int SCENE>
__global__ __launch_bounds__(256) void k_classify(const Point *__restrict__ vox, const uint32_t *__restrict__ n_ptr,
                                                   Point vs, const SceneParams *__restrict__ spp, uint8_t *__restrict__ cubeidx,
                                                   uint32_t *__restrict__ tilecnt) {
    const SceneParams &sp = *spp;  // scene uniform in device memory (a captured graph replays with new values)
    __shared__ uint32_t s_cnt;
    const uint32_t n = *n_ptr;
    const uint32_t ntiles = (n + TILE - 1) / TILE;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        const uint32_t v = tile * TILE + threadIdx.x;
        uint32_t ntri = 0;
        if (v < n) {
            const Point b = vox[v];
            uint32_t ci = 0;
#pragma unroll 1
            for (int cz = 0; cz < 2; cz++) {
                const float z = b.z + (cz ? vs.z : 0.0f);
#pragma unroll
                for (int cq = 0; cq < 4; cq++) {
                    const float x = b.x + (cox(cq) ? vs.x : 0.0f);
                    const float y = b.y + (coy(cq) ? vs.y : 0.0f);
                    if (sdf<SCENE>(x, y, z, sp) <= 0.0f) ci |= 1u << (cz * 4 + cq);
                }
            }
            cubeidx[v] = (uint8_t) ci;
            ntri = MC_TRI_COUNT[ci];
        }
        const uint32_t ws = wave_sum(ntri);
        if (lane_id() == 0 && ws) atomicAdd(&s_cnt, ws);
        __syncthreads();
        if (threadIdx.x == 0) tilecnt[tile] = s_cnt;
        __syncthreads();
    }
}

// triangle count per tile when the cube indices are already known (dense mode)
__global__ __launch_bounds__(256) void k_tile_tricount(const uint8_t *__restrict__ cubeidx, const uint32_t *__restrict__ n_ptr,
                                                        uint32_t *__restrict__ tilecnt) {
    __shared__ uint32_t s_cnt;
    const uint32_t n = *n_ptr;
    const uint32_t ntiles = (n + TILE - 1) / TILE;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        const uint32_t v = tile * TILE + threadIdx.x;
        const uint32_t ws = wave_sum(v < n ? (uint32_t) MC_TRI_COUNT[cubeidx[v]] : 0u);
        if (lane_id() == 0 && ws) atomicAdd(&s_cnt, ws);
        __syncthreads();
        if (threadIdx.x == 0) tilecnt[tile] = s_cnt;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// A4-A7 mesh stage 2.  Work is split by what bounds it:
//   k_tile_dedupe   per 256-voxel tile, LDS hash: tile-unique lattice edges, corner->vertex table
//                   (consecutive voxels are spatial neighbours, so most of the <=4 cubes sharing
//                   an edge sit in one tile; the remaining cross-tile duplicates carry bit-identical
//                   positions and are merged by the weld).  No SDF evaluation.
//   k_project       all tile-unique edges of the whole field, globally balanced: 4 lanes per edge
//                   (lane a<3 = 4-point stencil of axis a, lane 3 = centre sample), 16 edges in
//                   flight per wave, finished groups refill from the wave's queue.
//   k_flip          all triangles, 4 lanes each: centroid normal + orientation test.
//   k_tile_weld_*   per tile again: first-seen ordinals in LDS, weld insert, rank, assign.
// ------------------------------------------------------------------------------------------
static constexpr uint32_t T_HASH = 2048;     // LDS hash slots; a tile with more than T_UMAX unique edges takes the direct path
static constexpr uint32_t T_UMAX = 1536;     // (a 256-voxel surface patch has ~300; 3072 is the theoretical maximum)
static constexpr uint32_t T_TRI = 1280;      // 5 triangles x 256 voxels

// value of lane (quad base + K) of the caller's 4-lane group (DPP quad_perm broadcast, no LDS traffic)
template <int K>
__device__ __forceinline__ float quad_bcast(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), K * 0x55, 0xf, 0xf, true));
}

// 4th-order stencil of empirical_normal over a 4-lane group: lane `sub` owns ONE stencil offset (+2e, +e, -e, -2e)
// and evaluates it along x, then y, then z.  Every sample differs from the centre g in one coordinate only and all
// 64 lanes vary the SAME coordinate at a time, so after inlining the compiler shares the per-axis terms of the other
// two coordinates between the centre sample and the axis samples (sd_obj: 302 instead of 372 instructions per Newton
// round).  Sample positions (g.x + off ...) and the combination order ((-f(+2e) + 8 f(+e)) - 8 f(-e)) + f(-2e) are
// those of empirical_normal (signed_distance.cu:179-202).  `centre` = f(g) when WANT_CENTRE.
template <int SCENE, bool WANT_CENTRE>
__device__ __forceinline__ f3 stencil_quad(const f3 g, uint32_t sub, const SceneParams &sp, float &centre) {
    const float e1 = SDM_NORMAL_EPS, e2 = 2.0f * SDM_NORMAL_EPS;
    const float off = sub == 0 ? e2 : (sub == 1 ? e1 : (sub == 2 ? -e1 : -e2));
    if (WANT_CENTRE) centre = sdf<SCENE>(g.x, g.y, g.z, sp);
    const float fx = sdf<SCENE>(g.x + off, g.y, g.z, sp);
    const float fy = sdf<SCENE>(g.x, g.y + off, g.z, sp);
    const float fz = sdf<SCENE>(g.x, g.y, g.z + off, sp);
    f3 d;
    d.x = (-quad_bcast<0>(fx) + 8.0f * quad_bcast<1>(fx) - 8.0f * quad_bcast<2>(fx) + quad_bcast<3>(fx));
    d.y = (-quad_bcast<0>(fy) + 8.0f * quad_bcast<1>(fy) - 8.0f * quad_bcast<2>(fy) + quad_bcast<3>(fy));
    d.z = (-quad_bcast<0>(fz) + 8.0f * quad_bcast<1>(fz) - 8.0f * quad_bcast<2>(fz) + quad_bcast<3>(fz));
    return d;
}

// value held by the other lane of the caller's 2-lane group (DPP quad_perm [1, 0, 3, 2])
__device__ __forceinline__ float pair_swap(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
}

// The same stencil over a 2-lane group: lane 0 owns the offsets (+2e, +e), lane 1 owns (-e, -2e), both evaluate their two
// samples along x, then y, then z, plus the centre.  The two samples of an axis and the centre share the terms of the
// other two coordinates (like the one-lane layout), so the instruction count per vertex stays close to the one-lane
// kernel's while the dependent work per Newton round -- what the slowest wave's lifetime is made of -- is about half.
template <int SCENE>
__device__ __forceinline__ f3 stencil_pair(const f3 g, uint32_t sub, const SceneParams &sp, float &centre) {
    const float e1 = SDM_NORMAL_EPS, e2 = 2.0f * SDM_NORMAL_EPS;
    const float oa = sub == 0 ? e2 : -e1, ob = sub == 0 ? e1 : -e2;
    centre = sdf<SCENE>(g.x, g.y, g.z, sp);
    const float xa = sdf<SCENE>(g.x + oa, g.y, g.z, sp), xb = sdf<SCENE>(g.x + ob, g.y, g.z, sp);
    const float ya = sdf<SCENE>(g.x, g.y + oa, g.z, sp), yb = sdf<SCENE>(g.x, g.y + ob, g.z, sp);
    const float za = sdf<SCENE>(g.x, g.y, g.z + oa, sp), zb = sdf<SCENE>(g.x, g.y, g.z + ob, sp);
    const bool lo = sub == 0;
    f3 d;
    {
        const float pa = pair_swap(xa), pb = pair_swap(xb);
        d.x = (-(lo ? xa : pa) + 8.0f * (lo ? xb : pb) - 8.0f * (lo ? pa : xa) + (lo ? pb : xb));
    }
    {
        const float pa = pair_swap(ya), pb = pair_swap(yb);
        d.y = (-(lo ? ya : pa) + 8.0f * (lo ? yb : pb) - 8.0f * (lo ? pa : ya) + (lo ? pb : yb));
    }
    {
        const float pa = pair_swap(za), pb = pair_swap(zb);
        d.z = (-(lo ? za : pa) + 8.0f * (lo ? zb : pb) - 8.0f * (lo ? pa : za) + (lo ? pb : zb));
    }
    return d;
}

// tile descriptor: x = first vertex id, y = vertex count, z = first triangle, w = triangle count
__global__ __launch_bounds__(256) void k_tile_dedupe(const Point *__restrict__ vox, const uint32_t *__restrict__ n_ptr, Point vs,
                                                      const uint8_t *__restrict__ cubeidx, const uint32_t *__restrict__ tilecnt,
                                                      uint32_t *__restrict__ trioff, uint32_t *__restrict__ tri_e,
                                                      Vertex *__restrict__ uverts, uint4 *__restrict__ tdesc, uint32_t ucap,
                                                      uint32_t tcap, uint32_t *__restrict__ ntri_out,
                                                      uint32_t *__restrict__ ntiles_out, uint32_t *__restrict__ status,
                                                      uint64_t *__restrict__ dbg /* optional: 8 s_memtime stamps per tile (tools/tile_phases.py) */,
                                                      uint32_t prescanned /* tilecnt = exclusive prefix, *ntri_out = total (k_scan_tiles in front) */,
                                                      uint32_t *__restrict__ colstart /* optional (multi-GPU z-slabs): [c] = 1 + first triangle of level-0 (x0, y0) column c, for the columns that hold voxels; pre-zeroed */,
                                                      uint32_t *__restrict__ coltile /* with colstart: [c] = the tile that holds the column's first voxel */,
                                                      uint32_t colshift /* log2(voxels per level-0 cell edge) */) {
#define SDM_STAMP(k) do { if (dbg && tid == 0) dbg[8ull * tile + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
    __shared__ uint64_t s_keys[T_HASH];
    __shared__ uint64_t s_aux[2 * TILE];                    // insert phase: per-voxel lattice coordinates + triangle row;
    uint64_t *const s_vpack = s_aux, *const s_vrow = s_aux + TILE;
    uint16_t *const s_lid = reinterpret_cast<uint16_t *>(s_aux);      // afterwards: local vertex id per hash slot (T_HASH x u16 = the same 4 KB)
    __shared__ uint16_t s_tri[T_TRI * 3];
    __shared__ uint16_t s_vlo[TILE];
    __shared__ uint8_t s_tvox[T_TRI];
    __shared__ uint16_t s_col[TILE];
    __shared__ uint32_t s_scan[17];
    __shared__ uint32_t s_bad, s_full;
    static_assert(sizeof(uint64_t) * 2 * TILE == sizeof(uint16_t) * T_HASH, "s_aux is reused for the local ids");
    static_assert(T_HASH == 2048, "the cheap tile hash keeps 11 bits");

    const uint32_t n = *n_ptr;
    const uint32_t ntiles = (n + TILE - 1) / TILE;
    const uint32_t tid = threadIdx.x;
    if (blockIdx.x == 0 && tid == 0) {
        *ntiles_out = ntiles;
        if (n == 0 && !prescanned) *ntri_out = 0;
    }
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // issue every independent global load before the first barrier so their latencies overlap
        SDM_STAMP(0);
        const uint32_t v = tile * TILE + tid;
        const uint32_t ci = v < n ? cubeidx[v] : 0u;
        const Point b = vox[v < n ? v : (n - 1)];
        const float bpx = (colstart && tid == 0 && tile > 0) ? vox[v - 1u].x : 0.0f, bpy = (colstart && tid == 0 && tile > 0) ? vox[v - 1u].y : 0.0f;
        uint32_t part = 0;
        if (!prescanned) for (uint32_t i = tid; i < tile; i += 256) part += tilecnt[i];
        const uint32_t ntri = v < n ? (uint32_t) MC_TRI_COUNT[ci] : 0u;
        const uint64_t tri_row = MC_TRI_PACKED[ci];
        uint32_t tbase;
        if (prescanned) tbase = tilecnt[tile];
        else (void) block_excl_scan(part, tbase, s_scan);

        int mx = 0, my = 0, mz = 0;
        bool lat_ok = true;
        if (ntri) {
            const bool okx = lattice_coord(b.x, vs.x, mx), oky = lattice_coord(b.y, vs.y, my), okz = lattice_coord(b.z, vs.z, mz);
            lat_ok = okx && oky && okz;
        }
        // (multi-GPU z-slabs) a level-0 column is a contiguous range of the list: the voxel whose column differs from its
        // predecessor's publishes where the column's triangles start -- a plain store with one writer per column.  (Atomic adds of the
        // triangle counts from here cost the kernel 10 us at 512^3: ~14 k device-scope atomics.)  Column from the lattice coordinates:
        // 2^colshift voxels per level-0 cell edge.
        uint32_t col = 0xffffu;
        if (colstart && v < n) {
            int cx = mx, cy = my;
            if (!ntri) { (void) lattice_coord(b.x, vs.x, cx); (void) lattice_coord(b.y, vs.y, cy); }
            col = (((uint32_t) cx >> colshift) & 31u) * MESH_GENERATION_INIT_FACTOR + (((uint32_t) cy >> colshift) & 31u);
            s_col[tid] = (uint16_t) col;
        }
        SDM_STAMP(1);
        uint32_t ttri;
        const uint32_t lo = block_excl_scan(ntri, ttri, s_scan);
        SDM_STAMP(2);
        if (v < n) trioff[v] = tbase + lo;
        if (colstart && v < n) {          // (s_col: the scan above has synchronised)
            uint32_t prev = 0xffffu;
            if (tid) prev = s_col[tid - 1u];
            else if (tile) { int px = 0, py = 0; (void) lattice_coord(bpx, vs.x, px); (void) lattice_coord(bpy, vs.y, py);
                             prev = (((uint32_t) px >> colshift) & 31u) * MESH_GENERATION_INIT_FACTOR + (((uint32_t) py >> colshift) & 31u); }
            if (col != prev) { colstart[col] = tbase + lo + 1u; coltile[col] = tile; }
        }
        if (tile == ntiles - 1 && tid == 0 && !prescanned) {
            uint32_t tot = tbase + ttri;
            if (tot > tcap) { atomicOr(&status[ST_OVERFLOW], 1u); tot = tcap; }
            *ntri_out = tot;
        }
        for (uint32_t i = tid; i < T_HASH; i += 256) s_keys[i] = KEY_EMPTY;
        if (tid == 0) { s_bad = 0; s_full = 0; }
        __syncthreads();
        if (!lat_ok) { s_bad = 1; atomicOr(&status[ST_NOT_LATTICE], 1u); }
        SDM_STAMP(3);

        // Insert every triangle corner's edge key; remember the slot per corner.  The corners are dealt evenly to the threads
        // (corner i of the tile -> thread i % 256): a voxel has 3 to 15 corners, and with one voxel per thread every wave
        // ran its insert loop 12-15 times at ~40 % lane use -- 10 000 of the tile's 22 000 cycles (tools/tile_phases.py).
        // The voxel's lattice coordinates and its triangle row go through LDS (the arrays that later hold the local ids).
        if (ntri) {
            s_vpack[tid] = (uint64_t) (uint32_t) mx | ((uint64_t) (uint32_t) my << 21) | ((uint64_t) (uint32_t) mz << 42);
            s_vrow[tid] = tri_row;
            s_vlo[tid] = (uint16_t) lo;
            for (uint32_t j = 0; j < ntri; j++) s_tvox[lo + j] = (uint8_t) tid;
        }
        __syncthreads();
        if (s_bad == 0) {
            for (uint32_t i = tid; i < 3u * ttri; i += 256u) {
                const uint32_t t = i / 3u, q = i - 3u * t, vv = s_tvox[t];
                const uint64_t vp = s_vpack[vv];
                const int e = (int) ((s_vrow[vv] >> (4u * (3u * (t - s_vlo[vv]) + q))) & 0xFull);
                const uint64_t key = edge_key((int) (vp & 0x1FFFFFull), (int) ((vp >> 21) & 0x1FFFFFull), (int) ((vp >> 42) & 0x1FFFFFull), e);
                // cheap home slot: the keys of a tile are a few hundred neighbouring lattice edges; three 32-bit multiplies
                // spread them as well as the 64-bit mixer did (checked by the overflow statistics: none at any size)
                uint32_t sl = ((uint32_t) key * 0x9E3779B1u ^ (uint32_t) (key >> 21) * 0x85EBCA77u ^ (uint32_t) (key >> 42) * 0xC2B2AE3Du) >> (32 - 11);
                bool placed = false;
                for (uint32_t probe = 0; probe < T_HASH && !placed; probe++) {
                    const uint64_t prev = atomicCAS((unsigned long long *) &s_keys[sl], (unsigned long long) KEY_EMPTY, (unsigned long long) key);
                    if (prev == KEY_EMPTY || prev == key) placed = true;
                    else sl = hash_next(sl, T_HASH);
                }
                if (!placed) s_full = 1;
                s_tri[i] = (uint16_t) sl;
            }
        }
        __syncthreads();
        SDM_STAMP(4);
        // dense local ids in slot order (deterministic)
        uint32_t occ = 0;
#pragma unroll
        for (uint32_t k = 0; k < T_HASH / 256; k++) occ += s_keys[tid * (T_HASH / 256) + k] != KEY_EMPTY ? 1u : 0u;
        uint32_t nu;
        uint32_t lid = block_excl_scan(occ, nu, s_scan);
        SDM_STAMP(5);
        // vertex slots live in corner-ordinal space: tile t owns [3 * tbase, 3 * (tbase + ttri)), of which
        // the first nu <= 3 * ttri are used -- no global allocation, no atomics
        const uint32_t vbase = 3u * tbase;
        const bool tile_ok = s_bad == 0;
        if (tile_ok && (s_full || nu > T_UMAX) && tid == 0) atomicOr(&status[ST_TILE_OVERFLOW], 1u);
        const bool fits = tile_ok && !s_full && nu <= T_UMAX && (vbase + nu <= ucap) && (tbase + ttri <= tcap);
        if (tile_ok && !fits && tid == 0) atomicOr(&status[ST_OVERFLOW], 2u);
        if (tid == 0) tdesc[tile] = fits ? make_uint4(vbase, nu, tbase, ttri) : make_uint4(0u, 0u, tbase, 0u);
        if (fits) {
#pragma unroll
            for (uint32_t k = 0; k < T_HASH / 256; k++) {
                const uint32_t s = tid * (T_HASH / 256) + k;
                const uint64_t key = s_keys[s];
                if (key != KEY_EMPTY) {
                    const int kx = (int) (key & 0x1FFFFF), ky = (int) ((key >> 21) & 0x1FFFFF), kz = (int) ((key >> 42) & 0x1FFFFF);
                    // midpoint = a*0.5 + b*0.5 of the two lattice corners (marching_cubes.cu:13-16)
                    Point m;
                    m.x = lat(kx >> 1, vs.x) * 0.5f + lat((kx + 1) >> 1, vs.x) * 0.5f;
                    m.y = lat(ky >> 1, vs.y) * 0.5f + lat((ky + 1) >> 1, vs.y) * 0.5f;
                    m.z = lat(kz >> 1, vs.z) * 0.5f + lat((kz + 1) >> 1, vs.z) * 0.5f;
                    uverts[vbase + lid].position = m;
                    s_lid[s] = (uint16_t) lid;
                    lid++;
                }
            }
        }
        __syncthreads();
        SDM_STAMP(6);
        if (fits)
            for (uint32_t i = tid; i < 3 * ttri; i += 256)
                if (tbase + i / 3u < tcap) tri_e[3u * tbase + i] = vbase + s_lid[s_tri[i]];
        __syncthreads();
        SDM_STAMP(7);
    }
#undef SDM_STAMP
}

// ------------------------------------------------------------------------------------------
// A8 weld (src/cuda/mod.rs:263-296).
//   key  = 3 x 21-bit biased round(x * 1e5)  (covers +-10.48 units; the bounding cube is +-2.5)
//   claim: one CAS per pre-weld vertex; the claimer is the "winner" (bit 31 of its vslot entry), everyone
//   else who finds the key present is a "loser".  Later the winner stores its first-seen ordinal with a
//   plain store (wfm[slot].x) and only losers pay an atomicMin (wfm[slot].y): min(x, y) is the key's first-seen
//   ordinal whatever the arrival order.
// ------------------------------------------------------------------------------------------
static constexpr uint32_t VS_WINNER = 0x80000000u;
static constexpr uint32_t VS_SLOT = 0x7fffffffu;

// Locality-preserving home slot of a weld key (option "weld_shift" > 0; measured and NOT the default): the table is cut
// into blocks of WELD_BLOCK slots, the BLOCK is chosen by hashing the key's coarse cell (the low `shift` bits of each
// quantised coordinate dropped), the slot inside the block by hashing the whole key, so the vertices of a 256-voxel tile
// share a few KB of table.  Any block / shift gives the same weld.  It loses badly: the claim's compare-and-swaps then
// hammer a few cache lines at a time and atomics to one line serialise (512^3: claim 25 -> 157 us at 8-voxel cells;
// milliseconds once a cell holds more keys than a block).  What the locality was wanted for -- the per-slot gathers of
// the later weld kernels -- is obtained instead by indexing those arrays with the key's representative VERTEX (below).

static constexpr uint32_t WELD_BLOCK = 256;

__device__ __forceinline__ uint32_t weld_home(uint64_t key, uint64_t coarse, uint32_t wsize) {
    const uint32_t nblocks = wsize / WELD_BLOCK;
    if (nblocks == 0) return hash_home(key, wsize);
    const uint32_t block = (uint32_t) (((mix64(coarse) >> 32) * (uint64_t) nblocks) >> 32);
    return block * WELD_BLOCK + (uint32_t) (mix64(key) >> 40) % WELD_BLOCK;
}

__device__ __forceinline__ uint32_t weld_claim(const Point p, uint64_t *__restrict__ wkeys, uint32_t wsize,
                                               uint32_t *__restrict__ status, uint32_t shift) {
    const long long qx = weld_quant(p.x) + 1048576LL, qy = weld_quant(p.y) + 1048576LL, qz = weld_quant(p.z) + 1048576LL;
    if (((unsigned long long) qx | (unsigned long long) qy | (unsigned long long) qz) >> 21) {
        atomicOr(&status[ST_WELD_RANGE], 1u);
        return U32_NONE;
    }
    const uint64_t key = (uint64_t) qx | ((uint64_t) qy << 21) | ((uint64_t) qz << 42);
    const uint64_t low = (1ull << shift) - 1ull;
    uint32_t s = shift ? weld_home(key, key & ~(low | (low << 21) | (low << 42)), wsize) : hash_home(key, wsize);
    for (uint32_t probe = 0; probe < wsize; probe++) {
        const uint64_t prev = atomicCAS((unsigned long long *) &wkeys[s], (unsigned long long) KEY_EMPTY, (unsigned long long) key);
        if (prev == KEY_EMPTY) return s | VS_WINNER;
        if (prev == key) return s;
        s = hash_next(s, wsize);
    }
    atomicOr(&status[ST_OVERFLOW], 8u);
    return U32_NONE;
}

// The reference's key is [i64; 3] (src/cuda/mod.rs:268-273), unbounded.  When a vertex does not fit the packed 3 x 21-bit
// key the host re-runs the weld in WIDE mode: a slot then holds the id of the vertex that claimed it (its "owner") and
// two keys are equal iff the quantised triples of the two POSITIONS are (every position was written by an earlier
// kernel).  Same first-seen semantics, any coordinate the i64 cast can express (NaN -> 0, +-inf saturate, as `as i64`).
__device__ __forceinline__ uint32_t weld_claim_wide(const Vertex *__restrict__ uverts, uint32_t id, uint64_t *__restrict__ wkeys,
                                                    uint32_t wsize, uint32_t *__restrict__ status, uint32_t shift) {
    const Point p = uverts[id].position;
    const long long qx = weld_quant(p.x), qy = weld_quant(p.y), qz = weld_quant(p.z);
    const uint64_t hsh = mix64((uint64_t) qx * 0x9E3779B97F4A7C15ull ^ mix64((uint64_t) qy) ^ (mix64((uint64_t) qz) << 1));
    const uint64_t chs = mix64((uint64_t) (qx >> shift) * 0x9E3779B97F4A7C15ull ^ mix64((uint64_t) (qy >> shift)) ^ (mix64((uint64_t) (qz >> shift)) << 1));
    uint32_t s = shift ? weld_home(hsh, chs, wsize) : (uint32_t) (((hsh >> 32) * (uint64_t) wsize) >> 32);
    for (uint32_t probe = 0; probe < wsize; probe++) {
        const uint64_t prev = atomicCAS((unsigned long long *) &wkeys[s], (unsigned long long) KEY_EMPTY, (unsigned long long) id);
        if (prev == KEY_EMPTY) return s | VS_WINNER;
        const Point o = uverts[(uint32_t) prev].position;
        if (weld_quant(o.x) == qx && weld_quant(o.y) == qy && weld_quant(o.z) == qz) return s;
        s = hash_next(s, wsize);
    }
    atomicOr(&status[ST_OVERFLOW], 8u);
    return U32_NONE;
}

// exclusive prefix of the per-tile unique-vertex counts (dense work index for k_project); total -> *nu_out
__global__ __launch_bounds__(1024) void k_tile_prefix(const uint4 *__restrict__ tdesc, const uint32_t *__restrict__ ntiles_ptr,
                                                       uint32_t *__restrict__ tpre, uint32_t *__restrict__ nu_out,
                                                       uint32_t *__restrict__ coarse) {
    // coarse[b] = tile that holds work item 64 b: k_project finds an item's tile by walking forward from there (0..2
    // steps) instead of a 10-12 step binary search of dependent loads on every refill
    __shared__ uint32_t lds[17];
    const uint32_t ntiles = *ntiles_ptr;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < ntiles; base += blockDim.x) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < ntiles ? tdesc[i].y : 0u;
        uint32_t total;
        const uint32_t ex = block_excl_scan(v, total, lds);
        if (i < ntiles) {
            const uint32_t start = carry + ex, end = start + v;
            tpre[i] = start;
            for (uint32_t b = (start + 63u) >> 6; (b << 6) < end; b++) coarse[b] = i;
        }
        carry += total;
    }
    if (threadIdx.x == 0) { tpre[ntiles] = carry; *nu_out = carry; }
}

// optional bookkeeping of the projection kernels (option "profile" / tools): Newton rounds per wave into one of 256
// counters on separate cache lines (a single address would serialise at ~88 atomics per us), and the wave's timeline
__device__ __forceinline__ void project_epilogue(uint32_t wave, uint32_t rounds, uint64_t t_start, uint32_t *__restrict__ prof,
                                                 uint64_t *__restrict__ diag) {
    if (lane_id() != 0) return;
    if (prof) atomicAdd(&prof[(wave & 255u) * 16u], rounds);
    if (diag) {
        // per wave: {s_memtime at start (shader clock, per XCD), at end, rounds | XCC_ID << 28 | HW_ID << 32,
        //            s_memrealtime (constant 100 MHz) at start | at end << 32}
        diag[4ull * wave + 0] = t_start & 0x00FFFFFFFFFFFFFFull;
        diag[4ull * wave + 1] = __builtin_amdgcn_s_memtime() & 0x00FFFFFFFFFFFFFFull;
        diag[4ull * wave + 2] = (uint64_t) (rounds & 0x0FFFFFFFu) | ((uint64_t) (__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFu) << 28) |
                                ((uint64_t) __builtin_amdgcn_s_getreg((31 << 11) | 4) << 32);
        diag[4ull * wave + 3] = (diag[4ull * wave + 3] & 0xFFFFFFFFull) | ((__builtin_amdgcn_s_memrealtime() & 0xFFFFFFFFull) << 32);
    }
}

// A5/A6: closest_surface_point + vertex normal for every unique vertex; position read from and written
// back to uverts[].position.  16 items in flight per wave (4 lanes each); finished groups refill from
// the wave's round-robin share of the U work items; item i is mapped to its vertex slot with a binary
// search in the tile prefix.
// Weld claims of a projection wave's own vertices, issued when the wave has finished all of them (option "claim" = 1,
// packed-key form only).  The claim is bound by the chip's scattered-atomic rate (~20 G/s: 16 us for 323 k vertices at
// 512^3) and used to run beside the orientation test, which is half as long; here the waves that finish early claim
// while the slow ones still iterate, i.e. under the projection's tail, when most of the chip is idle.
// The wave re-derives its items (item -> tile by the coarse index) and re-reads the positions it wrote (agent-scope
// loads: the stores are this wave's own and were waited for).
// vertex slot of work item `item` of the projection: its tile by the workgroup's own LDS prefix (tpre == nullptr) or by
// k_tile_prefix's global prefix and coarse index
__device__ __forceinline__ uint32_t item_slot(uint32_t item, const uint4 *__restrict__ tdesc, const uint32_t *__restrict__ tpre,
                                              const uint32_t *__restrict__ coarse, const uint32_t *__restrict__ s_pre, uint32_t ntiles) {
    if (tpre == nullptr) {
        const uint32_t lo = local_find(s_pre, ntiles, item);
        return tdesc[lo].x + (item - s_pre[lo]);
    }
    uint32_t lo = coarse[item >> 6];                               // tile of item 64 * (item / 64): at or before ours
    while (tpre[lo + 1u] <= item) lo++;                            // (tpre[ntiles] = U > item ends the walk)
    return tdesc[lo].x + (item - tpre[lo]);
}

struct ClaimArgs {
    uint64_t *wkeys;
    uint32_t *vslot;
    uint32_t *wrep;
    uint2 *wfm;
    uint32_t wsize;
    uint32_t wmode;      // bit 0: wide (never set here), bits 8..: coarse shift; bit 1: claims enabled
    uint32_t bsh;        // k_project1: log2 of the dealing unit (runs of 2^bsh consecutive vertices go round robin to the waves)
};

__device__ __forceinline__ void claim_epilogue(uint32_t bsh, uint32_t wave, uint32_t nwaves, uint32_t myb, uint32_t U, const Vertex *__restrict__ uverts,
                                               const uint4 *__restrict__ tdesc, const uint32_t *__restrict__ tpre,
                                               const uint32_t *__restrict__ coarse, const ClaimArgs &ca, uint32_t *__restrict__ status,
                                               const uint32_t *__restrict__ s_pre, uint32_t ntiles) {
    if (!(ca.wmode & 2u)) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's position stores have left the CU
    const uint32_t shift = ca.wmode >> 8;
    for (uint32_t j = lane_id(); j < (myb << bsh); j += 64u) {
        const uint32_t item = (((j >> bsh) * nwaves + wave) << bsh) + (j & ((1u << bsh) - 1u));
        if (item >= U) continue;
        const uint32_t e = item_slot(item, tdesc, tpre, coarse, s_pre, ntiles);
        const float *pp = &uverts[e].position.x;
        Point p;
        p.x = __hip_atomic_load(pp + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        p.y = __hip_atomic_load(pp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        p.z = __hip_atomic_load(pp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t vs = weld_claim(p, ca.wkeys, ca.wsize, status, shift);
        ca.vslot[e] = vs;
        if (vs != U32_NONE && (vs & VS_WINNER)) {
            ca.wrep[vs & VS_SLOT] = e;
            ca.wfm[e] = make_uint2(U32_NONE, U32_NONE);
        }
    }
}

template <int SCENE, int LANES /* 4: lane = stencil offset; 2: lane = offset pair */>
__global__ __launch_bounds__(256) void k_project(Vertex *__restrict__ uverts, const uint4 *__restrict__ tdesc,
                                                  const uint32_t *__restrict__ tpre, const uint32_t *__restrict__ coarse, const uint32_t *__restrict__ ntiles_ptr,
                                                  const SceneParams *__restrict__ spp, uint32_t *__restrict__ status,
                                                  uint32_t *__restrict__ prof /* optional: 256 counters of Newton rounds, 64 B apart */,
                                                  uint64_t *__restrict__ diag /* optional: per wave {start, end, rounds, hw id} */, ClaimArgs ca,
                                                  uint32_t *__restrict__ nu_out /* tpre == nullptr: the vertex total goes here */) {
    __shared__ uint32_t s_pre[LOCAL_TILES + 1], s_scan[17];
    const SceneParams &sp = *spp;  // scene uniform in device memory (a captured graph replays with new values)
    const uint32_t ntiles = *ntiles_ptr;
    uint32_t U;
    if (tpre == nullptr) {
        if (ntiles > LOCAL_TILES) { if (threadIdx.x == 0) atomicOr(&status[ST_INTERNAL], 32u); return; }
        U = local_prefix(&tdesc[0].y, 4u, ntiles, s_pre, s_scan);
        if (blockIdx.x == 0 && threadIdx.x == 0) *nu_out = U;
    } else {
        U = ntiles ? tpre[ntiles] : 0u;
    }
    constexpr uint32_t PER = 64u / LANES, SH = LANES == 4 ? 4u : 5u;      // vertices per wave-batch, log2 of it
    const uint32_t lane = lane_id(), sub = lane & (LANES - 1u), gbase = lane & ~(LANES - 1u);
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    // Work items are dealt to the waves in batches of PER (one per lane group), round robin: iteration
    // counts are spatially correlated (slow convergence near the smooth-min blends), so a contiguous run
    // per wave leaves half the waves idle behind the slow ones (measured: waves alive 46 % of the kernel).
    // The wave's j-th item is item ((j / PER) * nwaves + wave) * PER + j % PER.
    const uint32_t nbatch = (U + PER - 1u) >> SH;
    if (wave >= nbatch) return;
    const uint64_t t_start = diag ? __builtin_amdgcn_s_memtime() : 0ull;
    if (diag && lane == 0) diag[4ull * wave + 3] = __builtin_amdgcn_s_memrealtime() & 0xFFFFFFFFull;
    uint32_t rounds = 0;                                              // vertex-rounds this wave evaluated (scalar bookkeeping)
    const uint32_t myb = (nbatch - wave - 1u) / nwaves + 1u;          // batches owned by this wave
    uint32_t qhead = 0;                                               // items handed out so far
    const uint32_t qend = myb * PER;
    const uint64_t lt_leaders = (LANES == 4 ? 0x1111111111111111ull : 0x5555555555555555ull) & ((1ull << gbase) - 1ull);   // leader lanes below this group
    bool have = false, finalize = false;
    uint32_t slot = 0;
    int it = 0;
    f3 g = mk(0.0f, 0.0f, 0.0f);
    for (;;) {
        const uint64_t need = __ballot(!have && sub == 0);
        const uint32_t avail = qend - qhead;
        if (need != 0ull && avail != 0u) {
            const uint32_t rank = __popcll(need & lt_leaders);
            const uint32_t j = qhead + rank;
            const uint32_t item = ((j >> SH) * nwaves + wave) * PER + (j & (PER - 1u));
            if (!have && rank < avail && item < U) {
                slot = item_slot(item, tdesc, tpre, coarse, s_pre, ntiles);
                const Point p = uverts[slot].position;
                g = mk(p.x, p.y, p.z);
                have = true; finalize = false; it = 0;
            }
            const uint32_t want = __popcll(need);
            qhead += want < avail ? want : avail;
        }
        const uint64_t act = __ballot(have && sub == 0);
        if (act == 0ull) break;
        rounds += __popcll(act);
        if (have) {
            float sd;
            const f3 nn = normalize3(LANES == 4 ? stencil_quad<SCENE, true>(g, sub, sp, sd) : stencil_pair<SCENE>(g, sub, sp, sd));
            if (!finalize) {
                g = sub3(g, scale3(nn, sd));
                it++;
                if (fabsf(sd) <= 0.00001f) finalize = true;
                else if (it >= SDM_MAX_NEWTON) { finalize = true; if (sub == 0) atomicOr(&status[ST_NEWTON_CAPPED], 1u); }
            } else {
                if (sub == 0) {
                    Vertex o;
                    o.position.x = g.x; o.position.y = g.y; o.position.z = g.z;
                    o.normal.x = nn.x; o.normal.y = nn.y; o.normal.z = nn.z;
                    uverts[slot] = o;
                    if (!((g.x - g.x) == 0.0f)) atomicOr(&status[ST_NONFINITE], 1u);   // src/cuda/mod.rs:289 needs the direct path
                }
                have = false;
            }
        }
    }
    claim_epilogue(SH, wave, nwaves, myb, U, uverts, tdesc, tpre, coarse, ca, status, s_pre, ntiles);
    project_epilogue(wave, rounds, t_start, prof, diag);
}

// A7 orientation test (compute_mesh_generation.cu:102-105) for every triangle: 4 lanes per triangle
// (lane k = stencil offset k of the centroid normal, evaluated along x, y, z), 16 triangles per wave
template <int SCENE>
__device__ __forceinline__ void flip_body4(uint32_t bid, uint32_t nblocks, const Vertex *__restrict__ uverts,
                                           const uint32_t *__restrict__ tri_e, uint32_t T, const SceneParams &sp,
                                           uint8_t *__restrict__ flip) {
    const uint32_t lane = lane_id(), g4 = lane >> 2, sub = lane & 3u;
    const uint32_t nwaves = nblocks * (blockDim.x >> 6);
    const uint32_t wave = bid * (blockDim.x >> 6) + (threadIdx.x >> 6);
    for (uint32_t t0 = wave * 16u; t0 < T; t0 += nwaves * 16u) {
        const uint32_t t = t0 + g4;
        const bool act = t < T;
        f3 v0 = mk(0, 0, 0), v1 = v0, v2 = v0, c = v0;
        if (act) {
            const Point p0 = uverts[tri_e[3 * t]].position, p1 = uverts[tri_e[3 * t + 1]].position, p2 = uverts[tri_e[3 * t + 2]].position;
            v0 = mk(p0.x, p0.y, p0.z); v1 = mk(p1.x, p1.y, p1.z); v2 = mk(p2.x, p2.y, p2.z);
            const f3 s = add3(add3(v0, v1), v2);
            c = mk(third_rn(s.x), third_rn(s.y), third_rn(s.z));
        }
        float unused;
        const f3 d = stencil_quad<SCENE, false>(c, sub, sp, unused);     // whole quads are active or idle together
        if (act && sub == 0) {
            const f3 an = normalize3(d);
            const f3 tn = normalize3(cross3(sub3(v1, v0), sub3(v2, v0)));
            flip[t] = dot3(tn, an) <= 0.0f ? 1 : 0;
        }
    }
}

// One lane per triangle: all 12 stencil samples of the centroid normal in one lane.  The 4 samples of an axis differ
// from the centroid in that coordinate only, so the compiler shares the other two coordinates' terms across them
// (sd_obj: ~70 instead of 93 instructions per sample), there are no cross-lane operations and the triangle set-up is
// not replicated over a lane group.
template <int SCENE>
__device__ __forceinline__ void flip_body1(uint32_t bid, uint32_t nblocks, const Vertex *__restrict__ uverts,
                                           const uint32_t *__restrict__ tri_e, uint32_t T, const SceneParams &sp,
                                           uint8_t *__restrict__ flip) {
    for (uint32_t t = bid * blockDim.x + threadIdx.x; t < T; t += nblocks * blockDim.x) {
        const Point p0 = uverts[tri_e[3 * t]].position, p1 = uverts[tri_e[3 * t + 1]].position, p2 = uverts[tri_e[3 * t + 2]].position;
        flip[t] = change_orientation<SCENE>(mk(p0.x, p0.y, p0.z), mk(p1.x, p1.y, p1.z), mk(p2.x, p2.y, p2.z), sp) ? 1 : 0;
    }
}

template <int SCENE>
__global__ __launch_bounds__(256) void k_flip(const Vertex *__restrict__ uverts, const uint32_t *__restrict__ tri_e,
                                               const uint32_t *__restrict__ ntri_ptr, const SceneParams *__restrict__ spp, uint8_t *__restrict__ flip) {
    flip_body4<SCENE>(blockIdx.x, gridDim.x, uverts, tri_e, *ntri_ptr, *spp, flip);
}

template <int SCENE>
__global__ __launch_bounds__(256) void k_flip1(const Vertex *__restrict__ uverts, const uint32_t *__restrict__ tri_e,
                                                const uint32_t *__restrict__ ntri_ptr, const SceneParams *__restrict__ spp, uint8_t *__restrict__ flip) {
    flip_body1<SCENE>(blockIdx.x, gridDim.x, uverts, tri_e, *ntri_ptr, *spp, flip);
}

// One lane per pre-weld vertex: the whole Newton round (centre sample + 12 stencil samples, terms shared as in
// k_flip1) in one lane, 64 vertices in flight per wave, finished lanes refill from the wave's round-robin share.
template <int SCENE>
__global__ __launch_bounds__(256) void k_project1(Vertex *__restrict__ uverts, const uint4 *__restrict__ tdesc,
                                                   const uint32_t *__restrict__ tpre, const uint32_t *__restrict__ coarse, const uint32_t *__restrict__ ntiles_ptr,
                                                   const SceneParams *__restrict__ spp, uint32_t *__restrict__ status,
                                                   uint32_t *__restrict__ prof, uint64_t *__restrict__ diag, ClaimArgs ca,
                                                   uint32_t *__restrict__ nu_out) {
    __shared__ uint32_t s_pre[LOCAL_TILES + 1], s_scan[17];
    const SceneParams &sp = *spp;
    const uint32_t ntiles = *ntiles_ptr;
    uint32_t U;
    if (tpre == nullptr) {                                            // (see k_project)
        if (ntiles > LOCAL_TILES) { if (threadIdx.x == 0) atomicOr(&status[ST_INTERNAL], 32u); return; }
        U = local_prefix(&tdesc[0].y, 4u, ntiles, s_pre, s_scan);
        if (blockIdx.x == 0 && threadIdx.x == 0) *nu_out = U;
    } else {
        U = ntiles ? tpre[ntiles] : 0u;
    }
    const uint32_t lane = lane_id();
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    // Runs of B = 2^bsh consecutive items are dealt round robin to the waves: the wave's j-th item is
    // ((j / B) * nwaves + wave) * B + j % B.  B = 64 (one contiguous run per wave) is the default.  Iteration counts are
    // spatially correlated, so with B = 64 some waves are slow in ALL lanes; short runs (B = 8: every wave sees the same
    // mix) were measured and lose: every wave then carries a few slow vertices through all the straggler rounds
    // (+16 % instructions, 48.7 -> 57 us at 512^3); fewer, persistent waves with lane refill lose more (2 per SIMD: 75 us).
    const uint32_t bsh = ca.bsh, bmask = (1u << bsh) - 1u;
    const uint32_t nbatch = (U + bmask) >> bsh;
    if (wave >= nbatch) return;
    const uint64_t t_start = diag ? __builtin_amdgcn_s_memtime() : 0ull;
    if (diag && lane == 0) diag[4ull * wave + 3] = __builtin_amdgcn_s_memrealtime() & 0xFFFFFFFFull;
    uint32_t rounds = 0, wround = 0;                                  // vertex-rounds evaluated (bookkeeping); loop iterations of this wave
    const uint32_t myb = (nbatch - wave - 1u) / nwaves + 1u;
    uint32_t qhead = 0;
    const uint32_t qend = myb << bsh;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    bool have = false, finalize = false;
    uint32_t slot = 0;
    int it = 0;
    f3 g = mk(0.0f, 0.0f, 0.0f);
    // Asynchronous weld claims (wmode bit 2; waves that own many batches): a lane that has written its vertex issues the
    // compare-and-swap for it and goes on with its next vertex; the returned value is looked at one Newton round later
    // (the top of the next loop iteration), by when it has long arrived -- the scattered-atomic traffic (the chip does
    // ~20 G/s: a millisecond for 20 M vertices at 4096^3) overlaps the projection's arithmetic instead of following it.
    const bool async_claims = (ca.wmode & 4u) != 0u;
    const uint32_t cshift = ca.wmode >> 8;
    bool pend = false;
    uint64_t pkey = 0;
    unsigned long long pres = 0;
    uint32_t pslot = 0, pe = 0;
    for (;;) {
        if (pend) {                                                   // the claim issued a round ago (or its next probe)
            if (pres == KEY_EMPTY || pres == pkey) {
                ca.vslot[pe] = pslot | (pres == KEY_EMPTY ? VS_WINNER : 0u);
                if (pres == KEY_EMPTY) { ca.wrep[pslot] = pe; ca.wfm[pe] = make_uint2(U32_NONE, U32_NONE); }
                pend = false;
            } else {
                pslot = hash_next(pslot, ca.wsize);
                pres = atomicCAS((unsigned long long *) &ca.wkeys[pslot], (unsigned long long) KEY_EMPTY, (unsigned long long) pkey);
            }
        }
        const uint64_t need = __ballot(!have);
        const uint32_t avail = qend - qhead;
        if (need != 0ull && avail != 0u) {
            const uint32_t rank = __popcll(need & lt_mask);
            const uint32_t j = qhead + rank;
            const uint32_t item = (((j >> bsh) * nwaves + wave) << bsh) + (j & bmask);
            if (!have && rank < avail && item < U) {
                slot = item_slot(item, tdesc, tpre, coarse, s_pre, ntiles);
                const Point p = uverts[slot].position;
                g = mk(p.x, p.y, p.z);
                have = true; finalize = false; it = 0;
            }
            const uint32_t want = __popcll(need);
            qhead += want < avail ? want : avail;
        }
        const uint64_t act = __ballot(have);
        if (act == 0ull) break;
#if defined(SDM_FAST_MATH)
        constexpr bool straggler_mode = false;   // fast build: one instruction sequence per vertex, so the copies of a vertex that two
                                                 // tiles compute stay bit-identical (contracted FMAs differ between the two layouts)
#else
        constexpr bool straggler_mode = true;
#endif
        if (straggler_mode && !async_claims && qend == qhead && __popcll(act) <= 16) {
            // Straggler mode.  Iteration counts are spatially correlated (slow convergence near the smooth-min blends), so a
            // few waves are left with a handful of live lanes for many more rounds while a full one-lane round costs ~960
            // instructions whatever the number of live lanes -- measured: the longest wave lives 2x the mean and sets the
            // kernel time.  With <= 16 vertices left and nothing to refill from, each vertex moves to a 4-lane group (lane k
            // = stencil offset k along x, y, z: k_project's layout, ~300 instructions per round).  Same sample positions,
            // same combination order, same values.
            const uint32_t sub = lane & 3u, grp = lane >> 2;
            uint64_t mm = act;
            for (uint32_t i = 0; i < 15u; i++) if (i < grp) mm &= mm - 1ull;          // drop the grp lowest live lanes
            const bool ghave = mm != 0ull;
            const int src = ghave ? (int) __builtin_ctzll(mm) : (int) lane;
            g.x = __shfl(g.x, src, 64); g.y = __shfl(g.y, src, 64); g.z = __shfl(g.z, src, 64);
            slot = (uint32_t) __shfl((int) slot, src, 64);
            it = __shfl(it, src, 64);
            finalize = __shfl((int) finalize, src, 64) != 0;
            have = ghave;
            for (;;) {
                const uint64_t qact = __ballot(have && sub == 0);
                if (qact == 0ull) break;
                rounds += __popcll(qact);
                if (have) {
                    float sd;
                    const f3 nn = normalize3(stencil_quad<SCENE, true>(g, sub, sp, sd));
                    if (!finalize) {
                        g = sub3(g, scale3(nn, sd));
                        it++;
                        if (fabsf(sd) <= 0.00001f) finalize = true;
                        else if (it >= SDM_MAX_NEWTON) { finalize = true; if (sub == 0) atomicOr(&status[ST_NEWTON_CAPPED], 1u); }
                    } else {
                        if (sub == 0) {
                            Vertex o;
                            o.position.x = g.x; o.position.y = g.y; o.position.z = g.z;
                            o.normal.x = nn.x; o.normal.y = nn.y; o.normal.z = nn.z;
                            uverts[slot] = o;
                            if (!((g.x - g.x) == 0.0f)) atomicOr(&status[ST_NONFINITE], 1u);
                        }
                        have = false;
                    }
                }
            }
            break;
        }
        rounds += __popcll(act);
        // Option "project_prio": issue priority by progress.  The SIMD's arbiter serves its waves oldest first: the five waves
        // of a SIMD finish one after the other (13.7 / 18.3 / 21.9 / 27.4 / 33.8 us at 512^3).  With this switch a wave that
        // has done fewer rounds outranks the ones ahead of it and they finish in order of their work instead (14.8 ... 33.6 us)
        // -- the last one at the same time: the kernel is bound by the SIMD's VALU issue, not by which wave is served.
        if (ca.wmode & 8u) {
            if (wround == 0u) __builtin_amdgcn_s_setprio(3);
            else if (wround == 1u) __builtin_amdgcn_s_setprio(2);
            else if (wround == 2u) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        wround++;
        if (have) {
            const float sd = sdf<SCENE>(g.x, g.y, g.z, sp);
            const f3 nn = empirical_normal<SCENE>(g, sp);
            if (!finalize) {
                g = sub3(g, scale3(nn, sd));
                it++;
                if (fabsf(sd) <= 0.00001f) finalize = true;
                else if (it >= SDM_MAX_NEWTON) { finalize = true; atomicOr(&status[ST_NEWTON_CAPPED], 1u); }
            } else {
                Vertex o;
                o.position.x = g.x; o.position.y = g.y; o.position.z = g.z;
                o.normal.x = nn.x; o.normal.y = nn.y; o.normal.z = nn.z;
                uverts[slot] = o;
                if (!((g.x - g.x) == 0.0f)) atomicOr(&status[ST_NONFINITE], 1u);       // src/cuda/mod.rs:289 needs the direct path
                have = false;
                if (async_claims) {
                    while (pend) {                                    // (rare: the previous claim is still probing -- finish it now)
                        if (pres == KEY_EMPTY || pres == pkey) {
                            ca.vslot[pe] = pslot | (pres == KEY_EMPTY ? VS_WINNER : 0u);
                            if (pres == KEY_EMPTY) { ca.wrep[pslot] = pe; ca.wfm[pe] = make_uint2(U32_NONE, U32_NONE); }
                            pend = false;
                        } else {
                            pslot = hash_next(pslot, ca.wsize);
                            pres = atomicCAS((unsigned long long *) &ca.wkeys[pslot], (unsigned long long) KEY_EMPTY, (unsigned long long) pkey);
                        }
                    }
                    uint64_t key;
                    if (packed_weld_key(o.position, key)) {
                        const uint64_t low = (1ull << cshift) - 1ull;
                        pkey = key; pe = slot;
                        pslot = cshift ? weld_home(key, key & ~(low | (low << 21) | (low << 42)), ca.wsize) : hash_home(key, ca.wsize);
                        pres = atomicCAS((unsigned long long *) &ca.wkeys[pslot], (unsigned long long) KEY_EMPTY, (unsigned long long) key);
                        pend = true;
                    } else {
                        atomicOr(&status[ST_WELD_RANGE], 1u);
                        ca.vslot[slot] = U32_NONE;
                    }
                }
            }
        }
    }
    while (__ballot(pend) != 0ull) {                                  // drain: the claims of the wave's last vertices
        if (pend) {
            if (pres == KEY_EMPTY || pres == pkey) {
                ca.vslot[pe] = pslot | (pres == KEY_EMPTY ? VS_WINNER : 0u);
                if (pres == KEY_EMPTY) { ca.wrep[pslot] = pe; ca.wfm[pe] = make_uint2(U32_NONE, U32_NONE); }
                pend = false;
            } else {
                pslot = hash_next(pslot, ca.wsize);
                pres = atomicCAS((unsigned long long *) &ca.wkeys[pslot], (unsigned long long) KEY_EMPTY, (unsigned long long) pkey);
            }
        }
    }
    claim_epilogue(bsh, wave, nwaves, myb, U, uverts, tdesc, tpre, coarse, ca, status, s_pre, ntiles);
    project_epilogue(wave, rounds, t_start, prof, diag);
}

// tile descriptors for a plain triangle list (generic weld): tiles of 256 triangles, every corner
// its own vertex (tri_e = identity), no flips
__global__ __launch_bounds__(256) void k_raw_desc(const uint32_t *__restrict__ ntri_ptr, uint4 *__restrict__ tdesc,
                                                   uint8_t *__restrict__ flip, uint32_t *__restrict__ ntiles_out,
                                                   uint32_t *__restrict__ nu_out, uint32_t *__restrict__ tri_e /* optional: identity */) {
    const uint32_t T = *ntri_ptr, ntiles = (T + 255u) / 256u;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < T; i += gridDim.x * blockDim.x) flip[i] = 0;
    if (tri_e)
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < 3u * T; i += gridDim.x * blockDim.x) tri_e[i] = i;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < ntiles; t += gridDim.x * blockDim.x) {
        const uint32_t tt = T - 256u * t < 256u ? T - 256u * t : 256u;
        tdesc[t] = make_uint4(768u * t, 3u * tt, 256u * t, tt);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { *ntiles_out = ntiles; *nu_out = 3u * T; }
}

// claim pass: one CAS per pre-weld vertex, per tile.  Latency-bound on scattered L2 atomics and independent of
// the orientation test, so the host runs it on a side stream concurrently with the VALU-bound k_flip.
__device__ __forceinline__ void claim_body(uint32_t bid, uint32_t nblocks, const Vertex *__restrict__ uverts,
                                           const uint4 *__restrict__ tdesc, uint32_t ntiles, uint64_t *__restrict__ wkeys,
                                           uint32_t wsize, uint32_t *__restrict__ vslot, uint32_t *__restrict__ status, uint32_t wmode,
                                           uint32_t *__restrict__ wrep, uint2 *__restrict__ wfm) {
    const uint32_t wide = wmode & 1u, shift = wmode >> 8;      // wmode = wide flag | coarse-cell shift << 8
    for (uint32_t tile = bid; tile < ntiles; tile += nblocks) {
        const uint4 d = tdesc[tile];
        for (uint32_t i = threadIdx.x; i < d.y; i += 256) {
            const uint32_t e = d.x + i;
            const uint32_t vs = wide ? weld_claim_wide(uverts, e, wkeys, wsize, status, shift) : weld_claim(uverts[e].position, wkeys, wsize, status, shift);
            vslot[e] = vs;
            if (vs != U32_NONE && (vs & VS_WINNER)) {
                wrep[vs & VS_SLOT] = e;                          // the key's representative vertex: this one
                wfm[e] = make_uint2(U32_NONE, U32_NONE);         // its record {winner's ordinal, minimum loser ordinal}
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_tile_claim(const Vertex *__restrict__ uverts, const uint4 *__restrict__ tdesc,
                                                     const uint32_t *__restrict__ ntiles_ptr, uint64_t *__restrict__ wkeys,
                                                     uint32_t wsize, uint32_t *__restrict__ vslot, uint32_t *__restrict__ status, uint32_t wmode,
                                                     uint32_t *__restrict__ wrep, uint2 *__restrict__ wfm) {
    claim_body(blockIdx.x, gridDim.x, uverts, tdesc, *ntiles_ptr, wkeys, wsize, vslot, status, wmode, wrep, wfm);
}

// Orientation test and weld claim in ONE dispatch: the LAST `claim_blocks` workgroups claim weld-table slots for the
// projected vertices (scattered CAS, latency-bound), the first ones run the orientation test (VALU-bound; few, long
// workgroups that take one wave slot per SIMD, dispatched first so the many short claim workgroups fill the rest).  Both only
// read the projection's output and write disjoint arrays.  Running them as two kernels on two streams costs a
// cross-queue fork and join (measured ~20 us of dependency latency at 512^3, more than the overlap wins); one
// dispatch overlaps them with no synchronisation at all, and the claim's workgroups are resident first.
template <int SCENE, int LANES>
__global__ __launch_bounds__(256) void k_flip_claim(const Vertex *__restrict__ uverts, const uint32_t *__restrict__ tri_e,
                                                     const uint32_t *__restrict__ ntri_ptr, const SceneParams *__restrict__ spp,
                                                     uint8_t *__restrict__ flip, uint32_t claim_blocks, const uint4 *__restrict__ tdesc,
                                                     const uint32_t *__restrict__ ntiles_ptr, uint64_t *__restrict__ wkeys, uint32_t wsize,
                                                     uint32_t *__restrict__ vslot, uint32_t *__restrict__ status, uint32_t wmode,
                                                     uint32_t *__restrict__ wrep, uint2 *__restrict__ wfm) {
    const uint32_t flip_blocks = gridDim.x - claim_blocks;
    if (blockIdx.x >= flip_blocks) {
        claim_body(blockIdx.x - flip_blocks, claim_blocks, uverts, tdesc, *ntiles_ptr, wkeys, wsize, vslot, status, wmode, wrep, wfm);
    } else if (LANES == 1) {
        flip_body1<SCENE>(blockIdx.x, flip_blocks, uverts, tri_e, *ntri_ptr, *spp, flip);
    } else {
        flip_body4<SCENE>(blockIdx.x, flip_blocks, uverts, tri_e, *ntri_ptr, *spp, flip);
    }
}


static constexpr uint32_t T_VMAX = 3072;

// per tile: apply the orientation flips to tri_e, derive every tile vertex's first-seen ordinal with LDS
// atomics, publish it (winner: plain store, loser: atomicMin)
// FLIP_SCENE >= 0: the orientation test (compute_mesh_generation.cu:102-105, one lane per triangle as in k_flip1) runs here
// instead of in its own launch -- possible once the weld claims have moved into the projection kernel, so that nothing
// else needs the separate dispatch; a tile's ~512 triangles are two passes of its 256 threads.
// Workgroup size: any multiple of 64 up to 1 024 (launch_weld).  512 threads per tile are the default when the orientation test
// runs here: a tile then takes half as long, so the tiles that do not divide evenly over the CUs (1 037 tiles on 256 CUs at
// 512^3: thirteen CUs get a fifth one) cost less at the end of the kernel (22.0 -> 21.2 us).
template <int FLIP_SCENE>
__global__ __launch_bounds__(1024) void k_tile_first(const uint4 *__restrict__ tdesc, const uint32_t *__restrict__ ntiles_ptr,
                                                     uint32_t *__restrict__ tri_e, const uint8_t *__restrict__ flip,
                                                     const Vertex *__restrict__ uverts, const SceneParams *__restrict__ spp,
                                                     uint32_t *__restrict__ vslot /* in: claim result; out: representative vertex | winner bit */,
                                                     uint32_t *__restrict__ first_ord, const uint32_t *__restrict__ wrep,
                                                     uint2 *__restrict__ wfm /* per REPRESENTATIVE vertex: {its own ordinal, minimum ordinal of the key's other vertices} */) {
    __shared__ uint32_t s_first[T_VMAX];
    const uint32_t ntiles = *ntiles_ptr, tid = threadIdx.x;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint4 d = tdesc[tile];
        const uint32_t vbase = d.x, nu = d.y, tbase = d.z, ttri = d.w;
        for (uint32_t i = tid; i < nu; i += blockDim.x) s_first[i] = U32_NONE;
        __syncthreads();
        for (uint32_t tl = tid; tl < ttri; tl += blockDim.x) {
            const uint32_t t = tbase + tl;
            uint32_t e0 = tri_e[3 * t], e1 = tri_e[3 * t + 1], e2 = tri_e[3 * t + 2];
            bool fl;
            if (FLIP_SCENE >= 0) {
                const Point p0 = uverts[e0].position, p1 = uverts[e1].position, p2 = uverts[e2].position;
                fl = change_orientation<(FLIP_SCENE >= 0 ? FLIP_SCENE : 0)>(mk(p0.x, p0.y, p0.z), mk(p1.x, p1.y, p1.z), mk(p2.x, p2.y, p2.z), *spp);
            } else {
                fl = flip[t] != 0;
            }
            if (fl) {
                const uint32_t x = e0; e0 = e2; e2 = x;
                tri_e[3 * t] = e0; tri_e[3 * t + 2] = e2;
            }
            atomicMin(&s_first[e0 - vbase], 3 * tl + 0);
            atomicMin(&s_first[e1 - vbase], 3 * tl + 1);
            atomicMin(&s_first[e2 - vbase], 3 * tl + 2);
        }
        __syncthreads();
        for (uint32_t i = tid; i < nu; i += blockDim.x) {
            const uint32_t e = vbase + i, lf = s_first[i];
            const uint32_t fo = lf == U32_NONE ? U32_NONE : 3u * tbase + lf;
            first_ord[e] = fo;
            // From here on a key is named by its representative vertex (the claim's winner), not by its hash slot: the
            // per-key records then live in the tile's own contiguous vertex range for every vertex that is not shared with
            // another tile, and the gathers of the next two kernels hit lines the tile has just touched.
            const uint32_t vs = vslot[e];
            if (vs != U32_NONE) {
                if (vs & VS_WINNER) {
                    vslot[e] = e | VS_WINNER;
                    wfm[e].x = fo;                                                   // (4-byte store: .y belongs to the losers)
                } else {
                    const uint32_t rep = wrep[vs & VS_SLOT];
                    vslot[e] = rep;
                    if (fo != U32_NONE) atomicMin(&wfm[rep].y, fo);
                }
            }
        }
        __syncthreads();
    }
}

// per tile: which vertices are first-seen, their rank by ordinal inside the tile, the tile's count.
// wv[rep] = tile * 4096 + rank for the key's first-seen vertex (rep = the key's representative vertex, vslot & VS_SLOT
// after k_tile_first); torder[3 * tbase + rank] = that vertex.
__global__ __launch_bounds__(256) void k_tile_rank(const uint4 *__restrict__ tdesc, const uint32_t *__restrict__ ntiles_ptr,
                                                    const uint32_t *__restrict__ first_ord, const uint2 *__restrict__ wfm,
                                                    const uint32_t *__restrict__ vslot,
                                                    uint32_t *__restrict__ wv, uint32_t *__restrict__ torder,
                                                    uint32_t *__restrict__ tfirst, const uint32_t *__restrict__ colstart,
                                                    const uint32_t *__restrict__ coltile, uint32_t *__restrict__ colvloc) {
    __shared__ uint32_t s_bits[3 * T_TRI / 32];      // 120 words: one bit per local ordinal
    __shared__ uint32_t s_pre[3 * T_TRI / 32];
    __shared__ uint32_t s_scan[17];
    const uint32_t ntiles = *ntiles_ptr, tid = threadIdx.x;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint4 d = tdesc[tile];
        if (tid < 3 * T_TRI / 32) s_bits[tid] = 0;
        __syncthreads();
        // up to T_VMAX / 256 = 12 vertices per thread; typically 2: keep their state in registers
        uint32_t lo_of[2], slot_of[2];
        uint32_t extra = 0;
#pragma unroll
        for (uint32_t k = 0; k < 2; k++) {
            const uint32_t i = tid + k * 256;
            lo_of[k] = U32_NONE; slot_of[k] = 0;
            if (i < d.y) {
                const uint32_t vs = vslot[d.x + i], fo = first_ord[d.x + i];
                if (vs != U32_NONE && fo != U32_NONE) {
                    const uint32_t s = vs & VS_SLOT; const uint2 fm = wfm[s]; const uint32_t a = fm.x, b = fm.y;   // one 8-byte gather
                    if ((a < b ? a : b) == fo) { lo_of[k] = fo - 3u * d.z; slot_of[k] = s; }
                }
            }
            if (lo_of[k] != U32_NONE) atomicOr(&s_bits[lo_of[k] >> 5], 1u << (lo_of[k] & 31u));
        }
        for (uint32_t i = tid + 512; i < d.y; i += 256) {            // rare: tiles with > 512 vertices
            const uint32_t vs = vslot[d.x + i], fo = first_ord[d.x + i];
            if (vs != U32_NONE && fo != U32_NONE) {
                const uint32_t s = vs & VS_SLOT; const uint2 fm = wfm[s]; const uint32_t a = fm.x, b = fm.y;   // one 8-byte gather
                if ((a < b ? a : b) == fo) { const uint32_t lo = fo - 3u * d.z; atomicOr(&s_bits[lo >> 5], 1u << (lo & 31u)); extra = 1; }
            }
        }
        __syncthreads();
        uint32_t total;
        const uint32_t w = tid < 3 * T_TRI / 32 ? __popc(s_bits[tid]) : 0u;
        const uint32_t ex = block_excl_scan(w, total, s_scan);
        if (tid < 3 * T_TRI / 32) s_pre[tid] = ex;
        __syncthreads();
        if (tid == 0) tfirst[tile] = total;
        if (colstart) {
            // (multi-GPU z-slabs) the columns whose first voxel lies in this tile: how many of the tile's first-seen vertices come before
            // the column's first triangle -- with the tiles' prefix (k_tile_finish) that is where the column's vertices start in the
            // welded order, because welded vertices are numbered in first-seen order (src/cuda/mod.rs:276-296)
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                const uint32_t c = tid + 256u * k, cs = colstart[c];
                if (cs && coltile[c] == tile) {
                    const uint32_t x = 3u * (cs - 1u - d.z);
                    colvloc[c] = x >= 3u * d.w ? total : s_pre[x >> 5] + __popc(s_bits[x >> 5] & ((1u << (x & 31u)) - 1u));
                }
            }
        }
#pragma unroll
        for (uint32_t k = 0; k < 2; k++) {
            if (lo_of[k] != U32_NONE) {
                const uint32_t lo = lo_of[k];
                const uint32_t r = s_pre[lo >> 5] + __popc(s_bits[lo >> 5] & ((1u << (lo & 31u)) - 1u));
                wv[slot_of[k]] = tile * 4096u + r;
                torder[3u * d.z + r] = d.x + tid + k * 256;
            }
        }
        if (d.y > 512) {
            for (uint32_t i = tid + 512; i < d.y; i += 256) {
                const uint32_t vs = vslot[d.x + i], fo = first_ord[d.x + i];
                if (vs != U32_NONE && fo != U32_NONE) {
                    const uint32_t s = vs & VS_SLOT; const uint2 fm = wfm[s]; const uint32_t a = fm.x, b = fm.y;   // one 8-byte gather
                    if ((a < b ? a : b) == fo) {
                        const uint32_t lo = fo - 3u * d.z;
                        const uint32_t r = s_pre[lo >> 5] + __popc(s_bits[lo >> 5] & ((1u << (lo & 31u)) - 1u));
                        wv[s] = tile * 4096u + r;
                        torder[3u * d.z + r] = d.x + i;
                    }
                }
            }
        }
        (void) extra;
        __syncthreads();
    }
}

// exclusive prefix of a u32 array of *n_ptr entries (single workgroup); out[n] = total = *total_out
__global__ __launch_bounds__(1024) void k_scan_u32(const uint32_t *__restrict__ in, const uint32_t *__restrict__ n_ptr,
                                                    uint32_t *__restrict__ out, uint32_t *__restrict__ total_out) {
    __shared__ uint32_t lds[17];
    const uint32_t n = *n_ptr;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n; base += blockDim.x) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n ? in[i] : 0u;
        uint32_t total;
        const uint32_t ex = block_excl_scan(v, total, lds);
        if (i < n) out[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) { out[n] = carry; *total_out = carry; }
}

// per tile: welded vertices (dense, first-seen order) and the welded index of every corner
// tfpre == nullptr (<= LOCAL_TILES tiles): every workgroup prefixes the tiles' first-seen counts for itself in LDS (no
// k_scan_u32 launch in front) and workgroup 0 publishes the welded total.
__global__ __launch_bounds__(256) void k_tile_finish(const Vertex *__restrict__ uverts, const uint4 *__restrict__ tdesc,
                                                      const uint32_t *__restrict__ ntiles_ptr, const uint32_t *__restrict__ tfpre,
                                                      const uint32_t *__restrict__ tfirst, const uint32_t *__restrict__ torder,
                                                      const uint32_t *__restrict__ tri_e, const uint32_t *__restrict__ vslot,
                                                      const uint32_t *__restrict__ wv, Vertex *__restrict__ vertices, uint32_t vcap,
                                                      uint32_t *__restrict__ indices, uint32_t *__restrict__ status,
                                                      uint32_t *__restrict__ nverts_out,
                                                      const uint32_t *__restrict__ pub_src, uint32_t *__restrict__ pub_dst, uint32_t pub_words,
                                                      uint32_t pub_ovf, uint32_t pub_nvert, const uint32_t *__restrict__ colstart,
                                                      const uint32_t *__restrict__ coltile, const uint32_t *__restrict__ colvloc,
                                                      uint32_t *__restrict__ colvstart /* [c] = 1 + first welded vertex first seen in column c, for the columns colstart names */) {
    __shared__ uint32_t s_pre[LOCAL_TILES + 1], s_scan[17];
    const uint32_t ntiles = *ntiles_ptr, tid = threadIdx.x;
    const bool local = tfpre == nullptr;
    if (blockIdx.x >= ntiles && blockIdx.x != 0) return;
    uint32_t total;
    if (local) {
        if (ntiles > LOCAL_TILES) {
            // (never reached: the host launches the prefix kernels above LOCAL_TILES.)  The host still has to SEE the flag: this
            // kernel is the only writer of its copy of the control block, so publish before leaving.
            if (tid == 0) atomicOr(&status[ST_INTERNAL], 64u);
            if (pub_dst && blockIdx.x == 0 && tid < pub_words) pub_dst[tid] = pub_src[tid] | (tid == pub_ovf + (ST_INTERNAL - ST_OVERFLOW) ? 64u : 0u);
            return;
        }
        total = local_prefix(tfirst, 1u, ntiles, s_pre, s_scan);
        if (blockIdx.x == 0 && tid == 0) *nverts_out = total;
    } else {
        total = tfpre[ntiles];
    }
    // The control block (counters + status) goes to the host from here: every word is final before this kernel starts
    // except the welded total (above) and this kernel's own overflow flag, which is known up front (the total against
    // the capacity).  A separate device-to-host copy behind the chain costs ~10 us of wall time per regen (barrier
    // packet + blit kernel).
    if (pub_dst && blockIdx.x == 0 && tid < pub_words) {
        uint32_t v = pub_src[tid];
        if (tid == pub_nvert) v = total;
        if (tid == pub_ovf && total > vcap) v |= 16u;
        pub_dst[tid] = v;
    }
    if (colstart && blockIdx.x == 0) {
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            const uint32_t c = tid + 256u * k, cs = colstart[c];
            uint32_t vs = 0;
            if (cs) { const uint32_t ct = coltile[c]; vs = (ct < ntiles ? (local ? s_pre[ct] : tfpre[ct]) : total) + colvloc[c] + 1u; }
            colvstart[c] = vs;
        }
    }
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint4 d = tdesc[tile];
        const uint32_t base = local ? s_pre[tile] : tfpre[tile], nf = tfirst[tile];
        for (uint32_t r = tid; r < nf; r += 256) {
            if (base + r < vcap) vertices[base + r] = uverts[torder[3u * d.z + r]];
            else atomicOr(&status[ST_OVERFLOW], 16u);
        }
        // a corner's welded index is a chain of four dependent loads (corner -> slot -> representative -> its tile's prefix): eight corners per
        // thread walk it side by side (a typical tile's 1536 corners in one pass), or the kernel is nothing but that latency, six times over
        // (12.4 -> 10.2 us at 512^3; starting the chains in front of the prefix, which only their last step needs, made it slower: 11.4-11.8)
        constexpr uint32_t U = 8;
        const uint32_t n3 = 3u * d.w, o0 = 3u * d.z;
        for (uint32_t i0 = tid; i0 < n3; i0 += 256u * U) {
            uint32_t e[U], vs[U], w[U], pre[U];
#pragma unroll
            for (uint32_t j = 0; j < U; j++) e[j] = i0 + 256u * j < n3 ? tri_e[o0 + i0 + 256u * j] : U32_NONE;
#pragma unroll
            for (uint32_t j = 0; j < U; j++) vs[j] = e[j] != U32_NONE ? vslot[e[j]] : U32_NONE;
#pragma unroll
            for (uint32_t j = 0; j < U; j++) w[j] = vs[j] != U32_NONE ? wv[vs[j] & VS_SLOT] : 0u;
#pragma unroll
            for (uint32_t j = 0; j < U; j++) pre[j] = vs[j] != U32_NONE ? (local ? s_pre[w[j] >> 12] : tfpre[w[j] >> 12]) : 0u;
#pragma unroll
            for (uint32_t j = 0; j < U; j++)
                if (i0 + 256u * j < n3) indices[o0 + i0 + 256u * j] = vs[j] != U32_NONE ? pre[j] + (w[j] & 4095u) : U32_NONE;
        }
    }
}

// ------------------------------------------------------------------------------------------
// layout converters
// ------------------------------------------------------------------------------------------
// compact triangle list (72 B each, emission order) from (uverts, tri_e)
// (flip == nullptr: tri_e already holds the final corner order, i.e. the weld has run)
__global__ __launch_bounds__(256) void k_expand_compact(const Vertex *__restrict__ uverts, const uint32_t *__restrict__ tri_e,
                                                         const uint8_t *__restrict__ flip, const uint32_t *__restrict__ ntri_ptr,
                                                         Triangle *__restrict__ out, uint32_t cap) {
    uint32_t n3 = 3u * *ntri_ptr;
    if (n3 > 3u * cap) n3 = 3u * cap;
    Vertex *o = reinterpret_cast<Vertex *>(out);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n3; i += gridDim.x * blockDim.x) {
        const uint32_t t = i / 3u, q = i - 3u * t;
        const uint32_t qq = (flip && flip[t]) ? 2u - q : q;
        o[i] = uverts[tri_e[3u * t + qq]];
    }
}

// reference raw layout: 5 slots per voxel, NaN/zero padded (compute_mesh_generation.cu:116-118)
__global__ __launch_bounds__(256) void k_expand_raw(const Vertex *__restrict__ uverts, const uint32_t *__restrict__ tri_e,
                                                     const uint8_t *__restrict__ flip, const uint8_t *__restrict__ cubeidx,
                                                     const uint32_t *__restrict__ trioff, uint32_t n, Triangle *__restrict__ out) {
    const uint32_t total = n * 5u;
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < total; s += gridDim.x * blockDim.x) {
        const uint32_t v = s / 5u, j = s - v * 5u;
        Triangle t;
        if (j < MC_TRI_COUNT[cubeidx[v]]) {
            const uint32_t tt = trioff[v] + j, o = 3u * tt;
            const bool f = flip && flip[tt];
            t.vertices[0] = uverts[tri_e[f ? o + 2 : o]]; t.vertices[1] = uverts[tri_e[o + 1]]; t.vertices[2] = uverts[tri_e[f ? o : o + 2]];
        } else {
            nan_pad(t);
        }
        out[s] = t;
    }
}

// ------------------------------------------------------------------------------------------
// A7 direct path: one thread per voxel, per-triangle-corner projection exactly as the reference
// kernel does it.  Used for non-lattice inputs, for the reference-ABI code object and as the
// "no edge sharing" A/B baseline.
// ------------------------------------------------------------------------------------------
template <int SCENE>
__global__ __launch_bounds__(128) void k_triangles_direct(const Point *__restrict__ vox, uint32_t n, Point vs, const SceneParams *__restrict__ spp,
                                                           Triangle *__restrict__ out) {
    const SceneParams &sp = *spp;  // scene uniform in device memory (a captured graph replays with new values)
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n) voxel_triangles_direct<SCENE>(vox[v], vs, sp, out + 5u * (size_t) v);
}

// ------------------------------------------------------------------------------------------
// A1 level-0 field on the device (src/cuda/mod.rs:105-122): x outer, y, z inner; optional
// z-slab [z0, z1) for the multi-GPU partition (same order, slab cells only)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_create_field(Point *__restrict__ vox, uint32_t x0, uint32_t x1, uint32_t z0, uint32_t z1,
                                                       uint32_t *__restrict__ n_out,
                                                       uint32_t *__restrict__ zero, uint32_t nzero, uint4 *__restrict__ ff,
                                                       uint32_t nff16) {
    // zero[0 .. nzero) (optional): the regen's control block is cleared here instead of by a memset node in front of
    // this kernel; n_out (inside that block) is the one word this kernel writes, and it is skipped by the clear.
    const float SIZE = MESH_GENERATION_BB_SIZE / (float) MESH_GENERATION_INIT_FACTOR;
    const uint32_t F = MESH_GENERATION_INIT_FACTOR, zs = z1 - z0, total = (x1 - x0) * F * zs;   // level-0 cells [x0,x1) x [0,F) x [z0,z1)
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nzero; i += gridDim.x * blockDim.x)
        if (zero + i != n_out) zero[i] = 0u;
    // ff[0 .. nff16) (optional): the weld table (keys + loser minima) starts as all-ones; written here, consumed ten
    // kernels later -- cheaper than a memset node, and far cheaper than a memset on a second stream (fork + join)
    const uint4 ones = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nff16; i += gridDim.x * blockDim.x) ff[i] = ones;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t z = z0 + i % zs, y = (i / zs) % F, x = x0 + i / (zs * F);
        Point p;
        p.x = (float) x * SIZE - MESH_GENERATION_BB_SIZE / 2.0f;
        p.y = (float) y * SIZE - MESH_GENERATION_BB_SIZE / 2.0f;
        p.z = (float) z * SIZE - MESH_GENERATION_BB_SIZE / 2.0f;
        vox[i] = p;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_out = total;
}

// ------------------------------------------------------------------------------------------
// raw NaN-padded slot list -> ordered compact list (the host loop of src/cuda/mod.rs:288-296
// skips a slot iff vertices[0].position.x is not finite)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool slot_valid(const Triangle *raw, uint32_t s) {
    const float x = raw[s].vertices[0].position.x;
    return (x - x) == 0.0f;   // finite
}

__global__ __launch_bounds__(256) void k_raw_count(const Triangle *__restrict__ raw, const uint32_t *__restrict__ n_ptr,
                                                    uint32_t *__restrict__ tilecnt) {
    __shared__ uint32_t s_cnt;
    const uint32_t n = *n_ptr;
    const uint32_t ntiles = (n + TILE - 1) / TILE;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        const uint32_t s = tile * TILE + threadIdx.x;
        const uint32_t ok = (s < n && slot_valid(raw, s)) ? 1u : 0u;
        const uint32_t ws = wave_sum(ok);
        if (lane_id() == 0 && ws) atomicAdd(&s_cnt, ws);
        __syncthreads();
        if (threadIdx.x == 0) tilecnt[tile] = s_cnt;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_raw_compact(const Triangle *__restrict__ raw, const uint32_t *__restrict__ n_ptr,
                                                      const uint32_t *__restrict__ tileoff, Triangle *__restrict__ out,
                                                      uint32_t *__restrict__ tri_e, uint32_t cap) {
    __shared__ uint32_t lds[17];
    const uint32_t n = *n_ptr;
    const uint32_t ntiles = (n + TILE - 1) / TILE;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t s = tile * TILE + threadIdx.x;
        const uint32_t ok = (s < n && slot_valid(raw, s)) ? 1u : 0u;
        uint32_t total;
        const uint32_t o = tileoff[tile] + block_excl_scan(ok, total, lds);
        if (ok && o < cap) {
            out[o] = raw[s];
            tri_e[3 * o] = 3 * o; tri_e[3 * o + 1] = 3 * o + 1; tri_e[3 * o + 2] = 3 * o + 2;
        }
    }
}

// ------------------------------------------------------------------------------------------
// multi-GPU order restoration: triangles per level-0 (x0, y0) column of this rank's list.
// The global order has z0 as the least significant level-0 digit, so z-slabs interleave per column;
// the per-column counts of every rank are all the merge needs.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_column_counts(const Point *__restrict__ vox, const uint32_t *__restrict__ n_ptr,
                                                        const uint8_t *__restrict__ cubeidx, uint32_t *__restrict__ colcnt) {
    __shared__ uint32_t s_hist[MESH_GENERATION_INIT_FACTOR * MESH_GENERATION_INIT_FACTOR];
    const uint32_t n = *n_ptr, F = MESH_GENERATION_INIT_FACTOR;
    const float cell = MESH_GENERATION_BB_SIZE / (float) MESH_GENERATION_INIT_FACTOR;
    for (uint32_t i = threadIdx.x; i < F * F; i += blockDim.x) s_hist[i] = 0;
    __syncthreads();
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) {
        const uint32_t nt = MC_TRI_COUNT[cubeidx[v]];
        if (nt) {
            const Point b = vox[v];
            uint32_t x0 = (uint32_t) floorf((b.x + 2.5f) / cell), y0 = (uint32_t) floorf((b.y + 2.5f) / cell);
            x0 = x0 < F ? x0 : F - 1; y0 = y0 < F ? y0 : F - 1;
            atomicAdd(&s_hist[x0 * F + y0], nt);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < F * F; i += blockDim.x)
        if (s_hist[i]) atomicAdd(&colcnt[i], s_hist[i]);
}


// ------------------------------------------------------------------------------------------
// multi-GPU order restoration, part 2: merge R per-slab triangle lists into the single-GPU order.
// counts[r][c] = triangles of rank r in level-0 column c (k_column_counts of every rank, all-gathered).  Every rank's
// list is column-sorted, and inside a column the global order is rank 0's triangles, then rank 1's, ...:
//     src_start[r][c] = sum_{c' < c} counts[r][c']                 (where the segment starts in rank r's list)
//     dst_start[r][c] = sum_{c' < c} sum_r' counts[r'][c'] + sum_{r' < r} counts[r'][c]
// k_merge_plan (one workgroup) writes both tables, k_merge_copy moves the segments: one workgroup per (rank, column)
// segment, 72-byte triangles as nine 8-byte words (a Triangle is only 8-byte aligned in general), fully coalesced on
// both sides.  HBM-bound: reads and writes 72 B x T once.
// ------------------------------------------------------------------------------------------
static constexpr uint32_t NCOL = MESH_GENERATION_INIT_FACTOR * MESH_GENERATION_INIT_FACTOR;     // 1024 columns = one per thread

__global__ __launch_bounds__(1024) void k_merge_plan(const uint32_t *__restrict__ counts, uint32_t cstride /* words per rank, >= NCOL */, uint32_t nranks,
                                                      uint32_t *__restrict__ plan, uint32_t cap, uint32_t *__restrict__ total_out, uint32_t *__restrict__ status) {
    __shared__ uint32_t lds[17];
    const uint32_t c = threadIdx.x;
    uint32_t col_total = 0;
    for (uint32_t r = 0; r < nranks; r++) col_total += counts[r * cstride + c];
    uint32_t total;
    const uint32_t col_base = block_excl_scan(col_total, total, lds);
    uint32_t within = 0;
    for (uint32_t r = 0; r < nranks; r++) {
        const uint32_t n = counts[r * cstride + c];
        uint32_t rank_total;
        const uint32_t src = block_excl_scan(n, rank_total, lds);
        plan[(r * NCOL + c) * 2u] = src;
        plan[(r * NCOL + c) * 2u + 1u] = col_base + within;
        within += n;
    }
    if (c == 0) {
        if (total > cap) { atomicOr(&status[ST_OVERFLOW], 32u); }
        *total_out = total;
    }
}

__global__ __launch_bounds__(256) void k_merge_copy(const Triangle *const *__restrict__ lists, const uint32_t *__restrict__ counts, uint32_t cstride,
                                                     const uint32_t *__restrict__ plan, uint32_t nranks, Triangle *__restrict__ out, uint32_t cap) {
    const uint32_t nseg = nranks * NCOL;
    for (uint32_t seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
        const uint32_t n = counts[(seg / NCOL) * cstride + seg % NCOL];
        if (n == 0) continue;
        const uint32_t src = plan[2u * seg], dst = plan[2u * seg + 1u];
        if (dst + n > cap) continue;                                  // flagged by k_merge_plan
        const uint2 *__restrict__ s = reinterpret_cast<const uint2 *>(lists[seg / NCOL] + src);
        uint2 *__restrict__ d = reinterpret_cast<uint2 *>(out + dst);
        const uint32_t words = 9u * n;
        for (uint32_t i = threadIdx.x; i < words; i += 256) d[i] = s[i];
    }
}


// ------------------------------------------------------------------------------------------
// Multi-GPU, DISTRIBUTED weld (x-slabs).  x is the outermost digit of the reference's order (src/cuda/mod.rs:105-122), so
// rank r's voxel and triangle lists are contiguous ranges of the single-GPU lists and the global first-seen vertex order
// is: rank 0's vertices in its local first-seen order, then rank 1's vertices that rank 0 does not already hold, ...
// Every rank runs the complete local pipeline (its own weld included); what remains is about the vertices within
// `band` of a slab face:
//   k_band_select        vertices near the upper face (keys go to rank r + 1) and near the lower face (candidates for a
//                        match with rank r - 1's upper band); wave-aggregated append
//   k_dist_hash_build /  the lower neighbour's band keys into a small hash set; my lower band looks itself up: equal key =
//   k_dist_hash_lookup   the same welded vertex of the single-GPU mesh = keep the lower rank's copy (dup_of = its position
//                        in that rank's band)
//   k_dist_keep_count /  ordered numbering of the vertices this rank keeps (three-kernel tile scan with k_scan_tiles)
//   k_dist_newidx
//   k_dist_pack_up       global numbers of my upper band (an upper-band vertex is never itself a duplicate: the two
//                        faces of a slab are at least one level-0 cell apart) -> rank r + 1
//   k_dist_finalize      kept vertices compacted, every index rewritten to its global number
// Nothing proportional to the mesh crosses a link or is hashed again; the mesh stays distributed.
// ------------------------------------------------------------------------------------------

// (counts come from device memory in every kernel of this group: the planned step enqueues them before the host knows any count;
//  `cap` arguments bound what a count that overflowed its plan can make a kernel touch -- such a step is repeated anyway)
__global__ __launch_bounds__(256) void k_band_select(const Vertex *__restrict__ v, const uint32_t *__restrict__ v_ptr, uint32_t vcap, float up_thr, float lo_thr, uint32_t has_up,
                                                      uint32_t has_lo, uint32_t *__restrict__ up_ids, uint64_t *__restrict__ up_keys,
                                                      uint32_t *__restrict__ lo_ids, uint32_t *__restrict__ ctr /* [0] n_up, [1] n_lo */,
                                                      uint32_t *__restrict__ dup_of, uint32_t *__restrict__ status) {
    const uint32_t lane = lane_id();
    const uint64_t lt = (1ull << lane) - 1ull;
    const uint32_t V = *v_ptr < vcap ? *v_ptr : vcap;
    const uint32_t Vr = (V + 63u) & ~63u;                      // whole waves stay together for the ballots
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < Vr; i += gridDim.x * blockDim.x) {
        bool up = false, lo = false;
        Point p = {0.0f, 0.0f, 0.0f};
        if (i < V) {
            dup_of[i] = U32_NONE;
            p = v[i].position;
            up = has_up && p.x >= up_thr;
            lo = has_lo && p.x <= lo_thr;
        }
        const uint64_t bu = __ballot(up), bl = __ballot(lo);
        if (bu) {
            const int leader = (int) __builtin_ctzll(bu);
            uint32_t base = 0;
            if ((int) lane == leader) base = atomicAdd(&ctr[0], (uint32_t) __popcll(bu));
            base = (uint32_t) __shfl((int) base, leader, 64);
            if (up) {
                const uint32_t pos = base + (uint32_t) __popcll(bu & lt);
                uint64_t key;
                if (!packed_weld_key(p, key)) atomicOr(&status[ST_WELD_RANGE], 1u);
                up_ids[pos] = i;
                up_keys[pos] = key;
            }
        }
        if (bl) {
            const int leader = (int) __builtin_ctzll(bl);
            uint32_t base = 0;
            if ((int) lane == leader) base = atomicAdd(&ctr[1], (uint32_t) __popcll(bl));
            base = (uint32_t) __shfl((int) base, leader, 64);
            if (lo) lo_ids[base + (uint32_t) __popcll(bl & lt)] = i;
        }
    }
}

__global__ __launch_bounds__(256) void k_dist_hash_build(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ n_ptr, uint32_t ncap,
                                                          uint64_t *__restrict__ hkeys, uint32_t *__restrict__ hpos, uint32_t hsize) {
    const uint32_t n = *n_ptr < ncap ? *n_ptr : ncap;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint64_t key = keys[i];
        uint32_t s = hash_home(key, hsize);
        for (uint32_t probe = 0; probe < hsize; probe++) {
            const uint64_t prev = atomicCAS((unsigned long long *) &hkeys[s], (unsigned long long) KEY_EMPTY, (unsigned long long) key);
            if (prev == KEY_EMPTY) { hpos[s] = i; break; }
            if (prev == key) break;                              // (a rank's welded vertices have distinct keys; harmless anyway)
            s = hash_next(s, hsize);
        }
    }
}

__global__ __launch_bounds__(256) void k_dist_hash_lookup(const Vertex *__restrict__ v, const uint32_t *__restrict__ lo_ids,
                                                           const uint32_t *__restrict__ n_lo_ptr, const uint64_t *__restrict__ hkeys,
                                                           const uint32_t *__restrict__ hpos, uint32_t hsize, uint32_t *__restrict__ dup_of,
                                                           uint32_t *__restrict__ status) {
    const uint32_t n = *n_lo_ptr;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const uint32_t i = lo_ids[j];
        uint64_t key;
        if (!packed_weld_key(v[i].position, key)) { atomicOr(&status[ST_WELD_RANGE], 1u); continue; }
        uint32_t s = hash_home(key, hsize);
        for (uint32_t probe = 0; probe < hsize; probe++) {
            const uint64_t k = hkeys[s];
            if (k == key) { dup_of[i] = hpos[s]; break; }
            if (k == KEY_EMPTY) break;
            s = hash_next(s, hsize);
        }
    }
}

__global__ __launch_bounds__(256) void k_dist_keep_count(const uint32_t *__restrict__ dup_of, const uint32_t *__restrict__ v_ptr, uint32_t vcap,
                                                          uint32_t *__restrict__ tilecnt) {
    __shared__ uint32_t s_cnt;
    const uint32_t V = *v_ptr < vcap ? *v_ptr : vcap;
    const uint32_t ntiles = (V + TILE - 1) / TILE;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        const uint32_t i = tile * TILE + threadIdx.x;
        const uint32_t ws = wave_sum((i < V && dup_of[i] == U32_NONE) ? 1u : 0u);
        if (lane_id() == 0 && ws) atomicAdd(&s_cnt, ws);
        __syncthreads();
        if (threadIdx.x == 0) tilecnt[tile] = s_cnt;
        __syncthreads();
    }
}

__device__ __forceinline__ void dist_record_body(uint32_t *__restrict__ dst, const uint32_t *__restrict__ count, const uint32_t *__restrict__ status, uint32_t host_status,
                                                 uint32_t cap_prev, uint32_t inject, uint32_t sig);
__device__ __forceinline__ void dist_publish_body(const uint32_t *__restrict__ nups, const uint32_t *__restrict__ counts, uint32_t nranks, uint32_t seq, uint32_t *__restrict__ pub);

// rec.dst != nullptr: workgroup 0 also writes this rank's record of the kept-count all-gather that follows (the count is the scan kernel's
// total in front of this launch, every status word is final): one launch less in the planned step.  (Prefixing the tile counts in every
// workgroup instead of the scan launch in front was measured and is not faster: 0.158-0.160 against 0.157-0.158 ms per 512^3 step.)
__global__ __launch_bounds__(256) void k_dist_newidx(const uint32_t *__restrict__ dup_of, const uint32_t *__restrict__ v_ptr, uint32_t vcap,
                                                      const uint32_t *__restrict__ tileoff, uint32_t *__restrict__ newidx, DistRecord rec) {
    __shared__ uint32_t lds[17];
    if (rec.dst && blockIdx.x == 0 && threadIdx.x == 0) dist_record_body(rec.dst, rec.count, rec.status, rec.host_status, rec.cap_prev, rec.inject, rec.sig);
    const uint32_t V = *v_ptr < vcap ? *v_ptr : vcap;
    const uint32_t ntiles = (V + TILE - 1) / TILE;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t i = tile * TILE + threadIdx.x;
        const uint32_t keep = (i < V && dup_of[i] == U32_NONE) ? 1u : 0u;
        uint32_t total;
        const uint32_t ex = block_excl_scan(keep, total, lds);
        if (i < V) newidx[i] = tileoff[tile] + ex;
    }
}

// (counts = the all-gathered per-rank records of the kept-count exchange: DIST_REC words each, word 0 = vertices kept)
__device__ __forceinline__ uint32_t dist_offset(const uint32_t *__restrict__ counts, uint32_t me) {
    uint32_t off = 0;
    for (uint32_t r = 0; r < me; r++) off += counts[r * DIST_REC];
    return off;
}

// One record of a status-carrying all-gather (the multi-GPU entry points agree on success before any payload moves):
// dst[0] = *count (a device-side count) or v0, dst[1] = v1 | the device status words that make the step fail, dst[2..3] = v2, v3.
__global__ void k_pack_record(uint32_t *__restrict__ dst, const uint32_t *__restrict__ count, const uint32_t *__restrict__ status,
                              uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3) {
    uint32_t st = v1;
    if (status) st |= (status[ST_WELD_RANGE] ? 0x100u : 0u) | (status[ST_INTERNAL] ? 0x200u : 0u) | (status[ST_OVERFLOW] ? 0x400u : 0u);
    dst[0] = count ? *count : v0;
    dst[1] = st;
    dst[2] = v2;
    dst[3] = v3;
}

// One DIST_REC-word record of the distributed weld's two all-gathers: {count, status, capacity for the lower neighbour's band, inject, sig, 0, 0, 0};
// status = the host's code for a local phase that already failed, else the device flags that make the chain's result unusable
__device__ __forceinline__ void dist_record_body(uint32_t *__restrict__ dst, const uint32_t *__restrict__ count, const uint32_t *__restrict__ status, uint32_t host_status,
                                                 uint32_t cap_prev, uint32_t inject, uint32_t sig) {
    uint32_t st = host_status;
    if (!st && status) {
        const uint32_t dev = (status[ST_OVERFLOW] ? 1u : 0u) | (status[ST_NOT_LATTICE] ? 2u : 0u) | (status[ST_INTERNAL] ? 8u : 0u) |
                             (status[ST_TILE_OVERFLOW] ? 16u : 0u) | (status[ST_NONFINITE] ? 32u : 0u);
        if (dev) st = 0x80000000u | dev;
        else if (status[ST_WELD_RANGE]) st = 0x100u;                   // (its own code: the caller reports SDMESH_ERR_WELD_RANGE)
    }
    dst[0] = (count && !host_status) ? *count : 0u;
    dst[1] = st; dst[2] = cap_prev; dst[3] = inject; dst[4] = sig; dst[5] = 0u; dst[6] = 0u; dst[7] = 0u;
}
__global__ void k_dist_record(uint32_t *__restrict__ dst, const uint32_t *__restrict__ count, const uint32_t *__restrict__ status, uint32_t host_status,
                              uint32_t cap_prev, uint32_t inject, uint32_t sig) {
    dist_record_body(dst, count, status, host_status, cap_prev, inject, sig);
}

// both record tables of a distributed step and its sequence number to pinned host memory: [seq | R records of the band sizes | R of the kept counts]
__device__ __forceinline__ void dist_publish_body(const uint32_t *__restrict__ nups, const uint32_t *__restrict__ counts, uint32_t nranks, uint32_t seq, uint32_t *__restrict__ pub) {
    const uint32_t t = threadIdx.x;                                  // (256 threads: 32 ranks x DIST_REC words)
    if (t == 0) pub[0] = seq;
    if (t < nranks * DIST_REC) { pub[8u + t] = nups[t]; pub[8u + nranks * DIST_REC + t] = counts[t]; }
}
__global__ void k_dist_publish(const uint32_t *__restrict__ nups, const uint32_t *__restrict__ counts, uint32_t nranks, uint32_t seq, uint32_t *__restrict__ pub) {
    dist_publish_body(nups, counts, nranks, seq, pub);
}

__global__ __launch_bounds__(256) void k_dist_pack_up(const uint32_t *__restrict__ up_ids, const uint32_t *__restrict__ n_up_ptr, uint32_t ncap,
                                                       const uint32_t *__restrict__ newidx, const uint32_t *__restrict__ counts, uint32_t me,
                                                       uint32_t *__restrict__ up_g) {
    const uint32_t off = dist_offset(counts, me);
    const uint32_t n_up = *n_up_ptr < ncap ? *n_up_ptr : ncap;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n_up; j += gridDim.x * blockDim.x) up_g[j] = off + newidx[up_ids[j]];
}

__global__ __launch_bounds__(256) void k_dist_finalize(const Vertex *__restrict__ v, const uint32_t *__restrict__ v_ptr, uint32_t vcap,
                                                        const uint32_t *__restrict__ t_ptr, uint32_t tcap, const uint32_t *__restrict__ dup_of,
                                                        const uint32_t *__restrict__ newidx, const uint32_t *__restrict__ prev_g, uint32_t pcap,
                                                        const uint32_t *__restrict__ counts, uint32_t me, uint32_t *__restrict__ indices,
                                                        Vertex *__restrict__ out_v, DistPublish pb) {
    // (pb.pub != nullptr: workgroup 0 publishes both record tables and the step's sequence number to pinned host memory -- every word of them
    //  was final before this kernel started)
    if (pb.pub && blockIdx.x == 0) dist_publish_body(pb.nups, pb.counts, pb.nranks, pb.seq, pb.pub);
    const uint32_t off = dist_offset(counts, me);
    const uint32_t V = *v_ptr < vcap ? *v_ptr : vcap, T = *t_ptr < tcap ? *t_ptr : tcap;
    // the kept vertices, 256 at a time: moved as dwords (coalesced both ways: almost every vertex is kept, so the destinations of a block are
    // one run with a few gaps) with the destinations staged in LDS -- a 24-byte vertex per lane touches every line six times
    __shared__ uint32_t s_dst[256];
    const uint32_t *__restrict__ v32 = reinterpret_cast<const uint32_t *>(v);
    uint32_t *__restrict__ o32 = reinterpret_cast<uint32_t *>(out_v);
    for (uint32_t b0 = blockIdx.x * 256u; b0 < V; b0 += gridDim.x * 256u) {
        const uint32_t i = b0 + threadIdx.x, nd = (V - b0 < 256u ? V - b0 : 256u) * 6u;
        uint32_t w[6];
#pragma unroll
        for (uint32_t j = 0; j < 6u; j++) w[j] = threadIdx.x + 256u * j < nd ? v32[6ull * b0 + threadIdx.x + 256u * j] : 0u;
        s_dst[threadIdx.x] = (i < V && dup_of[i] == U32_NONE) ? newidx[i] : U32_NONE;
        __syncthreads();
#pragma unroll
        for (uint32_t j = 0; j < 6u; j++) {
            const uint32_t k = threadIdx.x + 256u * j;
            if (k < nd) { const uint32_t d = s_dst[k / 6u]; if (d != U32_NONE) o32[6ull * d + k % 6u] = w[j]; }
        }
        __syncthreads();
    }
    // the indices in place, four per thread and pass with every load in front: an index costs two gathers (is its vertex a copy? its new
    // number) that do not depend on each other -- one after the other, one index at a time, the loop ran at a third of the copy rate
    const uint32_t n3 = 3u * T;
    for (uint32_t c0 = blockIdx.x * 1024u + threadIdx.x; c0 < n3; c0 += gridDim.x * 1024u) {
        uint32_t vi[4], d[4], n[4];
#pragma unroll
        for (uint32_t j = 0; j < 4u; j++) vi[j] = c0 + 256u * j < n3 ? indices[c0 + 256u * j] : U32_NONE;
#pragma unroll
        for (uint32_t j = 0; j < 4u; j++) {
            const bool ok = vi[j] < V;                                 // (always, of a chain that kept to its plan)
            d[j] = ok ? dup_of[vi[j]] : 0u;
            n[j] = ok ? newidx[vi[j]] : 0u;
        }
#pragma unroll
        for (uint32_t j = 0; j < 4u; j++)
            if (vi[j] < V) indices[c0 + 256u * j] = d[j] == U32_NONE ? off + n[j] : (d[j] < pcap ? prev_g[d[j]] : 0u);
    }
}

// ------------------------------------------------------------------------------------------
// Dense mode: the same voxel list as the hierarchical refine, derived from
// the dense (N+1)^3 sample lattice.
//   P2a k_sign_bits      one bit per sample (sd <= 0): a pure streaming read, 4 B x (N+1)^3 -> HBM-bound
//   P2b k_dense_classify one workgroup per level-0 cell: the (T+1)^3 sign bits of the cell (T = 2^k) are staged
//                        in LDS as (T+1)^2 rows of T+1 bits; every fine cell tests itself and all its ancestors
//                        (levels 1..k, corners at stride 2^(k-j): the emergent criterion A2' of the reference's
//                        refine-then-retain loop); threads run in the reference's child-slot (Morton, x-major)
//                        order, so a block scan gives the ordered compaction.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sign_bits(const float *__restrict__ samples, uint64_t total, uint32_t *__restrict__ bits) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const uint64_t nquads = (total + 3ull) >> 2;
    constexpr uint32_t UNR = 4;                                    // 16-byte loads in flight per lane: 1 / 4 / 8 -> 0.090 / 0.083 / 0.087 ms at 512^3; (one per lane leaves the read stream latency-bound)
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x * UNR;
    const uint32_t lane = lane_id();
    // grid-stride in runs of UNR whole waves so the 8 lanes that share an output word always run together
    for (uint64_t q0 = (((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) - lane) * UNR; q0 < nquads; q0 += stride) {
        v4f v[UNR];
        bool full[UNR];
#pragma unroll
        for (uint32_t j = 0; j < UNR; j++) {
            const uint64_t q = q0 + 64u * j + lane, i0 = q << 2;
            full[j] = q < nquads && i0 + 3 < total;
            if (full[j]) v[j] = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(samples + i0));
        }
#pragma unroll
        for (uint32_t j = 0; j < UNR; j++) {
            const uint64_t q = q0 + 64u * j + lane, i0 = q << 2;
            uint32_t nib = 0;
            if (full[j]) {
                nib = (v[j].x <= 0.0f ? 1u : 0u) | (v[j].y <= 0.0f ? 2u : 0u) | (v[j].z <= 0.0f ? 4u : 0u) | (v[j].w <= 0.0f ? 8u : 0u);
            } else if (q < nquads) {
                for (int k = 0; k < 4 && i0 + k < total; k++) nib |= (samples[i0 + k] <= 0.0f ? 1u : 0u) << k;
            }
            uint32_t w = nib << (4u * (lane & 7u));
            w |= __shfl_xor(w, 1, 64);
            w |= __shfl_xor(w, 2, 64);
            w |= __shfl_xor(w, 4, 64);
            if ((lane & 7u) == 0 && q < nquads) bits[q >> 3] = w;
        }
    }
}

__device__ __forceinline__ uint32_t lattice_bit(const uint32_t *__restrict__ bits, uint64_t i) {
    return (bits[i >> 5] >> (i & 31u)) & 1u;
}

template <bool EMIT>
__global__ __launch_bounds__(256) void k_dense_classify(const uint32_t *__restrict__ bits, uint32_t k /* levels */,
                                                         uint32_t *__restrict__ cellcnt /* per level-0 cell: count (in) / offset (EMIT) */,
                                                         Point *__restrict__ vox, uint8_t *__restrict__ cubeidx, uint32_t cap) {
    __shared__ uint64_t s_row[33 * 33];      // row (lx, ly): bit lz = sign at (X0*T + lx, Y0*T + ly, Z0*T + lz); T <= 32
    __shared__ uint32_t s_scan[17];
    __shared__ uint32_t s_mixed;
    const uint32_t F = MESH_GENERATION_INIT_FACTOR;
    const uint32_t T = 1u << k, N = F * T, S = N + 1u, R = T + 1u;
    const float h = (MESH_GENERATION_BB_SIZE / (float) MESH_GENERATION_INIT_FACTOR) / (float) T;
    const uint32_t tid = threadIdx.x;
    for (uint32_t cell = blockIdx.x; cell < F * F * F; cell += gridDim.x) {
        const uint32_t X0 = cell / (F * F), Y0 = (cell / F) % F, Z0 = cell % F;
        if (tid == 0) s_mixed = 0;
        __syncthreads();
        uint64_t any = 0, all = ~0ull;
        bool seen = false;
        const uint64_t rowmask = R >= 64 ? ~0ull : ((1ull << R) - 1ull);
        for (uint32_t r = tid; r < R * R; r += 256) {
            seen = true;
            const uint32_t lx = r / R, ly = r % R;
            const uint64_t i0 = ((uint64_t) (X0 * T + lx) * S + (Y0 * T + ly)) * S + (uint64_t) Z0 * T;
            // R <= 33 bits starting at bit i0: at most three 32-bit words
            const uint64_t w0 = i0 >> 5;
            const uint32_t sh = (uint32_t) (i0 & 31u);
            uint64_t lo = (uint64_t) bits[w0] | ((uint64_t) bits[w0 + 1] << 32);
            uint64_t v = lo >> sh;
            if (sh + R > 64u) v |= (uint64_t) bits[w0 + 2] << (64u - sh);
            v &= rowmask;
            s_row[r] = v;
            any |= v;
            all &= v | ~rowmask;
        }
        if (seen) {
            if (any != 0ull && all != ~0ull) atomicOr(&s_mixed, 1u);   // this thread saw both signs
            else if (any != 0ull) atomicOr(&s_mixed, 2u);              // saw only "inside"
            else atomicOr(&s_mixed, 4u);                               // saw only "outside"
        }
        __syncthreads();
        const uint32_t mm = s_mixed;
        const bool cell_mixed = (mm & 1u) || ((mm & 6u) == 6u);
        if (!cell_mixed && k > 0) {                             // uniform level-0 cell: nothing survives level 1 (level 0 itself is unfiltered)
            if (!EMIT && tid == 0) cellcnt[cell] = 0;
            __syncthreads();
            continue;
        }
        uint32_t running = EMIT ? cellcnt[cell] : 0u;
        const uint32_t ncode = 1u << (3u * k);
        for (uint32_t base = 0; base < ncode; base += 256) {
            const uint32_t code = base + tid;
            bool active = false;
            uint32_t x = 0, y = 0, z = 0, ci = 0;
            if (code < ncode) {
                for (uint32_t m = 0; m < k; m++) {               // triple m: z = bit 3m, y = 3m+1, x = 3m+2
                    z |= ((code >> (3u * m)) & 1u) << m;
                    y |= ((code >> (3u * m + 1u)) & 1u) << m;
                    x |= ((code >> (3u * m + 2u)) & 1u) << m;
                }
                active = true;
                for (uint32_t j = 1; j <= k && active; j++) {     // the cell's ancestor at level j (j = k: itself)
                    const uint32_t s = k - j, step = 1u << s;
                    const uint32_t ax = (x >> s) << s, ay = (y >> s) << s, az = (z >> s) << s;
                    uint32_t cnt = 0;
#pragma unroll
                    for (int c = 0; c < 8; c++) {
                        const uint32_t cx = ax + (cox(c) ? step : 0u), cy = ay + (coy(c) ? step : 0u), cz = az + (coz(c) ? step : 0u);
                        const uint32_t b = (uint32_t) (s_row[cx * R + cy] >> cz) & 1u;
                        cnt += b;
                        if (j == k) ci |= b << c;
                    }
                    active = cnt != 0 && cnt != 8;
                }
                if (k == 0) {                                     // level 0 is unfiltered (src/cuda/mod.rs:105-122)
                    active = true;
#pragma unroll
                    for (int c = 0; c < 8; c++)
                        ci |= ((uint32_t) (s_row[(uint32_t) cox(c) * R + (uint32_t) coy(c)] >> coz(c)) & 1u) << c;
                }
            }
            uint32_t total;
            const uint32_t ex = block_excl_scan(active ? 1u : 0u, total, s_scan);
            if (EMIT && active) {
                const uint32_t o = running + ex;
                if (o < cap) {
                    Point p;
                    p.x = (float) (X0 * T + x) * h - 2.5f;
                    p.y = (float) (Y0 * T + y) * h - 2.5f;
                    p.z = (float) (Z0 * T + z) * h - 2.5f;
                    vox[o] = p;
                    cubeidx[o] = (uint8_t) ci;
                }
            }
            running += total;
        }
        if (!EMIT && tid == 0) cellcnt[cell] = running;
        __syncthreads();
    }
}

__global__ void k_set_u32(uint32_t *p, uint32_t v) { *p = v; }

// ------------------------------------------------------------------------------------------
// P1: dense SDF sample pass, (N+1)^3 f32, x outer / z inner, one float4 store per thread
// ------------------------------------------------------------------------------------------
template <int SCENE, bool NT>
__global__ __launch_bounds__(256) void k_sample_dense(float *__restrict__ out, uint32_t N, uint64_t total, const SceneParams *__restrict__ spp,
                                                       uint32_t contig, uint64_t first /* flat index of out[0] in the volume (x-slabs) */) {
    const SceneParams &sp = *spp;  // scene uniform in device memory (a captured graph replays with new values)
    // Flat quads (4 consecutive samples, 16-B aligned float4 store) over the x-outer / z-inner volume.
    // (x, y, z) of a thread's first quad comes from one division; every further grid-stride step adds
    // the stride decomposed in base S, so the loop body has no division.
    typedef float v4f __attribute__((ext_vector_type(4)));
    const float h = 5.0f / (float) N;
    const uint32_t S = N + 1u;
    const uint64_t nquads = (total + 3ull) >> 2;
    // contig = 0: grid-stride (consecutive workgroups write consecutive 4 KiB pieces, each workgroup strides by the
    // whole grid); contig = 1: every workgroup owns one contiguous run of quads
    uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
    uint64_t qd = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t qlim = nquads;
    if (contig) {
        const uint64_t per = ((nquads + gridDim.x - 1) / gridDim.x + blockDim.x - 1) / blockDim.x * blockDim.x;
        qd = (uint64_t) blockIdx.x * per + threadIdx.x;
        qlim = (uint64_t) (blockIdx.x + 1) * per < nquads ? (uint64_t) (blockIdx.x + 1) * per : nquads;
        stride = blockDim.x;
    }
    if (qd >= qlim) return;
    uint64_t i0 = qd << 2;
    uint32_t z = (uint32_t) ((i0 + first) % S);
    uint64_t r = (i0 + first) / S;
    uint32_t y = (uint32_t) (r % S), x = (uint32_t) (r / S);
    const uint64_t st4 = stride << 2;
    const uint32_t sz = (uint32_t) (st4 % S), sy = (uint32_t) ((st4 / S) % S), sx = (uint32_t) (st4 / S / S);
    for (; qd < qlim; qd += stride) {
        float v[4];
        if (SCENE == SCENE_SPHERE) {
            // Sphere: the row context is one float (x^2 + y^2), so a quad that crosses a row end is handled with
            // selects instead of branches: samples k >= rem belong to the next row.  |p|^2 is 0 or a normal float
            // on the lattice, hence the lean sqrt.  (Same operations in the same order as sdf<SCENE_SPHERE>.)
            const uint32_t rem = S - z;                                          // samples left in this row (>= 1)
            const bool ywrap = (y + 1u == S);
            const float px = (float) x * h - 2.5f, py = (float) y * h - 2.5f;
            const float nx = (float) (ywrap ? x + 1u : x) * h - 2.5f, ny = (float) (ywrap ? 0u : y + 1u) * h - 2.5f;
            const float a_cur = px * px + py * py, a_nxt = nx * nx + ny * ny;
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                const bool nxt = k >= rem;
                const float pz = (float) (nxt ? k - rem : z + k) * h - 2.5f;
                v[k] = sqrt_rn_normal((nxt ? a_nxt : a_cur) + pz * pz) - 1.0f;
            }
        } else {
            // generic: the (x, y)-only part of the SDF is hoisted into a RowCtx, rebuilt when the quad crosses a row end
            uint32_t cx = x, cy = y, cz = z;
            RowCtx<SCENE> row;
            row.init((float) cx * h - 2.5f, (float) cy * h - 2.5f, sp);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                v[k] = row.eval((float) cz * h - 2.5f, sp);
                if (++cz == S) {
                    cz = 0;
                    if (++cy == S) { cy = 0; ++cx; }
                    row.init((float) cx * h - 2.5f, (float) cy * h - 2.5f, sp);
                }
            }
        }
        if (i0 + 3 < total) {
            v4f q = {v[0], v[1], v[2], v[3]};
            if (NT) __builtin_nontemporal_store(q, reinterpret_cast<v4f *>(out + i0));   // write-once stream
            else *reinterpret_cast<v4f *>(out + i0) = q;
        } else {
            for (int k = 0; k < 4 && i0 + k < total; k++) out[i0 + k] = v[k];
        }
        i0 += st4;
        z += sz; if (z >= S) { z -= S; y++; }
        y += sy; if (y >= S) { y -= S; x++; }
        x += sx;
    }
}

// Workgroup ids are dealt to the eight XCDs round robin; with `xcd` set, the workgroups of one XCD take one contiguous eighth of the
// pieces instead of every eighth piece, so that an XCD's L2 writes ONE ascending stream (measured: launch_sample_dense_rows).
__device__ __forceinline__ uint32_t xcd_contiguous(uint32_t wg, uint32_t nwg, uint32_t xcd) {
    const uint32_t per = nwg >> 3;
    return (xcd && wg < (per << 3)) ? (wg & 7u) * per + (wg >> 3) : wg;
}
// Write ceiling for P1: the same store stream (contiguous run of float4 per workgroup, nontemporal) with no
// arithmetic in front of it.  bench.py times it beside P1 to show how much of the achievable write rate P1 reaches.
__global__ __launch_bounds__(256) void k_stream_fill(float *__restrict__ out, uint64_t total, float value, uint32_t xcd) {
    // (grid = ceil(quads / 256): every thread issues one store -- the shape the write stream is fastest in, see k_sample_dense_q1)
    typedef float v4f __attribute__((ext_vector_type(4)));
    const uint64_t nquads = total >> 2;
    const uint64_t per = ((nquads + gridDim.x - 1) / gridDim.x + blockDim.x - 1) / blockDim.x * blockDim.x;
    const uint32_t wg = xcd_contiguous(blockIdx.x, gridDim.x, xcd);
    uint64_t qd = (uint64_t) wg * per + threadIdx.x;
    const uint64_t qlim = (uint64_t) (wg + 1) * per < nquads ? (uint64_t) (wg + 1) * per : nquads;
    const v4f q = {value, value, value, value};
    for (; qd < qlim; qd += blockDim.x) __builtin_nontemporal_store(q, reinterpret_cast<v4f *>(out) + qd);
    if (blockIdx.x == 0 && threadIdx.x < (total & 3ull)) out[(nquads << 2) + threadIdx.x] = value;
}

// P1, row-structured variant: one wave per z-row of S = N + 1 samples.  Lane l evaluates the quads l, l + 64, ...
// of its row and stores them as dword-aligned float4 (a row starts at byte 4 * S * row, S odd), the S % 4 tail samples
// go out as scalars.  No div/mod or row-wrap logic in the sample loop; the (x, y) terms are computed once per row.
// P1, one 16-byte quad per thread, one 4 KiB piece per workgroup, as many workgroups as pieces (no loop).
//
// The write stream itself decides the shape: a pure fill of the 540 MB volume sustains 7.1 TB/s when every thread issues
// exactly ONE float4 store and retires (131 842 workgroups at 512^3), 6.1 TB/s with two stores per thread, 5.3 TB/s in
// the persistent shapes of k_sample_dense (profiles: r02 bench, write_ceiling / tools sweep) -- a wave that still has
// stores to issue holds its slot, and the memory pipeline prefers many short waves.  With one quad per thread the index
// arithmetic cannot be amortised over a loop, so (x, y, z) comes from two exact 32-bit divisions by S = N + 1 done with a
// host-computed magic number (mulhi + shift + fixup: 6 instructions each), and the sphere's four samples are evaluated
// as two packed pairs (pz = (float) z * h - 2.5f etc. on v_pk_*; same operations in the same order per component).
// Volumes of 2^32 samples or more go out as several launches of whole planes (the kernel indexes relative to its launch's first plane).
struct DivMagic { uint32_t m, add, sh; };           // q = add ? (((n - hi) >> 1) + hi) >> sh : hi >> sh,  hi = mulhi(n, m)
__device__ __forceinline__ uint32_t div_magic(uint32_t n, DivMagic d) {
    const uint32_t hi = __umulhi(n, d.m);
    return d.add ? (((n - hi) >> 1) + hi) >> d.sh : hi >> d.sh;
}

template <int SCENE, bool NT>
__global__ __launch_bounds__(256) void k_sample_dense_q1(float *__restrict__ out, uint32_t N, uint32_t total, const SceneParams *__restrict__ spp,
                                                          DivMagic dS, float h /* = 5.0f / (float) N, computed on the host: the same IEEE quotient */,
                                                          uint32_t xplane0 /* out[0] is sample (xplane0, 0, 0): x-slabs, and volumes of >= 2^32 samples in batches of planes */,
                                                          uint32_t xcd) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const SceneParams &sp = *spp;
    const uint32_t S = N + 1u;
    const uint32_t wg = xcd_contiguous(blockIdx.x, gridDim.x, xcd);
    const uint32_t i0 = (wg * 256u + threadIdx.x) << 2;
    if (i0 >= total) return;
    // (x, y, z) of the WAVE's first sample on the scalar unit: two magic divisions of the workgroup's first index, then the wave's
    // offset and at most two row steps (a piece is 1 024 consecutive samples, a row has S >= 512 of them), all on uniform values.
    // A wave's 256 samples touch at most two rows, so a lane needs one compare and three selects.
    const uint32_t iw = wg << 10;                              // (indices relative to the slab: it starts on a plane, so x = xplane0 + the quotient)
    const uint32_t r0 = div_magic(iw, dS), x0 = xplane0 + div_magic(r0, dS);
    uint32_t x, y, z;
    if (S >= 512u) {                                         // (the usual case: no loop)
        const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x) & ~63u;
        uint32_t zw = (iw - r0 * S) + (wv << 2), yw = r0 - (x0 - xplane0) * S, xw = x0;
        if (zw >= S) { zw -= S; yw++; }
        if (zw >= S) { zw -= S; yw++; }
        if (yw >= S) { yw -= S; xw++; }
        uint32_t ya = yw + 1u, xa = xw;                      // the wave's second row
        if (ya == S) { ya = 0u; xa++; }
        z = zw + ((threadIdx.x & 63u) << 2);
        const bool c = z >= S;
        z = c ? z - S : z; y = c ? ya : yw; x = c ? xa : xw;
    } else {
        const uint32_t rr = div_magic(i0, dS), xl = div_magic(rr, dS);
        z = i0 - rr * S; y = rr - xl * S; x = xplane0 + xl;
    }
    float v[4];
    if (SCENE == SCENE_SPHERE) {
        // |p|^2 is 0 or a normal float on the lattice, hence the lean sqrt; its 2^-120 nudge rides on the row term a = x^2 + y^2
        // (a is 0 or >= 2^-46, so (a + 2^-120) + z^2 == (a + z^2) + 2^-120 bit for bit).  A quad may cross a row end (S is odd:
        // 37 % of the waves hold such a lane): its samples k with z + k >= S belong to the next row.  Only those lanes replace
        // the z coordinate and the row term of the samples concerned, behind a branch the other waves skip; the arithmetic
        // itself is one code path.  All the integers involved are < 2^24, so the float adds are exact.
        const float px = (float) x * h - 2.5f, py = (float) y * h - 2.5f;
#if defined(SDM_HAVE_PAIRS)
        const float a_cur = (px * px + py * py) + SDM_SQRT_NUDGE;
        const float zf = (float) z;
        v2f q0 = mk2(zf, zf + 1.0f), q1 = mk2(zf + 2.0f, zf + 3.0f);
        v2f a0 = mk2(a_cur, a_cur), a1 = a0;
        if (__builtin_expect(z + 3u >= S, 0)) {
            const bool ywrap = (y + 1u == S);
            const float nx = (float) (ywrap ? x + 1u : x) * h - 2.5f, ny = (float) (ywrap ? 0u : y + 1u) * h - 2.5f;
            const float a_nxt = (nx * nx + ny * ny) + SDM_SQRT_NUDGE;
            const uint32_t rem = S - z;                      // 1 .. 3 samples left in this row
            if (rem <= 1u) { q0.y = (float) (1u - rem); a0.y = a_nxt; }
            if (rem <= 2u) { q1.x = (float) (2u - rem); a1.x = a_nxt; }
            q1.y = (float) (3u - rem); a1.y = a_nxt;
        }
        const v2f p0 = q0 * h - 2.5f, p1 = q1 * h - 2.5f;
        const v2f r0v = sqrt_rn_normal2_nonudge(a0 + p0 * p0) - 1.0f;
        const v2f r1v = sqrt_rn_normal2_nonudge(a1 + p1 * p1) - 1.0f;
        v[0] = r0v.x; v[1] = r0v.y; v[2] = r1v.x; v[3] = r1v.y;
#else
        {
            const float a_cur = px * px + py * py;
            const uint32_t rem = S - z;
            const bool ywrap = (y + 1u == S);
            const float nx = (float) (ywrap ? x + 1u : x) * h - 2.5f, ny = (float) (ywrap ? 0u : y + 1u) * h - 2.5f;
            const float a_nxt = nx * nx + ny * ny;
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                const bool nxt = k >= rem;
                const float pz = (float) (nxt ? k - rem : z + k) * h - 2.5f;
                v[k] = sqrt_rn_normal((nxt ? a_nxt : a_cur) + pz * pz) - 1.0f;
            }
        }
#endif
    } else {
        uint32_t cx = x, cy = y, cz = z;
        RowCtx<SCENE> row;
        row.init((float) cx * h - 2.5f, (float) cy * h - 2.5f, sp);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            v[k] = row.eval((float) cz * h - 2.5f, sp);
            if (++cz == S) {
                cz = 0;
                if (++cy == S) { cy = 0; ++cx; }
                row.init((float) cx * h - 2.5f, (float) cy * h - 2.5f, sp);
            }
        }
    }
    if (i0 + 3u < total) {
        const v4f q = {v[0], v[1], v[2], v[3]};
        if (NT) __builtin_nontemporal_store(q, reinterpret_cast<v4f *>(out + i0));
        else *reinterpret_cast<v4f *>(out + i0) = q;
    } else {
        for (uint32_t k = 0; k < 4u && i0 + k < total; k++) out[i0 + k] = v[k];
    }
}

template <int SCENE, bool NT>
__global__ __launch_bounds__(256) void k_sample_dense_rows(float *__restrict__ out, uint32_t N, const SceneParams *__restrict__ spp,
                                                            uint32_t xb /* first x plane */, uint32_t xe /* one past the last: out[0] is sample (xb, 0, 0) */) {
    typedef float v4fu __attribute__((ext_vector_type(4), aligned(4)));
    const SceneParams &sp = *spp;
    const float h = 5.0f / (float) N;
    const uint32_t S = N + 1u, nrows = S * (xe - xb), nq = S >> 2, lane = lane_id();
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
    // A wave takes ROWS_PER_WAVE consecutive rows at a time.  The S % 4 samples behind a row's last whole quad (one, for the odd
    // S of a power-of-two grid) would cost the whole wave a full evaluation per row for one busy lane (a third on top of the two
    // quad passes of a 513-sample row); they are evaluated for all rows of the group together instead, one lane per sample
    // with its own row context.
    constexpr uint32_t ROWS_PER_WAVE = 16;
    const uint32_t tails = S & 3u, ngroups = (nrows + ROWS_PER_WAVE - 1u) / ROWS_PER_WAVE;
    for (uint32_t g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); g < ngroups; g += nwaves) {
        const uint32_t row0 = g * ROWS_PER_WAVE, rows = nrows - row0 < ROWS_PER_WAVE ? nrows - row0 : ROWS_PER_WAVE;
        uint32_t xl = row0 / S, y = row0 - xl * S;                      // (uniform)
        if (nq) {
            for (uint32_t r = 0; r < rows; r++) {
                RowCtx<SCENE> rc;
                rc.init((float) (xb + xl) * h - 2.5f, (float) y * h - 2.5f, sp);
                float *__restrict__ o = out + (uint64_t) (row0 + r) * S;
                for (uint32_t q = lane; q < nq; q += 64) {
                    const uint32_t z = q << 2;
                    v4fu v;
#if defined(SDM_HAVE_PAIRS)
                    const v2f lo = row_eval2<SCENE>(rc, mk2((float) z * h - 2.5f, (float) (z + 1u) * h - 2.5f), sp);
                    const v2f hi = row_eval2<SCENE>(rc, mk2((float) (z + 2u) * h - 2.5f, (float) (z + 3u) * h - 2.5f), sp);
                    v.x = lo.x; v.y = lo.y; v.z = hi.x; v.w = hi.y;
#else
                    v.x = rc.eval((float) z * h - 2.5f, sp);
                    v.y = rc.eval((float) (z + 1u) * h - 2.5f, sp);
                    v.z = rc.eval((float) (z + 2u) * h - 2.5f, sp);
                    v.w = rc.eval((float) (z + 3u) * h - 2.5f, sp);
#endif
                    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v4fu *>(o + z));
                    else *reinterpret_cast<v4fu *>(o + z) = v;
                }
                if (++y == S) { y = 0; ++xl; }
            }
        }
        if (lane < rows * tails) {                                      // (tails <= 3, so at most 48 lanes)
            const uint32_t r = lane / tails, k = lane - r * tails, row = row0 + r;
            const uint32_t tx = row / S, ty = row - tx * S, tz = (nq << 2) + k;
            RowCtx<SCENE> rc;
            rc.init((float) (xb + tx) * h - 2.5f, (float) ty * h - 2.5f, sp);
            out[(uint64_t) row * S + tz] = rc.eval((float) tz * h - 2.5f, sp);
        }
    }
}

// P1 for sd_obj / its animated form, row-structured with WHOLE-LINE stores and the z terms out of the row loop.
//
// k_sample_dense_rows pays, per 513-sample row, two 117-instruction quad passes of which a third rebuilds terms that depend on z
// alone (the lattice coordinate, the skeleton's three z residuals, z^2), a 50-instruction row context every lane computes
// redundantly, and dword-aligned 16-byte stores that straddle cache lines (1.10x the bytes).  Here a wave takes ROWS (8) rows whose
// starts have the SAME phase against the 16-byte quads of the output (rows lr = c + 4 i: a row starts at lr * S floats), so that
//   * one set of z terms per lane (two aligned quads = 8 samples x {sz, mz, az, z^2}) serves all the wave's rows;
//   * every quad store is 16-byte aligned (nontemporal, whole lines across the wave);
//   * the row contexts are computed once, one lane per row, and broadcast with v_readlane (4 scalars per row);
//   * the <= 3 samples in front of a row's first aligned quad and the <= 3 behind its last are evaluated for the whole group
//     together, one lane per sample.
// Same operations in the same order per sample as RowCtx<SCENE>::eval (the 2^-120 nudge of the lean square roots rides on the
// row terms: every term is 0 or >= 2^-60, so (a + 2^-120) + t == (a + t) + 2^-120 bit for bit) -> the same bits
// (tests: test_dense_sample_matches_oracle, ..._at_full_size_..., test_dense_slabs_concatenate_to_the_whole_volume).
template <int SCENE, bool NT, uint32_t ROWS, int QPL /* aligned quads per lane: a wave covers 64 QPL quads of each of its rows */>
__global__ __launch_bounds__(256) void k_sample_dense_rows_aligned(float *__restrict__ out /* 16-byte aligned */, uint32_t N,
                                                                    const SceneParams *__restrict__ spp, uint32_t xb, uint32_t xe, uint32_t flags) {
    static_assert(SCENE == SCENE_SDOBJ || SCENE == SCENE_ANIM, "sd_obj family only");
    static_assert(ROWS <= 64 && (ROWS & (ROWS - 1u)) == 0, "one lane per row context");
#if defined(SDM_HAVE_PAIRS)                      // (device pass only: the packed 2-vector helpers of sdf.hpp)
    typedef float v4f __attribute__((ext_vector_type(4)));
    const SceneParams &sp = *spp;
    const float h = 5.0f / (float) N;
    const uint32_t S = N + 1u, nrows = S * (xe - xb), lane = lane_id();
#if defined(SDM_FAST_MATH)
    const float nud = 0.0f;
#else
    const float nud = SDM_SQRT_NUDGE;
#endif
    // wave -> (chunk of 64 QPL quads along z, phase class, group of ROWS rows of that class)
    const uint32_t per_class = (((nrows + 3u) >> 2) + ROWS - 1u) / ROWS;            // row groups per phase class
    const uint32_t nchunks = ((S >> 2) + 64u * QPL - 1u) / (64u * QPL);               // (S >> 2 = the most aligned quads a row can have)
    // (the wave's number through v_readfirstlane: chunk, class, group, the row loop's bound and every row's base address stay in
    //  scalar registers, and the quad stores take the scalar base + a 32-bit lane offset)
    const uint32_t wg = xcd_contiguous(blockIdx.x, gridDim.x, flags & 1u);
    uint32_t g = (uint32_t) __builtin_amdgcn_readfirstlane((int) (wg * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    if (g >= nchunks * 4u * per_class) return;
    const uint32_t chunk = g % nchunks;
    g /= nchunks;
    const uint32_t c = g & 3u, i0 = (g >> 2) * ROWS;                                 // rows lr = c + 4 (i0 + j), j < ROWS
    const uint32_t head = (4u - ((c * (S & 3u)) & 3u)) & 3u;                         // samples in front of the first aligned quad
    const uint32_t nqa = (S - head) >> 2, tail0 = head + (nqa << 2), ht = head + (S - tail0);
    // ---- the group's row contexts, one lane per row (2^-120 on every row term: see above)
    float ca0, ca1, ca2, cc;
    {
        const uint32_t lr = c + 4u * (i0 + (lane & (ROWS - 1u)));
        const uint32_t xl = lr / S, y = lr - xl * S;
        const float px = (float) (xb + xl) * h - 2.5f, py = (float) y * h - 2.5f;
        const SkelX sx = skel_x(px);
        const SkelY sy = skel_y(py);
        float ay = py;
        if (SCENE == SCENE_ANIM) ay = py - sp.p[0];
        ca0 = (sx.ax + sy.sy) + nud; ca1 = (sx.sx15 + sy.ay) + nud; ca2 = (sx.mx + sy.sy) + nud;
        cc = (px * px + ay * ay) + nud;
    }
    uint32_t rows = 0;                                                               // valid rows of the group (a prefix)
    if (c + 4u * i0 < nrows) { rows = (nrows - 1u - c - 4u * i0) / 4u + 1u; if (rows > ROWS) rows = ROWS; }
    // ---- this wave's aligned quads, z terms hoisted over the rows
    {
        v2f sz[2 * QPL], mz[2 * QPL], az[2 * QPL], z2[2 * QPL];
        uint32_t zq[QPL];
        bool on[QPL];
#pragma unroll
        for (int u = 0; u < QPL; u++) {
            const uint32_t q = chunk * (64u * QPL) + lane + 64u * u;
            on[u] = q < nqa;
            zq[u] = head + (q << 2);
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const v2f pz = mk2((float) (zq[u] + 2u * p) * h - 2.5f, (float) (zq[u] + 2u * p + 1u) * h - 2.5f);
                az[2 * u + p] = along_sq2(pz, -0.25f, 0.5f);
                sz[2 * u + p] = sq2(abs2(pz) - 0.25f);
                mz[2 * u + p] = min2(sq2(pz - (-0.25f)), sq2(pz - 2.75f));
                z2[2 * u + p] = pz * pz;
            }
        }
        for (uint32_t j = 0; j < rows; j++) {
            const float a0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ca0), j));
            const float a1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ca1), j));
            const float a2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ca2), j));
            const float cs = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cc), j));
            char *__restrict__ o = reinterpret_cast<char *>(out + (uint64_t) (c + 4u * (i0 + j)) * S);
#pragma unroll
            for (int u = 0; u < QPL; u++) {
                if (on[u]) {
                    v2f r[2];
#pragma unroll
                    for (int p = 0; p < 2; p++) {
                        const v2f m = min2(min2(a0 + sz[2 * u + p], a1 + mz[2 * u + p]), a2 + az[2 * u + p]);
                        const v2f d1 = sqrt_rn_normal2_nonudge(m) - 0.1f;
                        const v2f d2 = sqrt_rn_normal2_nonudge(cs + z2[2 * u + p]) - 1.0f;
                        r[p] = smooth_min_half2(d1, d2);
                    }
                    const v4f v = {r[0].x, r[0].y, r[1].x, r[1].y};
                    uint32_t off = zq[u] << 2;
                    asm volatile("" : "+v"(off));          // (keeps the zero-extension in this block: scalar base + 32-bit lane offset)
                    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(o + off));
                    else *reinterpret_cast<v4f *>(o + off) = v;
                }
            }
        }
    }
    // ---- the rows' unaligned ends (ht <= 6 samples per row), by the wave of the first chunk: lane = (row, end sample) with eight slots per
    // row -- shifts instead of the two integer divisions a dense numbering costs (they were a tenth of the kernel at 512^3: one extra sample
    // per 513-sample row, 0.577 against 0.621 of the HBM peak at N = 511) -- rows j and j + 8 in the two components of one packed
    // evaluation (same slot, so the same z terms), the row contexts fetched from the lanes that hold them (ds_bpermute: no LDS memory)
    static_assert(ROWS <= 16, "eight slots per row, two rows per lane");
    if (chunk == 0 && ht) {
        const uint32_t k = lane & 7u, ja = lane >> 3, jb = ROWS > 8u ? ja + 8u : ja;
        const uint32_t z = k < head ? k : tail0 + (k - head);
        const int sa = (int) (ja << 2), sb = (int) (jb << 2);
#define SDM_ROWCTX2(v) mk2(__builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(sa, __builtin_bit_cast(int, v))), \
                           __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(sb, __builtin_bit_cast(int, v))))
        const v2f a0 = SDM_ROWCTX2(ca0), a1 = SDM_ROWCTX2(ca1), a2 = SDM_ROWCTX2(ca2), cs = SDM_ROWCTX2(cc);
#undef SDM_ROWCTX2
        const float pz1 = (float) z * h - 2.5f;
        const v2f pz = mk2(pz1, pz1);
        const v2f eaz = along_sq2(pz, -0.25f, 0.5f), esz = sq2(abs2(pz) - 0.25f), emz = min2(sq2(pz - (-0.25f)), sq2(pz - 2.75f));
        const v2f m = min2(min2(a0 + esz, a1 + emz), a2 + eaz);
        const v2f d1 = sqrt_rn_normal2_nonudge(m) - 0.1f;
        const v2f d2 = sqrt_rn_normal2_nonudge(cs + pz * pz) - 1.0f;
        const v2f r = smooth_min_half2(d1, d2);
        if (k < ht && ja < rows) out[(uint64_t) (c + 4u * (i0 + ja)) * S + z] = r.x;
        if (ROWS > 8u && k < ht && jb < rows) out[(uint64_t) (c + 4u * (i0 + jb)) * S + z] = r.y;
    }
#endif
}

// ------------------------------------------------------------------------------------------
// host-side launchers (scene dispatch)
// ------------------------------------------------------------------------------------------
#define SDM_DISPATCH(scene, CALL)                                                                 \
    switch (scene) {                                                                              \
    case SCENE_SDOBJ: { constexpr int S = SCENE_SDOBJ; CALL; } break;                             \
    case SCENE_SPHERE: { constexpr int S = SCENE_SPHERE; CALL; } break;                           \
    case SCENE_CSG3: { constexpr int S = SCENE_CSG3; CALL; } break;                               \
    case SCENE_ANIM: { constexpr int S = SCENE_ANIM; CALL; } break;                               \
    case SCENE_PROGRAM: { constexpr int S = SCENE_PROGRAM; CALL; } break;                         \
    default: return hipErrorInvalidValue;                                                         \
    }
// the field passes (dense sampling, the preview) also take SCENE_BULB, which the mesh path refuses (api.hip mesh_scene_ok)
#define SDM_DISPATCH_FIELD(scene, CALL)                                                           \
    switch (scene) {                                                                              \
    case SCENE_SDOBJ: { constexpr int S = SCENE_SDOBJ; CALL; } break;                             \
    case SCENE_SPHERE: { constexpr int S = SCENE_SPHERE; CALL; } break;                           \
    case SCENE_CSG3: { constexpr int S = SCENE_CSG3; CALL; } break;                               \
    case SCENE_ANIM: { constexpr int S = SCENE_ANIM; CALL; } break;                               \
    case SCENE_PROGRAM: { constexpr int S = SCENE_PROGRAM; CALL; } break;                         \
    case SCENE_BULB: { constexpr int S = SCENE_BULB; CALL; } break;                               \
    default: return hipErrorInvalidValue;                                                         \
    }

hipError_t launch_refine_mask(hipStream_t st, int grid, int scene, const SceneParams *sp, const Point *vox,
                              const uint32_t *n_ptr, Point cs, uint8_t *mask, uint32_t *tilecnt, int chunk, uint32_t *signs) {
    if (chunk == 1) {            // one lane per parent
        SDM_DISPATCH(scene, hipLaunchKernelGGL(k_refine_mask1<S>, dim3(grid), dim3(256), 0, st, vox, n_ptr, cs, sp, mask, tilecnt, signs));
        return hipGetLastError();
    }
    if (chunk == 8) { SDM_DISPATCH(scene, hipLaunchKernelGGL((k_refine_mask<S, 8>), dim3(grid), dim3(256), 0, st, vox, n_ptr, cs, sp, mask, tilecnt, signs)); }
    else if (chunk == 16) { SDM_DISPATCH(scene, hipLaunchKernelGGL((k_refine_mask<S, 16>), dim3(grid), dim3(256), 0, st, vox, n_ptr, cs, sp, mask, tilecnt, signs)); }
    else { SDM_DISPATCH(scene, hipLaunchKernelGGL((k_refine_mask<S, 32>), dim3(grid), dim3(256), 0, st, vox, n_ptr, cs, sp, mask, tilecnt, signs)); }
    return hipGetLastError();
}

hipError_t launch_scan_tiles(hipStream_t st, uint32_t *cnt, const uint32_t *n_ptr, uint32_t items_per_tile,
                             uint32_t *total_out, uint32_t cap, uint32_t *status) {
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, st, cnt, n_ptr, items_per_tile, total_out, cap, status);
    return hipGetLastError();
}

hipError_t launch_refine_emit(hipStream_t st, int grid, const Point *vox, const uint32_t *n_ptr, Point cs,
                              const uint8_t *mask, const uint32_t *tilecnt, Point *out, uint32_t cap, uint32_t *n_out,
                              uint32_t *status, const uint32_t *signs, uint8_t *cubeidx, uint32_t *mtilecnt, bool prescan) {
    if (prescan)        // many tiles: one single-workgroup scan in front instead of every workgroup re-summing all counts before it
        hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, st, const_cast<uint32_t *>(tilecnt), n_ptr, 256u, n_out, cap, status);
    hipLaunchKernelGGL(k_refine_emit, dim3(grid), dim3(256), 0, st, vox, n_ptr, cs, mask, tilecnt, out, cap, n_out, status, signs,
                       cubeidx, mtilecnt, prescan ? 1u : 0u);
    return hipGetLastError();
}

// z segments per column (units of the column kernels): the largest power of two <= pref that divides the slab's z extent
static uint32_t col_segments(uint32_t zs, int pref) {
    uint32_t q = 1;
    while (q * 2u <= (uint32_t) (pref < 1 ? 1 : (pref > (int) SDM_COL_SEGS_MAX ? (int) SDM_COL_SEGS_MAX : pref)) && zs % (q * 2u) == 0u) q *= 2u;
    return q;
}

hipError_t launch_col_refine(hipStream_t st, int scene, const SceneParams *sp, uint32_t x0, uint32_t x1, uint32_t z0, uint32_t z1, int J,
                             uint8_t *g_list1, uint8_t *g_mask2, uint8_t *g_mask3, uint4 *colinfo, uint32_t *zero, uint32_t nzero,
                             void *ff, size_t ff_bytes, int segs_pref) {
    const uint32_t ncols = (x1 - x0) * MESH_GENERATION_INIT_FACTOR;
    if (ncols == 0 || z1 <= z0) return hipErrorInvalidValue;
    const uint32_t segs = col_segments(z1 - z0, segs_pref);
    SDM_DISPATCH(scene, hipLaunchKernelGGL(k_col_refine<S>, dim3(ncols * segs), dim3(256), 0, st, x0, z0, z1 - z0, (uint32_t) J, sp, g_list1, g_mask2, g_mask3,
                                           colinfo, zero, nzero, (uint4 *) ff, ff ? (uint32_t) (ff_bytes / 16) : 0u, segs));
    return hipGetLastError();
}

hipError_t launch_col_emit(hipStream_t st, uint32_t x0, uint32_t x1, uint32_t z0, uint32_t z1, int J, const uint8_t *g_list1,
                           const uint8_t *g_mask2, const uint8_t *g_mask3, const uint4 *colinfo, Point *out, uint32_t cap,
                           uint32_t *level_ctr, uint32_t *status, int segs_pref) {
    const uint32_t ncols = (x1 - x0) * MESH_GENERATION_INIT_FACTOR;
    const uint32_t segs = col_segments(z1 - z0, segs_pref);
    hipLaunchKernelGGL(k_col_emit, dim3(ncols * segs), dim3(256), 0, st, x0, z0, z1 - z0, ncols * segs, (uint32_t) J, g_list1, g_mask2, g_mask3, colinfo, out, cap,
                       level_ctr, status, segs);
    return hipGetLastError();
}

hipError_t launch_sum_tiles(hipStream_t st, const uint32_t *tilecnt, const uint32_t *n_ptr, uint32_t *total_out) {
    hipLaunchKernelGGL(k_sum_tiles, dim3(1), dim3(1024), 0, st, tilecnt, n_ptr, total_out);
    return hipGetLastError();
}

hipError_t launch_refine2_mask(hipStream_t st, int grid, int scene, const SceneParams *sp, const Point *vox, const uint32_t *n_ptr,
                               Point cs1, uint8_t *mask1, uint64_t *mask2, uint32_t *tilecnt, uint32_t *n_mid) {
    SDM_DISPATCH(scene, hipLaunchKernelGGL(k_refine2_mask<S>, dim3(grid), dim3(256), 0, st, vox, n_ptr, cs1, sp, mask1, mask2, tilecnt, n_mid));
    return hipGetLastError();
}

hipError_t launch_refine2_emit(hipStream_t st, int grid, const Point *vox, const uint32_t *n_ptr, Point cs1, const uint8_t *mask1,
                               const uint64_t *mask2, const uint32_t *tilecnt, Point *out, uint32_t cap, uint32_t *n_out,
                               uint32_t *status) {
    hipLaunchKernelGGL(k_refine2_emit, dim3(grid), dim3(256), 0, st, vox, n_ptr, cs1, mask1, mask2, tilecnt, out, cap, n_out, status);
    return hipGetLastError();
}

hipError_t launch_refine_expand_raw(hipStream_t st, int grid, const Point *vox, uint32_t n, Point cs, const uint8_t *mask,
                                    Point *out) {
    hipLaunchKernelGGL(k_refine_expand_raw, dim3(grid), dim3(256), 0, st, vox, n, cs, mask, out);
    return hipGetLastError();
}

hipError_t launch_classify(hipStream_t st, int grid, int scene, const SceneParams *sp, const Point *vox,
                           const uint32_t *n_ptr, Point vs, uint8_t *cubeidx, uint32_t *tilecnt) {
    SDM_DISPATCH(scene, hipLaunchKernelGGL(k_classify<S>, dim3(grid), dim3(256), 0, st, vox, n_ptr, vs, sp, cubeidx, tilecnt));
    return hipGetLastError();
}

hipError_t launch_tile_dedupe(hipStream_t st, int grid, const Point *vox, const uint32_t *n_ptr, Point vs, const uint8_t *cubeidx,
                              const uint32_t *tilecnt, uint32_t *trioff, uint32_t *tri_e, Vertex *uverts, uint4 *tdesc,
                              uint32_t ucap, uint32_t tcap, uint32_t *ntri_out, uint32_t *ntiles_out, uint32_t *status, uint64_t *dbg,
                              bool prescan, uint32_t *colstart, uint32_t *coltile) {
    // voxels per level-0 cell edge = (5 / 32) / vs.x, a power of two for every field the lattice path takes
    uint32_t colshift = 0;
    {
        const float ratio = (MESH_GENERATION_BB_SIZE / (float) MESH_GENERATION_INIT_FACTOR) / vs.x;
        while ((float) (1u << (colshift + 1u)) <= ratio && colshift < 20u) colshift++;
    }
    if (prescan)
        hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, st, const_cast<uint32_t *>(tilecnt), n_ptr, 256u, ntri_out, tcap, status);
    hipLaunchKernelGGL(k_tile_dedupe, dim3(grid), dim3(256), 0, st, vox, n_ptr, vs, cubeidx, tilecnt, trioff, tri_e, uverts, tdesc,
                       ucap, tcap, ntri_out, ntiles_out, status, dbg, prescan ? 1u : 0u, colstart, coltile, colshift);
    return hipGetLastError();
}

hipError_t launch_tile_prefix(hipStream_t st, const uint4 *tdesc, const uint32_t *ntiles_ptr, uint32_t *tpre, uint32_t *nu_out,
                              uint32_t *coarse) {
    hipLaunchKernelGGL(k_tile_prefix, dim3(1), dim3(1024), 0, st, tdesc, ntiles_ptr, tpre, nu_out, coarse);
    return hipGetLastError();
}

hipError_t launch_project(hipStream_t st, int grid, int scene, const SceneParams *sp, Vertex *uverts, const uint4 *tdesc,
                          const uint32_t *tpre, const uint32_t *coarse, const uint32_t *ntiles_ptr, uint32_t *status, int lanes,
                          uint32_t *prof, uint64_t *diag, uint64_t *wkeys, uint32_t wsize, uint32_t *vslot, uint32_t *wrep, uint2 *wfm,
                          uint32_t wmode, uint32_t *nu_out, uint32_t batch_shift) {
    ClaimArgs ca;
    ca.wkeys = wkeys; ca.vslot = vslot; ca.wrep = wrep; ca.wfm = wfm; ca.wsize = wsize; ca.wmode = wmode;
    ca.bsh = batch_shift > 6u ? 6u : batch_shift;
    if (lanes == 1) { SDM_DISPATCH(scene, hipLaunchKernelGGL(k_project1<S>, dim3(grid), dim3(256), 0, st, uverts, tdesc, tpre, coarse, ntiles_ptr, sp, status, prof, diag, ca, nu_out)); }
    else if (lanes == 2) { SDM_DISPATCH(scene, hipLaunchKernelGGL((k_project<S, 2>), dim3(grid), dim3(256), 0, st, uverts, tdesc, tpre, coarse, ntiles_ptr, sp, status, prof, diag, ca, nu_out)); }
    else { SDM_DISPATCH(scene, hipLaunchKernelGGL((k_project<S, 4>), dim3(grid), dim3(256), 0, st, uverts, tdesc, tpre, coarse, ntiles_ptr, sp, status, prof, diag, ca, nu_out)); }
    return hipGetLastError();
}

hipError_t launch_flip(hipStream_t st, int grid, int scene, const SceneParams *sp, const Vertex *uverts, const uint32_t *tri_e,
                       const uint32_t *ntri_ptr, uint8_t *flip, int lanes) {
    if (lanes == 1) { SDM_DISPATCH(scene, hipLaunchKernelGGL(k_flip1<S>, dim3(grid), dim3(256), 0, st, uverts, tri_e, ntri_ptr, sp, flip)); }
    else { SDM_DISPATCH(scene, hipLaunchKernelGGL(k_flip<S>, dim3(grid), dim3(256), 0, st, uverts, tri_e, ntri_ptr, sp, flip)); }
    return hipGetLastError();
}

hipError_t launch_flip_claim(hipStream_t st, int flip_grid, int claim_grid, int scene, const SceneParams *sp, const Vertex *uverts,
                            const uint32_t *tri_e, const uint32_t *ntri_ptr, uint8_t *flip, int lanes, const uint4 *tdesc,
                            const uint32_t *ntiles_ptr, uint64_t *wkeys, uint32_t wsize, uint32_t *vslot, uint32_t *status, uint32_t wmode,
                            uint32_t *wrep, uint2 *wfm) {
    const dim3 grid(flip_grid + claim_grid);
    const uint32_t w = wmode;
    if (lanes == 1) { SDM_DISPATCH(scene, hipLaunchKernelGGL((k_flip_claim<S, 1>), grid, dim3(256), 0, st, uverts, tri_e, ntri_ptr, sp, flip, (uint32_t) claim_grid, tdesc, ntiles_ptr, wkeys, wsize, vslot, status, w, wrep, wfm)); }
    else { SDM_DISPATCH(scene, hipLaunchKernelGGL((k_flip_claim<S, 4>), grid, dim3(256), 0, st, uverts, tri_e, ntri_ptr, sp, flip, (uint32_t) claim_grid, tdesc, ntiles_ptr, wkeys, wsize, vslot, status, w, wrep, wfm)); }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Row N4: ray-march preview (render_device.hpp): one thread per pixel, 8 x 8 pixel workgroups
// ------------------------------------------------------------------------------------------
template <int SCENE>
__global__ __launch_bounds__(64) void k_render(RenderParams r, const SceneParams *__restrict__ spp, uchar4 *__restrict__ out) {
    const uint32_t x = blockIdx.x * 8u + (threadIdx.x & 7u), y = blockIdx.y * 8u + (threadIdx.x >> 3);
    if (x >= r.width || y >= r.height) return;
    out[(size_t) y * r.width + x] = render_pixel<SCENE>(x, y, r, *spp);
}

hipError_t launch_render(hipStream_t st, int scene, const SceneParams *sp, const float *cam13, float fov_fac, float screen_x,
                         float screen_y, float gtex_x, float gtex_y, uint32_t width, uint32_t height, void *rgba) {
    RenderParams r;
    for (int i = 0; i < 3; i++) { r.pos[i] = cam13[i]; r.fwd[i] = cam13[3 + i]; r.up[i] = cam13[6 + i]; r.right[i] = cam13[9 + i]; }
    r.fov_fac = fov_fac; r.screen_x = screen_x; r.screen_y = screen_y; r.gtex_x = gtex_x; r.gtex_y = gtex_y;
    r.width = width; r.height = height;
    const dim3 grid((width + 7) / 8, (height + 7) / 8);
    SDM_DISPATCH_FIELD(scene, hipLaunchKernelGGL(k_render<S>, grid, dim3(64), 0, st, r, sp, (uchar4 *) rgba));
    return hipGetLastError();
}

hipError_t launch_raw_desc(hipStream_t st, int grid, const uint32_t *ntri_ptr, uint4 *tdesc, uint8_t *flip, uint32_t *ntiles_out,
                           uint32_t *nu_out, uint32_t *tri_e) {
    hipLaunchKernelGGL(k_raw_desc, dim3(grid), dim3(256), 0, st, ntri_ptr, tdesc, flip, ntiles_out, nu_out, tri_e);
    return hipGetLastError();
}

hipError_t launch_tile_claim(hipStream_t st, int grid, const Vertex *uverts, const uint4 *tdesc, const uint32_t *ntiles_ptr,
                             uint64_t *wkeys, uint32_t wsize, uint32_t *vslot, uint32_t *status, uint32_t wmode, uint32_t *wrep, uint2 *wfm) {
    hipLaunchKernelGGL(k_tile_claim, dim3(grid), dim3(256), 0, st, uverts, tdesc, ntiles_ptr, wkeys, wsize, vslot, status, wmode, wrep, wfm);
    return hipGetLastError();
}

// first -> rank -> scan -> finish (the claim has already happened: in k_project or k_weld_claim)
hipError_t launch_weld(hipStream_t st, int grid, const Vertex *uverts, const uint4 *tdesc, const uint32_t *ntiles_ptr,
                       uint32_t *tri_e, const uint8_t *flip, uint32_t *first_ord, uint2 *wfm, const uint32_t *wrep,
                       uint32_t *vslot, uint32_t *wv, uint32_t *torder, uint32_t *tfirst, uint32_t *tfpre,
                       Vertex *vertices, uint32_t vcap, uint32_t *nverts_out, uint32_t *indices, uint32_t *status,
                       int flip_scene, const SceneParams *sp, int part, const uint32_t *pub_src, uint32_t *pub_dst, uint32_t pub_words,
                       uint32_t pub_ovf, uint32_t pub_nvert, int tf_threads, const uint32_t *colstart, const uint32_t *coltile, uint32_t *colvloc,
                       uint32_t *colvstart) {
    if (tf_threads < 128 || tf_threads > 1024 || tf_threads % 64) tf_threads = 256;
    // part: 0 = the whole weld, 1 = k_tile_first only, 2 = everything behind it (so that a stage marker fits between)
    const bool stop_after_first = part == 1;
    if (part == 2) goto rest;
    if (flip_scene >= 0) {
        switch (flip_scene) {
        case SCENE_SDOBJ: hipLaunchKernelGGL(k_tile_first<SCENE_SDOBJ>, dim3(grid), dim3(tf_threads), 0, st, tdesc, ntiles_ptr, tri_e, flip, uverts, sp, vslot, first_ord, wrep, wfm); break;
        case SCENE_SPHERE: hipLaunchKernelGGL(k_tile_first<SCENE_SPHERE>, dim3(grid), dim3(tf_threads), 0, st, tdesc, ntiles_ptr, tri_e, flip, uverts, sp, vslot, first_ord, wrep, wfm); break;
        case SCENE_CSG3: hipLaunchKernelGGL(k_tile_first<SCENE_CSG3>, dim3(grid), dim3(tf_threads), 0, st, tdesc, ntiles_ptr, tri_e, flip, uverts, sp, vslot, first_ord, wrep, wfm); break;
        case SCENE_ANIM: hipLaunchKernelGGL(k_tile_first<SCENE_ANIM>, dim3(grid), dim3(tf_threads), 0, st, tdesc, ntiles_ptr, tri_e, flip, uverts, sp, vslot, first_ord, wrep, wfm); break;
        case SCENE_PROGRAM: hipLaunchKernelGGL(k_tile_first<SCENE_PROGRAM>, dim3(grid), dim3(tf_threads), 0, st, tdesc, ntiles_ptr, tri_e, flip, uverts, sp, vslot, first_ord, wrep, wfm); break;
        default: return hipErrorInvalidValue;
        }
    } else {
        hipLaunchKernelGGL(k_tile_first<-1>, dim3(grid), dim3(tf_threads), 0, st, tdesc, ntiles_ptr, tri_e, flip, uverts, sp, vslot, first_ord, wrep, wfm);
    }
    if (stop_after_first) return hipGetLastError();
rest:
    hipLaunchKernelGGL(k_tile_rank, dim3(grid), dim3(256), 0, st, tdesc, ntiles_ptr, first_ord, wfm, vslot, wv, torder,
                       tfirst, colstart, coltile, colvloc);
    // tfpre == nullptr: few enough tiles for k_tile_finish to prefix the first-seen counts itself
    if (tfpre) hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, st, tfirst, ntiles_ptr, tfpre, nverts_out);
    hipLaunchKernelGGL(k_tile_finish, dim3(grid), dim3(256), 0, st, uverts, tdesc, ntiles_ptr, tfpre, tfirst, torder, tri_e, vslot,
                       wv, vertices, vcap, indices, status, nverts_out, pub_src, pub_dst, pub_words, pub_ovf, pub_nvert, colstart, coltile,
                       (const uint32_t *) colvloc, colvstart);
    return hipGetLastError();
}

hipError_t launch_expand_compact(hipStream_t st, int grid, const Vertex *uverts, const uint32_t *tri_e, const uint8_t *flip,
                                 const uint32_t *ntri_ptr, Triangle *out, uint32_t cap) {
    hipLaunchKernelGGL(k_expand_compact, dim3(grid), dim3(256), 0, st, uverts, tri_e, flip, ntri_ptr, out, cap);
    return hipGetLastError();
}

hipError_t launch_expand_raw(hipStream_t st, int grid, const Vertex *uverts, const uint32_t *tri_e, const uint8_t *flip,
                             const uint8_t *cubeidx, const uint32_t *trioff, uint32_t n, Triangle *out) {
    hipLaunchKernelGGL(k_expand_raw, dim3(grid), dim3(256), 0, st, uverts, tri_e, flip, cubeidx, trioff, n, out);
    return hipGetLastError();
}

hipError_t launch_triangles_direct(hipStream_t st, int scene, const SceneParams *sp, const Point *vox, uint32_t n, Point vs,
                                   Triangle *out) {
    const int grid = (int) ((n + 127u) / 128u);
    if (grid == 0) return hipSuccess;
    SDM_DISPATCH(scene, hipLaunchKernelGGL(k_triangles_direct<S>, dim3(grid), dim3(128), 0, st, vox, n, vs, sp, out));
    return hipGetLastError();
}

hipError_t launch_sample_dense_rows(hipStream_t st, int grid, int scene, const SceneParams *sp, float *out, uint32_t N, bool nt,
                                    uint32_t x0, uint32_t x1, bool aligned_rows, bool nt_rows, int occ, bool xcd) {
    if (x1 <= x0) return hipSuccess;
#if defined(SDM_NO_PAIRS)
    aligned_rows = false;                    // (the aligned kernel is written on the packed 2-vector helpers)
#endif
    if ((scene == SCENE_SDOBJ || scene == SCENE_ANIM) && aligned_rows && ((uintptr_t) out & 15u) == 0 && N >= 3u) {
        // whole-line stores, z terms hoisted over the same-phase rows of a wave (k_sample_dense_rows_aligned), one wave per (chunk, group).
        // Measured at 512^3 (sd_obj, plain stores, same box): 8 rows x 2 quads per lane 0.127-0.131 ms, 4 x 2: 0.134-0.147, 8 x 1: 0.133-0.140,
        // 16 x 2: 0.125-0.133 on another box where 8 x 2 gave 0.120-0.126 (long waves leave the SIMDs half empty in their last round and
        // hide their dependent chains badly; short ones lose the hoisting); nontemporal stores cost this kernel 5-15 %, hence
        // `nt_rows` (option dense_nt = 2) is off by default here
        // Rows longer than 512 samples: FOUR quads per lane, so that a wave still writes whole rows (1024^3, sd_obj: 8 x 4 0.844 ms = 64 % of the HBM
        // peak against 1.084 ms = 50 % for 8 x 2, 0.89-0.94 for 4 x 4 / 4 x 2 / 16 x 4: half-row pieces from twice as many waves cost the write path more
        // than the 64 extra registers cost occupancy; at 512^3, where a row is one 8 x 2 piece anyway, both shapes time the same).
        // (Balancing the chunks of a longer row instead -- 1449-sample rows as two chunks of 3 x 64 quads rather than 4 x 64 + a 41 % full one -- was
        // slower, 2.97 against 2.84 ms: as few pieces per row as possible is what counts.)
        // Rows of at most 64 quads (256^3 and below): ONE quad per lane and sixteen rows per wave -- with two quads per lane half of them lie behind
        // the row's end, and eight one-quad rows are too little work for the wave's 200-instruction prologue.
        const uint32_t nq = (N + 1) >> 2, Q = nq <= 64u ? 1u : (nq <= 128u ? 2u : 4u), R = Q == 1u ? 16u : 8u;
        const uint64_t nrows = (uint64_t) (N + 1) * (x1 - x0), nchunks = ((uint64_t) nq + 64ull * Q - 1ull) / (64ull * Q);
        const uint64_t ngroups = nchunks * 4ull * ((((nrows + 3ull) >> 2) + R - 1ull) / R);
        const int g2 = (int) std::min<uint64_t>((ngroups + 3ull) / 4ull, 0x7FFFFFFFull);
        // Workgroups per CU and their order (MEASUREMENTS.md "Round 4, last"; same boxes, ms at 512^3 / 1024^3):
        //   ids dealt round robin to the XCDs as the hardware does, no cap (8 / 5 workgroups per CU by registers)   0.124 / 0.903
        //   the same, capped at 3 per CU through an LDS allotment the kernel never touches                          0.111 / 0.861
        //   an XCD's workgroups on one contiguous eighth of the row groups (`xcd`), no cap                           0.104 / 0.873
        //   `xcd`, capped at 3                                                                                       0.110 / 0.805
        // The write stream wants few rows in flight per L2 and ascending addresses from each; the arithmetic wants waves: two quads per
        // lane (rows of <= 512 samples) keep every wave, four quads per lane run best with three workgroups per CU.  Rows of <= 64 quads
        // (256^3): no difference either way.  `occ` > 0 = the caller's cap (option dense_occ), < 0 = none.
        const uint32_t flags = xcd && Q > 1u ? 1u : 0u;
        if (occ == 0) occ = Q == 4u ? 3 : (Q == 2u && !flags ? 3 : -1);
        const uint32_t lds = occ >= 1 && occ <= 8 ? (160u * 1024u / (uint32_t) occ) & ~1023u : 0u;
#define SDM_AL(SC, NTF, RR, QQ) { if (lds > 65536u) (void) hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sample_dense_rows_aligned<SC, NTF, RR, QQ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds); \
        hipLaunchKernelGGL((k_sample_dense_rows_aligned<SC, NTF, RR, QQ>), dim3(g2), dim3(256), lds, st, out, N, sp, x0, x1, flags); }
#define SDM_ALQ(SC, NTF) { if (Q == 4) SDM_AL(SC, NTF, 8, 4) else if (Q == 2) SDM_AL(SC, NTF, 8, 2) else SDM_AL(SC, NTF, 16, 1) }
        if (scene == SCENE_SDOBJ) { if (nt_rows) SDM_ALQ(SCENE_SDOBJ, true) else SDM_ALQ(SCENE_SDOBJ, false) }
        else { if (nt_rows) SDM_ALQ(SCENE_ANIM, true) else SDM_ALQ(SCENE_ANIM, false) }
#undef SDM_ALQ
#undef SDM_AL
        return hipGetLastError();
    }
    if (grid <= 0) {       // one group of 16 rows per wave: the dispatcher balances the waves (a fixed grid leaves 2 or 3 groups per wave)
        const uint64_t ngroups = ((uint64_t) (N + 1) * (x1 - x0) + 15ull) / 16ull;
        grid = (int) std::min<uint64_t>((ngroups + 3ull) / 4ull, 0x7FFFFFFFull);
    }
    if (nt) { SDM_DISPATCH_FIELD(scene, hipLaunchKernelGGL((k_sample_dense_rows<S, true>), dim3(grid), dim3(256), 0, st, out, N, sp, x0, x1)); }
    else { SDM_DISPATCH_FIELD(scene, hipLaunchKernelGGL((k_sample_dense_rows<S, false>), dim3(grid), dim3(256), 0, st, out, N, sp, x0, x1)); }
    return hipGetLastError();
}

hipError_t launch_sample_dense(hipStream_t st, int grid, int scene, const SceneParams *sp, float *out, uint32_t N, bool nt,
                               bool contig, uint32_t x0, uint32_t x1) {
    const uint64_t plane = (uint64_t) (N + 1) * (N + 1), total = plane * (x1 - x0), first = plane * x0;
    const uint32_t cg = contig ? 1u : 0u;
    if (total == 0) return hipSuccess;
    if (nt) { SDM_DISPATCH_FIELD(scene, hipLaunchKernelGGL((k_sample_dense<S, true>), dim3(grid), dim3(256), 0, st, out, N, total, sp, cg, first)); }
    else { SDM_DISPATCH_FIELD(scene, hipLaunchKernelGGL((k_sample_dense<S, false>), dim3(grid), dim3(256), 0, st, out, N, total, sp, cg, first)); }
    return hipGetLastError();
}

hipError_t launch_stream_fill(hipStream_t st, int grid, float *out, uint64_t total, float value, bool xcd) {
    if (grid <= 0) grid = (int) std::min<uint64_t>(((total >> 2) + 255ull) / 256ull, 0x7FFFFFFFull);      // one quad per thread
    hipLaunchKernelGGL(k_stream_fill, dim3(grid), dim3(256), 0, st, out, total, value, xcd ? 1u : 0u);
    return hipGetLastError();
}

// exact unsigned division of any 32-bit n by d (2 <= d < 2^31) as mulhi + shift (+ fixup): Granlund & Montgomery / libdivide
static DivMagic make_div_magic(uint32_t d) {
    DivMagic r;
    uint32_t l = 0;
    while ((1ull << l) < d) l++;                                    // ceil(log2 d)
    const uint64_t m = ((1ull << 32) * ((1ull << l) - d)) / d + 1ull;   // low 32 bits of the 33-bit multiplier
    r.m = (uint32_t) m; r.add = 1u; r.sh = l - 1u;                    // q = (((n - hi) >> 1) + hi) >> (l - 1)
    return r;
}

hipError_t launch_sample_dense_q1(hipStream_t st, int scene, const SceneParams *sp, float *out, uint32_t N, bool nt, uint32_t x0, uint32_t x1, bool xcd) {
    // planes x0 .. x1 - 1 of the volume into out[0 ..).  The kernel indexes relative to its launch's first plane in 32 bits: a range of
    // 2^32 samples or more goes out as several launches of whole planes (2048^3: three) -- every one in the shape the store stream likes.
    if (N < 1u || x1 < x0 || x1 > N + 1u) return hipErrorInvalidValue;
    const uint64_t plane = (uint64_t) (N + 1) * (N + 1);
    // planes per launch: a multiple of four, so that every launch's first sample is 16-byte aligned in `out` whatever the parity of N + 1
    const uint32_t batch = (uint32_t) std::max<uint64_t>(4, (0xFFFFF000ull / plane) & ~3ull);
    const DivMagic dS = make_div_magic(N + 1u);
    for (uint32_t xa = x0; xa < x1; xa += batch) {
        const uint32_t xb = std::min(x1, xa + batch);
        const uint64_t total = plane * (xb - xa);
        float *o = out + plane * (xa - x0);
        const uint32_t grid = (uint32_t) (((total + 3ull) / 4ull + 255ull) / 256ull);
        if (nt) { SDM_DISPATCH_FIELD(scene, hipLaunchKernelGGL((k_sample_dense_q1<S, true>), dim3(grid), dim3(256), 0, st, o, N, (uint32_t) total, sp, dS, 5.0f / (float) N, xa, xcd ? 1u : 0u)); }
        else { SDM_DISPATCH_FIELD(scene, hipLaunchKernelGGL((k_sample_dense_q1<S, false>), dim3(grid), dim3(256), 0, st, o, N, (uint32_t) total, sp, dS, 5.0f / (float) N, xa, xcd ? 1u : 0u)); }
    }
    return hipGetLastError();
}

hipError_t launch_create_field(hipStream_t st, Point *vox, uint32_t x0, uint32_t x1, uint32_t z0, uint32_t z1, uint32_t *n_out,
                               uint32_t *zero, uint32_t nzero, void *ff, size_t ff_bytes) {
    hipLaunchKernelGGL(k_create_field, dim3(ff ? 1024 : 128), dim3(256), 0, st, vox, x0, x1, z0, z1, n_out, zero, nzero, (uint4 *) ff,
                       ff ? (uint32_t) (ff_bytes / 16) : 0u);       // (no table pointer -> nothing to clear, whatever size was passed)
    return hipGetLastError();
}

hipError_t launch_raw_count(hipStream_t st, int grid, const Triangle *raw, const uint32_t *n_ptr, uint32_t *tilecnt) {
    hipLaunchKernelGGL(k_raw_count, dim3(grid), dim3(256), 0, st, raw, n_ptr, tilecnt);
    return hipGetLastError();
}

hipError_t launch_raw_compact(hipStream_t st, int grid, const Triangle *raw, const uint32_t *n_ptr, const uint32_t *tileoff,
                              Triangle *out, uint32_t *tri_e, uint32_t cap) {
    hipLaunchKernelGGL(k_raw_compact, dim3(grid), dim3(256), 0, st, raw, n_ptr, tileoff, out, tri_e, cap);
    return hipGetLastError();
}

hipError_t launch_column_counts(hipStream_t st, int grid, const Point *vox, const uint32_t *n_ptr, const uint8_t *cubeidx,
                                uint32_t *colcnt) {
    hipLaunchKernelGGL(k_column_counts, dim3(grid), dim3(256), 0, st, vox, n_ptr, cubeidx, colcnt);
    return hipGetLastError();
}

hipError_t launch_merge_by_column(hipStream_t st, const Triangle *const *d_lists, const uint32_t *d_counts, uint32_t cstride, int nranks, uint32_t *d_plan,
                                  Triangle *d_out, uint32_t cap, uint32_t *total_out, uint32_t *status) {
    hipLaunchKernelGGL(k_merge_plan, dim3(1), dim3(1024), 0, st, d_counts, cstride, (uint32_t) nranks, d_plan, cap, total_out, status);
    hipLaunchKernelGGL(k_merge_copy, dim3(nranks * 1024), dim3(256), 0, st, d_lists, d_counts, cstride, d_plan, (uint32_t) nranks, d_out, cap);
    return hipGetLastError();
}


hipError_t launch_band_select(hipStream_t st, int grid, const Vertex *v, const uint32_t *v_ptr, uint32_t vcap, float up_thr, float lo_thr, bool has_up, bool has_lo,
                              uint32_t *up_ids, uint64_t *up_keys, uint32_t *lo_ids, uint32_t *ctr, uint32_t *dup_of, uint32_t *status) {
    hipLaunchKernelGGL(k_band_select, dim3(grid), dim3(256), 0, st, v, v_ptr, vcap, up_thr, lo_thr, has_up ? 1u : 0u, has_lo ? 1u : 0u, up_ids, up_keys, lo_ids, ctr,
                       dup_of, status);
    return hipGetLastError();
}

hipError_t launch_dist_match(hipStream_t st, int grid, const uint64_t *prev_keys, const uint32_t *n_prev_ptr, uint32_t ncap, uint64_t *hkeys, uint32_t *hpos, uint32_t hsize,
                             const Vertex *v, const uint32_t *lo_ids, const uint32_t *n_lo_ptr, uint32_t *dup_of, uint32_t *status) {
    hipLaunchKernelGGL(k_dist_hash_build, dim3(grid), dim3(256), 0, st, prev_keys, n_prev_ptr, ncap, hkeys, hpos, hsize);
    hipLaunchKernelGGL(k_dist_hash_lookup, dim3(grid), dim3(256), 0, st, v, lo_ids, n_lo_ptr, hkeys, hpos, hsize, dup_of, status);
    return hipGetLastError();
}

hipError_t launch_dist_number(hipStream_t st, int grid, const uint32_t *dup_of, const uint32_t *v_ptr, uint32_t vcap, uint32_t *tilecnt,
                              uint32_t *n_keep_out, uint32_t *newidx, uint32_t *status, const DistRecord &rec) {
    hipLaunchKernelGGL(k_dist_keep_count, dim3(grid), dim3(256), 0, st, dup_of, v_ptr, vcap, tilecnt);
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, st, tilecnt, v_ptr, 256u, n_keep_out, 0xFFFFFFFFu, status);
    hipLaunchKernelGGL(k_dist_newidx, dim3(std::max(grid, 1)), dim3(256), 0, st, dup_of, v_ptr, vcap, (const uint32_t *) tilecnt, newidx, rec);
    return hipGetLastError();
}

hipError_t launch_dist_pack_up(hipStream_t st, int grid, const uint32_t *up_ids, const uint32_t *n_up_ptr, uint32_t ncap, const uint32_t *newidx, const uint32_t *counts,
                               uint32_t me, uint32_t *up_g) {
    hipLaunchKernelGGL(k_dist_pack_up, dim3(grid), dim3(256), 0, st, up_ids, n_up_ptr, ncap, newidx, counts, me, up_g);
    return hipGetLastError();
}

hipError_t launch_dist_finalize(hipStream_t st, int grid, const Vertex *v, const uint32_t *v_ptr, uint32_t vcap, const uint32_t *t_ptr, uint32_t tcap,
                                const uint32_t *dup_of, const uint32_t *newidx, const uint32_t *prev_g, uint32_t pcap, const uint32_t *counts, uint32_t me,
                                uint32_t *indices, Vertex *out_v, const DistPublish &pb) {
    hipLaunchKernelGGL(k_dist_finalize, dim3(std::max(grid, 1)), dim3(256), 0, st, v, v_ptr, vcap, t_ptr, tcap, dup_of, newidx, prev_g, pcap, counts, me, indices, out_v, pb);
    return hipGetLastError();
}

hipError_t launch_merge_plan(hipStream_t st, const uint32_t *d_counts, uint32_t cstride, int nranks, uint32_t *d_plan, uint32_t cap,
                             uint32_t *total_out, uint32_t *status) {
    hipLaunchKernelGGL(k_merge_plan, dim3(1), dim3(1024), 0, st, d_counts, cstride, (uint32_t) nranks, d_plan, cap, total_out, status);
    return hipGetLastError();
}

hipError_t launch_tile_tricount(hipStream_t st, int grid, const uint8_t *cubeidx, const uint32_t *n_ptr, uint32_t *tilecnt) {
    hipLaunchKernelGGL(k_tile_tricount, dim3(grid), dim3(256), 0, st, cubeidx, n_ptr, tilecnt);
    return hipGetLastError();
}

hipError_t launch_sign_bits(hipStream_t st, int grid, const float *samples, uint64_t total, uint32_t *bits) {
    hipLaunchKernelGGL(k_sign_bits, dim3(grid), dim3(256), 0, st, samples, total, bits);
    return hipGetLastError();
}

hipError_t launch_dense_classify(hipStream_t st, int grid, bool emit, const uint32_t *bits, uint32_t k, uint32_t *cellcnt,
                                 Point *vox, uint8_t *cubeidx, uint32_t cap) {
    if (emit) hipLaunchKernelGGL(k_dense_classify<true>, dim3(grid), dim3(256), 0, st, bits, k, cellcnt, vox, cubeidx, cap);
    else hipLaunchKernelGGL(k_dense_classify<false>, dim3(grid), dim3(256), 0, st, bits, k, cellcnt, vox, cubeidx, cap);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Self-check of the lean exact arithmetic (sdf.hpp: sqrt_rn_normal / sqrt_rn / inv_length_rn / third_rn) against
// hipcc's own correctly rounded expansions, for EVERY input of each function's domain: 2^32 bit patterns per check,
// about a second in all.  The strict build's bit-exactness rests on "one Markstein step on this chip's v_rsq_f32 /
// v_rcp_f32 rounds correctly"; that is a property of the silicon, so the library can prove it on the device it runs on.
//   0  sqrt_rn_normal(x) == sqrtf(x) for x in [2^-95, FLT_MAX]; for 0 <= x < 2^-95 the result is finite and <= 2^-40
//   1  sqrt_rn(x) == sqrtf(x) for every bit pattern (NaNs compare as NaN)
//   2  inv_length_rn(d) == 1.0f / sqrtf(d) for every bit pattern
//   3  third_rn(a) == a / 3.0f for every finite a with a normal quotient, and for +-0 by value (+-inf gives NaN: the centroid of
//      a non-finite triangle, whose orientation test is "no flip" either way)
//   4  sqrt_rn_normal2({x, y}) == {sqrt_rn_normal(x), sqrt_rn_normal(y)} (y = a second walk through the same range)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool same_f32(float a, float b) {
    return __builtin_bit_cast(uint32_t, a) == __builtin_bit_cast(uint32_t, b) || (a != a && b != b);
}
template <int WHICH>
__global__ __launch_bounds__(256) void k_selfcheck(unsigned long long *__restrict__ bad) {
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
    unsigned long long mine = 0;
    for (uint64_t b = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; b < (1ull << 32); b += stride) {
        const float x = __builtin_bit_cast(float, (uint32_t) b);
        if (WHICH == 0) {
            if ((uint32_t) b >= 0x7f800000u) continue;                       // negative, inf, NaN: outside the domain
            const float r = sqrt_rn_normal(x);
            if ((uint32_t) b >= 0x10000000u) { if (!same_f32(r, sqrtf(x))) mine++; }
            else if (!(r >= 0.0f && r <= 9.094947017729282e-13f)) mine++;
        } else if (WHICH == 1) {
            if (!same_f32(sqrt_rn(x), sqrtf(x))) mine++;
        } else if (WHICH == 2) {
            if (!same_f32(inv_length_rn(x), 1.0f / sqrtf(x))) mine++;
        } else if (WHICH == 3) {
            const float q = x / 3.0f, t = third_rn(x);
            if (x == 0.0f) { if (t != 0.0f) mine++; }
            else if (fabsf(x) >= 4.70197740328915e-38f && fabsf(x) < __builtin_inff()) { if (!same_f32(t, q)) mine++; }   // 2^-124 <= |a| < inf
            else if (x != x && t == t) mine++;                                                                       // NaN stays NaN
        } else {
#if defined(SDM_HAVE_PAIRS)
            if ((uint32_t) b >= 0x7f800000u) continue;
            const float y = __builtin_bit_cast(float, (uint32_t) ((b * 2654435761ull) % 0x7f800000ull));
            const v2f r = sqrt_rn_normal2(mk2(x, y));
            if (!same_f32(r.x, sqrt_rn_normal(x)) || !same_f32(r.y, sqrt_rn_normal(y))) mine++;
#endif
        }
    }
    if (mine) atomicAdd(bad, mine);
}

hipError_t launch_selfcheck(hipStream_t st, int which, unsigned long long *bad) {
    const dim3 g(256 * 8), b(256);
    switch (which) {
    case 0: hipLaunchKernelGGL(k_selfcheck<0>, g, b, 0, st, bad); break;
    case 1: hipLaunchKernelGGL(k_selfcheck<1>, g, b, 0, st, bad); break;
    case 2: hipLaunchKernelGGL(k_selfcheck<2>, g, b, 0, st, bad); break;
    case 3: hipLaunchKernelGGL(k_selfcheck<3>, g, b, 0, st, bad); break;
    default: hipLaunchKernelGGL(k_selfcheck<4>, g, b, 0, st, bad); break;
    }
    return hipGetLastError();
}

hipError_t launch_pack_record(hipStream_t st, uint32_t *dst, const uint32_t *count, const uint32_t *status, uint32_t v0, uint32_t v1, uint32_t v2,
                              uint32_t v3) {
    hipLaunchKernelGGL(k_pack_record, dim3(1), dim3(1), 0, st, dst, count, status, v0, v1, v2, v3);
    return hipGetLastError();
}

hipError_t launch_dist_record(hipStream_t st, uint32_t *dst, const uint32_t *count, const uint32_t *status, uint32_t host_status, uint32_t cap_prev,
                              uint32_t inject, uint32_t sig) {
    hipLaunchKernelGGL(k_dist_record, dim3(1), dim3(1), 0, st, dst, count, status, host_status, cap_prev, inject, sig);
    return hipGetLastError();
}

hipError_t launch_dist_publish(hipStream_t st, const uint32_t *nups, const uint32_t *counts, uint32_t nranks, uint32_t seq, uint32_t *pub) {
    hipLaunchKernelGGL(k_dist_publish, dim3(1), dim3(256), 0, st, nups, counts, nranks, seq, pub);
    return hipGetLastError();
}

hipError_t launch_set_u32(hipStream_t st, uint32_t *p, uint32_t v) {
    hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(1), 0, st, p, v);
    return hipGetLastError();
}


}  // namespace sdm
