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
#include "mc_tables.hpp"
#include "mesh_device.hpp"
#include "sdf.hpp"

#pragma clang fp contract(off)

namespace sdm {

static constexpr uint32_t TILE = 256;
static constexpr uint64_t KEY_EMPTY = ~0ull;
static constexpr uint32_t U32_NONE = 0xffffffffu;

// ------------------------------------------------------------------------------------------
// wave / block primitives (wave64)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    const uint32_t lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= (uint32_t) d) v += t;
    }
    return v;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// Exclusive scan over the workgroup (blockDim.x multiple of 64, <= 1024).  `lds` needs 17 words.
// All threads must call.  Returns the exclusive prefix of `v`; `total` = workgroup sum.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t &total, uint32_t *lds) {
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint32_t incl = wave_incl_scan(v);
    __syncthreads();                       // protect lds from the previous use
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        uint32_t w = lane < nw ? lds[lane] : 0;
        uint32_t wi = wave_incl_scan(w);
        if (lane < nw) lds[lane] = wi - w;
        if (lane == 63) lds[16] = wi;
    }
    __syncthreads();
    total = lds[16];
    return lds[wave] + incl - v;
}

__device__ __forceinline__ uint64_t mix64(uint64_t h) {
    h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull; h ^= h >> 33;
    return h;
}

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

__device__ __forceinline__ uint64_t edge_key(int mx, int my, int mz, int e) {
    const int c0 = MC_EDGE_CORNERS[e] & 0xF, c1 = MC_EDGE_CORNERS[e] >> 4;
    const uint64_t kx = (uint64_t) (2 * mx + cox(c0) + cox(c1));
    const uint64_t ky = (uint64_t) (2 * my + coy(c0) + coy(c1));
    const uint64_t kz = (uint64_t) (2 * mz + coz(c0) + coz(c1));
    return kx | (ky << 21) | (kz << 42);
}

// open-addressing tables have an arbitrary (non power-of-two) size: home slot by fast range
// reduction of the mixed key, linear probing with wrap-around
__device__ __forceinline__ uint32_t hash_home(uint64_t key, uint32_t hsize) {
    return (uint32_t) (((mix64(key) >> 32) * (uint64_t) hsize) >> 32);
}
__device__ __forceinline__ uint32_t hash_next(uint32_t s, uint32_t hsize) { return s + 1 == hsize ? 0u : s + 1; }

__device__ __forceinline__ uint32_t hash_find(const uint64_t *__restrict__ keys, uint32_t hsize, uint64_t key) {
    uint32_t s = hash_home(key, hsize);
    for (uint32_t probe = 0; probe < hsize; probe++) {
        uint64_t k = keys[s];
        if (k == key) return s;
        if (k == KEY_EMPTY) return U32_NONE;
        s = hash_next(s, hsize);
    }
    return U32_NONE;
}

// ------------------------------------------------------------------------------------------
// A2 refine, pass 1: 27 shared lattice samples per parent -> 8-bit child mask + per-tile count
// ------------------------------------------------------------------------------------------
template <int SCENE>
__global__ __launch_bounds__(256) void k_refine_mask(const Point *__restrict__ vox, const uint32_t *__restrict__ n_ptr,
                                                      Point cs /* child size = parent size / 2 */, SceneParams sp,
                                                      uint8_t *__restrict__ mask, uint32_t *__restrict__ tilecnt) {
    __shared__ uint32_t s_cnt;
    const uint32_t n = *n_ptr;
    const uint32_t ntiles = (n + TILE - 1) / TILE;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        const uint32_t p = tile * TILE + threadIdx.x;
        uint32_t m = 0;
        if (p < n) {
            const Point b = vox[p];
            m = refine_child_mask<SCENE>(b, cs, sp);
            mask[p] = (uint8_t) m;
        }
        const uint32_t ws = wave_sum(__popc(m));
        if (lane_id() == 0 && ws) atomicAdd(&s_cnt, ws);
        __syncthreads();
        if (threadIdx.x == 0) tilecnt[tile] = s_cnt;
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

// A2 refine, pass 2: ordered emit of the border children (slot order id*8 + i*4 + j*2 + k)
__global__ __launch_bounds__(256) void k_refine_emit(const Point *__restrict__ vox, const uint32_t *__restrict__ n_ptr,
                                                      Point cs, const uint8_t *__restrict__ mask,
                                                      const uint32_t *__restrict__ tileoff, Point *__restrict__ out,
                                                      uint32_t cap) {
    __shared__ uint32_t lds[17];
    const uint32_t n = *n_ptr;
    const uint32_t ntiles = (n + TILE - 1) / TILE;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t p = tile * TILE + threadIdx.x;
        const uint32_t m = p < n ? mask[p] : 0u;
        uint32_t total;
        uint32_t o = tileoff[tile] + block_excl_scan(__popc(m), total, lds);
        if (m) {
            const Point b = vox[p];
#pragma unroll
            for (int c = 0; c < 8; c++) {
                if (m & (1u << c)) {
                    if (o < cap) {
                        Point q;
                        q.x = b.x + (float) (c >> 2) * cs.x;
                        q.y = b.y + (float) ((c >> 1) & 1) * cs.y;
                        q.z = b.z + (float) (c & 1) * cs.z;
                        out[o] = q;
                    }
                    o++;
                }
            }
        }
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
// A4/A7 mesh stage 1: cube index per voxel, triangle count per tile, unique-edge hash insert
// ------------------------------------------------------------------------------------------
template <int SCENE>
__global__ __launch_bounds__(256) void k_classify(const Point *__restrict__ vox, const uint32_t *__restrict__ n_ptr,
                                                   Point vs, SceneParams sp, uint8_t *__restrict__ cubeidx,
                                                   uint32_t *__restrict__ tilecnt, uint64_t *__restrict__ hkeys,
                                                   uint32_t *__restrict__ hids, uint32_t hsize,
                                                   uint64_t *__restrict__ ukeys, uint32_t ucap,
                                                   uint32_t *__restrict__ n_unique, uint32_t *__restrict__ status) {
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
            uint32_t em = MC_EDGE_MASK[ci];
            if (em) {
                int mx, my, mz;
                const bool okx = lattice_coord(b.x, vs.x, mx), oky = lattice_coord(b.y, vs.y, my), okz = lattice_coord(b.z, vs.z, mz);
                const bool ok = okx && oky && okz;
                if (!ok) {
                    atomicOr(&status[ST_NOT_LATTICE], 1u);
                } else {
                    while (em) {
                        const int e = __ffs(em) - 1;
                        em &= em - 1;
                        const uint64_t key = edge_key(mx, my, mz, e);
                        uint32_t s = hash_home(key, hsize);
                        bool done = false;
                        for (uint32_t probe = 0; probe < hsize && !done; probe++) {
                            const uint64_t prev = atomicCAS((unsigned long long *) &hkeys[s], (unsigned long long) KEY_EMPTY,
                                                            (unsigned long long) key);
                            if (prev == KEY_EMPTY) {
                                const uint32_t id = atomicAdd(n_unique, 1u);
                                if (id < ucap) ukeys[id] = key; else atomicOr(&status[ST_OVERFLOW], 2u);
                                hids[s] = id;
                                done = true;
                            } else if (prev == key) {
                                done = true;
                            } else {
                                s = hash_next(s, hsize);
                            }
                        }
                        if (!done) atomicOr(&status[ST_OVERFLOW], 4u);
                    }
                }
            }
        }
        const uint32_t ws = wave_sum(ntri);
        if (lane_id() == 0 && ws) atomicAdd(&s_cnt, ws);
        __syncthreads();
        if (threadIdx.x == 0) tilecnt[tile] = s_cnt;
        __syncthreads();
    }
}

// mesh stage 1b: per-voxel triangle offset + (voxel, slot) list in emission order
__global__ __launch_bounds__(256) void k_emit_trilist(const uint8_t *__restrict__ cubeidx, const uint32_t *__restrict__ n_ptr,
                                                       const uint32_t *__restrict__ tileoff, uint32_t *__restrict__ trioff,
                                                       uint32_t *__restrict__ trilist, uint32_t cap) {
    __shared__ uint32_t lds[17];
    const uint32_t n = *n_ptr;
    const uint32_t ntiles = (n + TILE - 1) / TILE;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t v = tile * TILE + threadIdx.x;
        const uint32_t ntri = v < n ? MC_TRI_COUNT[cubeidx[v]] : 0u;
        uint32_t total;
        const uint32_t o = tileoff[tile] + block_excl_scan(ntri, total, lds);
        if (v < n) trioff[v] = o;
        for (uint32_t j = 0; j < ntri; j++)
            if (o + j < cap) trilist[o + j] = (v << 3) | j;
    }
}

// ------------------------------------------------------------------------------------------
// A5/A6 mesh stage 2: one Newton projection + vertex normal per UNIQUE edge, lane refill
// ------------------------------------------------------------------------------------------
template <int SCENE>
__global__ __launch_bounds__(256) void k_project(const uint64_t *__restrict__ ukeys, const uint32_t *__restrict__ n_unique_ptr,
                                                  uint32_t ucap, uint32_t min_chunk, Point vs, SceneParams sp,
                                                  Vertex *__restrict__ uverts, uint32_t *__restrict__ status) {
    uint32_t U = *n_unique_ptr;
    if (U > ucap) U = ucap;
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    uint32_t chunk = (U + nwaves - 1) / nwaves;
    if (chunk < min_chunk) chunk = min_chunk;                         // keep the refill queue deep
    const uint64_t q0 = (uint64_t) wave * chunk;
    uint32_t qhead = q0 < U ? (uint32_t) q0 : U;                      // wave-uniform work queue
    const uint32_t qend = q0 + chunk < U ? (uint32_t) (q0 + chunk) : U;
    const uint32_t lane = lane_id();
    const uint64_t lt = (1ull << lane) - 1ull;

    bool active = false, finalize = false;
    uint32_t id = 0;
    int it = 0;
    f3 g = mk(0.0f, 0.0f, 0.0f);
    for (;;) {
        const uint64_t need = __ballot(!active);
        const uint32_t avail = qend - qhead;
        if (need != 0ull && avail != 0u) {
            const uint32_t rank = __popcll(need & lt);
            if (!active && rank < avail) {
                id = qhead + rank;
                const uint64_t key = ukeys[id];
                const int kx = (int) (key & 0x1FFFFF), ky = (int) ((key >> 21) & 0x1FFFFF), kz = (int) ((key >> 42) & 0x1FFFFF);
                // midpoint = a*0.5 + b*0.5 of the two lattice corners (marching_cubes.cu:13-16)
                g.x = lat(kx >> 1, vs.x) * 0.5f + lat((kx + 1) >> 1, vs.x) * 0.5f;
                g.y = lat(ky >> 1, vs.y) * 0.5f + lat((ky + 1) >> 1, vs.y) * 0.5f;
                g.z = lat(kz >> 1, vs.z) * 0.5f + lat((kz + 1) >> 1, vs.z) * 0.5f;
                active = true; finalize = false; it = 0;
            }
            const uint32_t want = __popcll(need);
            qhead += want < avail ? want : avail;
        }
        if (__ballot(active) == 0ull) break;      // queue drained and every lane idle
        if (active) {
            const f3 nn = empirical_normal<SCENE>(g, sp);
            if (!finalize) {
                const float sd = sdf<SCENE>(g.x, g.y, g.z, sp);
                g = sub3(g, scale3(nn, sd));
                it++;
                if (fabsf(sd) <= 0.00001f) finalize = true;
                else if (it >= SDM_MAX_NEWTON) { finalize = true; atomicOr(&status[ST_NEWTON_CAPPED], 1u); }
            } else {
                Vertex o;
                o.position.x = g.x; o.position.y = g.y; o.position.z = g.z;
                o.normal.x = nn.x; o.normal.y = nn.y; o.normal.z = nn.z;
                uverts[id] = o;
                active = false;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// A7 mesh stage 3: per triangle -- gather its 3 unique-edge ids, orientation test, first-seen
// ordinal per unique edge (atomicMin: order independent)
// ------------------------------------------------------------------------------------------
template <int SCENE>
__global__ __launch_bounds__(256) void k_tri_orient(const Point *__restrict__ vox, Point vs, const uint8_t *__restrict__ cubeidx,
                                                     const uint32_t *__restrict__ trilist, const uint32_t *__restrict__ ntri_ptr,
                                                     const uint64_t *__restrict__ hkeys, const uint32_t *__restrict__ hids,
                                                     uint32_t hsize, const Vertex *__restrict__ uverts, SceneParams sp,
                                                     uint32_t *__restrict__ tri_e, uint32_t *__restrict__ first_ord,
                                                     uint32_t *__restrict__ status) {
    const uint32_t T = *ntri_ptr;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < T; t += gridDim.x * blockDim.x) {
        const uint32_t tv = trilist[t], v = tv >> 3, j = tv & 7u;
        const uint32_t ci = cubeidx[v];
        const Point b = vox[v];
        int mx, my, mz;
        lattice_coord(b.x, vs.x, mx); lattice_coord(b.y, vs.y, my); lattice_coord(b.z, vs.z, mz);
        const uint32_t row = (uint32_t) (MC_TRI_PACKED[ci] >> (12 * j)) & 0xFFFu;
        uint32_t e[3];
        f3 pv[3];
        bool ok = true;
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const uint32_t s = hash_find(hkeys, hsize, edge_key(mx, my, mz, (int) ((row >> (4 * q)) & 0xFu)));
            ok = ok && (s != U32_NONE);
            e[q] = ok ? hids[s] : 0u;
            const Vertex u = uverts[e[q]];
            pv[q] = mk(u.position.x, u.position.y, u.position.z);
        }
        if (!ok) atomicOr(&status[ST_INTERNAL], 1u);
        const bool flip = change_orientation<SCENE>(pv[0], pv[1], pv[2], sp);
        const uint32_t e0 = flip ? e[2] : e[0], e2 = flip ? e[0] : e[2];
        tri_e[3 * t + 0] = e0; tri_e[3 * t + 1] = e[1]; tri_e[3 * t + 2] = e2;
        atomicMin(&first_ord[e0], 3 * t + 0);
        atomicMin(&first_ord[e[1]], 3 * t + 1);
        atomicMin(&first_ord[e2], 3 * t + 2);
    }
}

// first-seen ordinals for an arbitrary (tri_e) list -- used by the generic weld entry point
__global__ __launch_bounds__(256) void k_first_ord(const uint32_t *__restrict__ tri_e, const uint32_t *__restrict__ ntri_ptr,
                                                    uint32_t *__restrict__ first_ord) {
    const uint32_t n3 = 3u * *ntri_ptr;
    for (uint32_t o = blockIdx.x * blockDim.x + threadIdx.x; o < n3; o += gridDim.x * blockDim.x)
        atomicMin(&first_ord[tri_e[o]], o);
}

// ------------------------------------------------------------------------------------------
// A8 weld (src/cuda/mod.rs:263-296) on the unique-vertex list.
//   key   = 3 x 21-bit biased round(x * 1e5)  (covers +-10.48 units; bounding cube is +-2.5)
//   value = smallest first-seen ordinal among the unique vertices that share the key
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_weld_insert(const Vertex *__restrict__ uverts, const uint32_t *__restrict__ nu_ptr,
                                                      uint32_t ucap, const uint32_t *__restrict__ first_ord,
                                                      uint64_t *__restrict__ wkeys, uint32_t *__restrict__ wmin, uint32_t wsize,
                                                      uint32_t *__restrict__ vslot, uint32_t *__restrict__ status) {
    uint32_t U = *nu_ptr;
    if (U > ucap) U = ucap;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < U; e += gridDim.x * blockDim.x) {
        const uint32_t fo = first_ord[e];
        uint32_t slot = U32_NONE;
        if (fo != U32_NONE) {
            const Point p = uverts[e].position;
            const long long qx = weld_quant(p.x) + 1048576LL, qy = weld_quant(p.y) + 1048576LL, qz = weld_quant(p.z) + 1048576LL;
            if (((unsigned long long) qx | (unsigned long long) qy | (unsigned long long) qz) >> 21) {
                atomicOr(&status[ST_WELD_RANGE], 1u);
            } else {
                const uint64_t key = (uint64_t) qx | ((uint64_t) qy << 21) | ((uint64_t) qz << 42);
                uint32_t s = hash_home(key, wsize);
                for (uint32_t probe = 0; probe < wsize; probe++) {
                    const uint64_t prev = atomicCAS((unsigned long long *) &wkeys[s], (unsigned long long) KEY_EMPTY,
                                                    (unsigned long long) key);
                    if (prev == KEY_EMPTY || prev == key) { slot = s; break; }
                    s = hash_next(s, wsize);
                }
                if (slot == U32_NONE) atomicOr(&status[ST_OVERFLOW], 8u);
                else atomicMin(&wmin[slot], fo);
            }
        }
        vslot[e] = slot;
    }
}

__global__ __launch_bounds__(256) void k_weld_mark(const uint32_t *__restrict__ nu_ptr, uint32_t ucap,
                                                    const uint32_t *__restrict__ first_ord, const uint32_t *__restrict__ wmin,
                                                    const uint32_t *__restrict__ vslot, uint32_t *__restrict__ bitmap) {
    uint32_t U = *nu_ptr;
    if (U > ucap) U = ucap;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < U; e += gridDim.x * blockDim.x) {
        const uint32_t s = vslot[e];
        if (s == U32_NONE) continue;
        const uint32_t fo = first_ord[e];
        if (wmin[s] == fo) atomicOr(&bitmap[fo >> 5], 1u << (fo & 31u));
    }
}

// exclusive scan of popcounts over the first-seen bitmap (3T bits); total -> n_verts
__global__ __launch_bounds__(1024) void k_scan_bitmap(const uint32_t *__restrict__ bitmap, const uint32_t *__restrict__ ntri_ptr,
                                                       uint32_t *__restrict__ wordpre, uint32_t *__restrict__ nverts_out) {
    __shared__ uint32_t lds[17];
    const uint32_t nwords = (3u * *ntri_ptr + 31u) >> 5;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nwords; base += blockDim.x) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nwords ? __popc(bitmap[i]) : 0u;
        uint32_t total;
        const uint32_t ex = block_excl_scan(v, total, lds);
        if (i < nwords) wordpre[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) *nverts_out = carry;
}

__global__ __launch_bounds__(256) void k_weld_assign(const Vertex *__restrict__ uverts, const uint32_t *__restrict__ nu_ptr,
                                                      uint32_t ucap, const uint32_t *__restrict__ first_ord,
                                                      const uint32_t *__restrict__ wmin, const uint32_t *__restrict__ vslot,
                                                      const uint32_t *__restrict__ bitmap, const uint32_t *__restrict__ wordpre,
                                                      uint32_t *__restrict__ wvidx, Vertex *__restrict__ vertices, uint32_t vcap,
                                                      uint32_t *__restrict__ status) {
    uint32_t U = *nu_ptr;
    if (U > ucap) U = ucap;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < U; e += gridDim.x * blockDim.x) {
        const uint32_t s = vslot[e];
        if (s == U32_NONE) continue;
        const uint32_t fo = first_ord[e];
        if (wmin[s] != fo) continue;
        const uint32_t vi = wordpre[fo >> 5] + __popc(bitmap[fo >> 5] & ((1u << (fo & 31u)) - 1u));
        wvidx[s] = vi;
        if (vi < vcap) vertices[vi] = uverts[e]; else atomicOr(&status[ST_OVERFLOW], 16u);
    }
}

__global__ __launch_bounds__(256) void k_weld_index(const uint32_t *__restrict__ tri_e, const uint32_t *__restrict__ ntri_ptr,
                                                     const uint32_t *__restrict__ vslot, const uint32_t *__restrict__ wvidx,
                                                     uint32_t *__restrict__ indices) {
    const uint32_t n3 = 3u * *ntri_ptr;
    for (uint32_t o = blockIdx.x * blockDim.x + threadIdx.x; o < n3; o += gridDim.x * blockDim.x) {
        const uint32_t s = vslot[tri_e[o]];
        indices[o] = s != U32_NONE ? wvidx[s] : U32_NONE;
    }
}

// ------------------------------------------------------------------------------------------
// layout converters
// ------------------------------------------------------------------------------------------
// compact triangle list (72 B each, emission order) from (uverts, tri_e)
__global__ __launch_bounds__(256) void k_expand_compact(const Vertex *__restrict__ uverts, const uint32_t *__restrict__ tri_e,
                                                         const uint32_t *__restrict__ ntri_ptr, Triangle *__restrict__ out,
                                                         uint32_t cap) {
    uint32_t n3 = 3u * *ntri_ptr;
    if (n3 > 3u * cap) n3 = 3u * cap;
    Vertex *o = reinterpret_cast<Vertex *>(out);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n3; i += gridDim.x * blockDim.x) o[i] = uverts[tri_e[i]];
}

// reference raw layout: 5 slots per voxel, NaN/zero padded (compute_mesh_generation.cu:116-118)
__global__ __launch_bounds__(256) void k_expand_raw(const Vertex *__restrict__ uverts, const uint32_t *__restrict__ tri_e,
                                                     const uint8_t *__restrict__ cubeidx, const uint32_t *__restrict__ trioff,
                                                     uint32_t n, Triangle *__restrict__ out) {
    const uint32_t total = n * 5u;
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < total; s += gridDim.x * blockDim.x) {
        const uint32_t v = s / 5u, j = s - v * 5u;
        Triangle t;
        if (j < MC_TRI_COUNT[cubeidx[v]]) {
            const uint32_t o = 3u * (trioff[v] + j);
            t.vertices[0] = uverts[tri_e[o]]; t.vertices[1] = uverts[tri_e[o + 1]]; t.vertices[2] = uverts[tri_e[o + 2]];
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
__global__ __launch_bounds__(128) void k_triangles_direct(const Point *__restrict__ vox, uint32_t n, Point vs, SceneParams sp,
                                                           Triangle *__restrict__ out) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n) voxel_triangles_direct<SCENE>(vox[v], vs, sp, out + 5u * (size_t) v);
}

// ------------------------------------------------------------------------------------------
// A1 level-0 field on the device (src/cuda/mod.rs:105-122): x outer, y, z inner; optional
// z-slab [z0, z1) for the multi-GPU partition (same order, slab cells only)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_create_field(Point *__restrict__ vox, uint32_t z0, uint32_t z1, uint32_t *__restrict__ n_out) {
    const float SIZE = MESH_GENERATION_BB_SIZE / (float) MESH_GENERATION_INIT_FACTOR;
    const uint32_t F = MESH_GENERATION_INIT_FACTOR, zs = z1 - z0, total = F * F * zs;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t z = z0 + i % zs, y = (i / zs) % F, x = i / (zs * F);
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

__global__ void k_set_u32(uint32_t *p, uint32_t v) { *p = v; }
__global__ void k_mul3_u32(const uint32_t *a, uint32_t *out) { *out = 3u * *a; }

// ------------------------------------------------------------------------------------------
// P1: dense SDF sample pass, (N+1)^3 f32, x outer / z inner, one float4 store per thread
// ------------------------------------------------------------------------------------------
template <int SCENE>
__global__ __launch_bounds__(256) void k_sample_dense(float *__restrict__ out, uint32_t N, uint64_t total, SceneParams sp) {
    const float h = 5.0f / (float) N;
    const uint32_t S = N + 1u;
    const uint64_t nquads = (total + 3ull) >> 2;
    for (uint64_t qd = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; qd < nquads; qd += (uint64_t) gridDim.x * blockDim.x) {
        const uint64_t i0 = qd << 2;
        uint32_t z = (uint32_t) (i0 % S);
        const uint64_t r = i0 / S;
        uint32_t y = (uint32_t) (r % S), x = (uint32_t) (r / S);
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            v[k] = sdf<SCENE>((float) x * h - 2.5f, (float) y * h - 2.5f, (float) z * h - 2.5f, sp);
            if (++z == S) { z = 0; if (++y == S) { y = 0; ++x; } }
        }
        if (i0 + 3 < total) {
            *reinterpret_cast<float4 *>(out + i0) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            for (int k = 0; k < 4 && i0 + k < total; k++) out[i0 + k] = v[k];
        }
    }
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
    default: return hipErrorInvalidValue;                                                         \
    }

hipError_t launch_refine_mask(hipStream_t st, int grid, int scene, const SceneParams &sp, const Point *vox,
                              const uint32_t *n_ptr, Point cs, uint8_t *mask, uint32_t *tilecnt) {
    SDM_DISPATCH(scene, hipLaunchKernelGGL(k_refine_mask<S>, dim3(grid), dim3(256), 0, st, vox, n_ptr, cs, sp, mask, tilecnt));
    return hipGetLastError();
}

hipError_t launch_scan_tiles(hipStream_t st, uint32_t *cnt, const uint32_t *n_ptr, uint32_t items_per_tile,
                             uint32_t *total_out, uint32_t cap, uint32_t *status) {
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, st, cnt, n_ptr, items_per_tile, total_out, cap, status);
    return hipGetLastError();
}

hipError_t launch_refine_emit(hipStream_t st, int grid, const Point *vox, const uint32_t *n_ptr, Point cs,
                              const uint8_t *mask, const uint32_t *tileoff, Point *out, uint32_t cap) {
    hipLaunchKernelGGL(k_refine_emit, dim3(grid), dim3(256), 0, st, vox, n_ptr, cs, mask, tileoff, out, cap);
    return hipGetLastError();
}

hipError_t launch_refine_expand_raw(hipStream_t st, int grid, const Point *vox, uint32_t n, Point cs, const uint8_t *mask,
                                    Point *out) {
    hipLaunchKernelGGL(k_refine_expand_raw, dim3(grid), dim3(256), 0, st, vox, n, cs, mask, out);
    return hipGetLastError();
}

hipError_t launch_classify(hipStream_t st, int grid, int scene, const SceneParams &sp, const Point *vox,
                           const uint32_t *n_ptr, Point vs, uint8_t *cubeidx, uint32_t *tilecnt, uint64_t *hkeys,
                           uint32_t *hids, uint32_t hsize, uint64_t *ukeys, uint32_t ucap, uint32_t *n_unique,
                           uint32_t *status) {
    SDM_DISPATCH(scene, hipLaunchKernelGGL(k_classify<S>, dim3(grid), dim3(256), 0, st, vox, n_ptr, vs, sp, cubeidx, tilecnt,
                                           hkeys, hids, hsize, ukeys, ucap, n_unique, status));
    return hipGetLastError();
}

hipError_t launch_emit_trilist(hipStream_t st, int grid, const uint8_t *cubeidx, const uint32_t *n_ptr,
                               const uint32_t *tileoff, uint32_t *trioff, uint32_t *trilist, uint32_t cap) {
    hipLaunchKernelGGL(k_emit_trilist, dim3(grid), dim3(256), 0, st, cubeidx, n_ptr, tileoff, trioff, trilist, cap);
    return hipGetLastError();
}

hipError_t launch_project(hipStream_t st, int grid, int scene, const SceneParams &sp, const uint64_t *ukeys,
                          const uint32_t *n_unique_ptr, uint32_t ucap, uint32_t min_chunk, Point vs, Vertex *uverts,
                          uint32_t *status) {
    SDM_DISPATCH(scene, hipLaunchKernelGGL(k_project<S>, dim3(grid), dim3(256), 0, st, ukeys, n_unique_ptr, ucap, min_chunk, vs, sp,
                                           uverts, status));
    return hipGetLastError();
}

hipError_t launch_tri_orient(hipStream_t st, int grid, int scene, const SceneParams &sp, const Point *vox, Point vs,
                             const uint8_t *cubeidx, const uint32_t *trilist, const uint32_t *ntri_ptr,
                             const uint64_t *hkeys, const uint32_t *hids, uint32_t hsize, const Vertex *uverts,
                             uint32_t *tri_e, uint32_t *first_ord, uint32_t *status) {
    SDM_DISPATCH(scene, hipLaunchKernelGGL(k_tri_orient<S>, dim3(grid), dim3(256), 0, st, vox, vs, cubeidx, trilist, ntri_ptr,
                                           hkeys, hids, hsize, uverts, sp, tri_e, first_ord, status));
    return hipGetLastError();
}

hipError_t launch_first_ord(hipStream_t st, int grid, const uint32_t *tri_e, const uint32_t *ntri_ptr, uint32_t *first_ord) {
    hipLaunchKernelGGL(k_first_ord, dim3(grid), dim3(256), 0, st, tri_e, ntri_ptr, first_ord);
    return hipGetLastError();
}

hipError_t launch_weld(hipStream_t st, int grid, const Vertex *uverts, const uint32_t *nu_ptr, uint32_t ucap,
                       const uint32_t *tri_e, const uint32_t *ntri_ptr, const uint32_t *first_ord, uint64_t *wkeys,
                       uint32_t *wmin, uint32_t wsize, uint32_t *vslot, uint32_t *bitmap, uint32_t *wordpre,
                       uint32_t *wvidx, Vertex *vertices, uint32_t vcap, uint32_t *nverts_out, uint32_t *indices,
                       uint32_t *status) {
    hipLaunchKernelGGL(k_weld_insert, dim3(grid), dim3(256), 0, st, uverts, nu_ptr, ucap, first_ord, wkeys, wmin, wsize, vslot,
                       status);
    hipLaunchKernelGGL(k_weld_mark, dim3(grid), dim3(256), 0, st, nu_ptr, ucap, first_ord, wmin, vslot, bitmap);
    hipLaunchKernelGGL(k_scan_bitmap, dim3(1), dim3(1024), 0, st, bitmap, ntri_ptr, wordpre, nverts_out);
    hipLaunchKernelGGL(k_weld_assign, dim3(grid), dim3(256), 0, st, uverts, nu_ptr, ucap, first_ord, wmin, vslot, bitmap, wordpre,
                       wvidx, vertices, vcap, status);
    hipLaunchKernelGGL(k_weld_index, dim3(grid), dim3(256), 0, st, tri_e, ntri_ptr, vslot, wvidx, indices);
    return hipGetLastError();
}

hipError_t launch_expand_compact(hipStream_t st, int grid, const Vertex *uverts, const uint32_t *tri_e,
                                 const uint32_t *ntri_ptr, Triangle *out, uint32_t cap) {
    hipLaunchKernelGGL(k_expand_compact, dim3(grid), dim3(256), 0, st, uverts, tri_e, ntri_ptr, out, cap);
    return hipGetLastError();
}

hipError_t launch_expand_raw(hipStream_t st, int grid, const Vertex *uverts, const uint32_t *tri_e, const uint8_t *cubeidx,
                             const uint32_t *trioff, uint32_t n, Triangle *out) {
    hipLaunchKernelGGL(k_expand_raw, dim3(grid), dim3(256), 0, st, uverts, tri_e, cubeidx, trioff, n, out);
    return hipGetLastError();
}

hipError_t launch_triangles_direct(hipStream_t st, int scene, const SceneParams &sp, const Point *vox, uint32_t n, Point vs,
                                   Triangle *out) {
    const int grid = (int) ((n + 127u) / 128u);
    if (grid == 0) return hipSuccess;
    SDM_DISPATCH(scene, hipLaunchKernelGGL(k_triangles_direct<S>, dim3(grid), dim3(128), 0, st, vox, n, vs, sp, out));
    return hipGetLastError();
}

hipError_t launch_sample_dense(hipStream_t st, int grid, int scene, const SceneParams &sp, float *out, uint32_t N) {
    const uint64_t total = (uint64_t) (N + 1) * (N + 1) * (N + 1);
    SDM_DISPATCH(scene, hipLaunchKernelGGL(k_sample_dense<S>, dim3(grid), dim3(256), 0, st, out, N, total, sp));
    return hipGetLastError();
}

hipError_t launch_create_field(hipStream_t st, Point *vox, uint32_t z0, uint32_t z1, uint32_t *n_out) {
    hipLaunchKernelGGL(k_create_field, dim3(128), dim3(256), 0, st, vox, z0, z1, n_out);
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

hipError_t launch_set_u32(hipStream_t st, uint32_t *p, uint32_t v) {
    hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(1), 0, st, p, v);
    return hipGetLastError();
}

hipError_t launch_mul3_u32(hipStream_t st, const uint32_t *a, uint32_t *out) {
    hipLaunchKernelGGL(k_mul3_u32, dim3(1), dim3(1), 0, st, a, out);
    return hipGetLastError();
}

}  // namespace sdm
