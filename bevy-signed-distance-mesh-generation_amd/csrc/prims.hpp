// prims.hpp -- wave64 / workgroup primitives and the hash helpers shared by the kernel files (kernels.hip, zslab.hip)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sdmesh/bindings.h"
#include "sdf.hpp"

namespace sdm {

static constexpr uint32_t TILE = 256;
static constexpr uint64_t KEY_EMPTY = ~0ull;
static constexpr uint32_t U32_NONE = 0xffffffffu;

// ------------------------------------------------------------------------------------------
// wave / block primitives (wave64)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// Wave-wide inclusive prefix sum with DPP row shifts / row broadcasts (gfx9 data-parallel primitives: ~6 VALU operations)
// instead of six __shfl_up steps, each of which is a ds_bpermute_b32 round trip through the LDS crossbar.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add(uint32_t v) {
    // lanes whose DPP source is out of range (or whose row is masked off) contribute `old` = 0
    return v + (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, CTRL, ROW_MASK, 0xf, false);
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    v = dpp_add<0x111, 0xf>(v);      // row_shr:1
    v = dpp_add<0x112, 0xf>(v);      // row_shr:2
    v = dpp_add<0x114, 0xf>(v);      // row_shr:4
    v = dpp_add<0x118, 0xf>(v);      // row_shr:8   -> inclusive sums inside each row of 16 lanes
    v = dpp_add<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3
    return v;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    return (uint32_t) __builtin_amdgcn_readlane((int) wave_incl_scan(v), 63);
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

// Tile counts that a 256-thread workgroup prefixes for itself in LDS (api.hip: LOCAL_TILES).  A single-workgroup scan kernel
// between two per-tile kernels is a whole dependent launch (~4.7 us at 512^3: dispatch, two dependent loads, the scan, the
// stores) for ~1 000 numbers; every consumer workgroup re-deriving the prefix costs it one 8-load step instead.
static constexpr uint32_t LOCAL_TILES = SDM_LOCAL_TILES;

// exclusive prefix of n <= LOCAL_TILES counts (count i at base[i * stride]) -> s_pre[0 .. n]; returns the total.
// blockDim.x == 256, all threads must call; s_scan needs 17 words.
__device__ __forceinline__ uint32_t local_prefix(const uint32_t *__restrict__ base, uint32_t stride, uint32_t n,
                                                 uint32_t *__restrict__ s_pre, uint32_t *__restrict__ s_scan) {
    constexpr uint32_t PER = LOCAL_TILES / 256;
    uint32_t v[PER], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
        const uint32_t i = threadIdx.x * PER + k;
        v[k] = i < n ? base[(size_t) i * stride] : 0u;
        sum += v[k];
    }
    uint32_t total;
    uint32_t ex = block_excl_scan(sum, total, s_scan);
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
        const uint32_t i = threadIdx.x * PER + k;
        if (i < n) s_pre[i] = ex;
        ex += v[k];
    }
    if (threadIdx.x == 0) s_pre[n] = total;
    __syncthreads();
    return total;
}

// the tile lo in [0, n) with s_pre[lo] <= item < s_pre[lo + 1]   (item < s_pre[n])
__device__ __forceinline__ uint32_t local_find(const uint32_t *__restrict__ s_pre, uint32_t n, uint32_t item) {
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (s_pre[mid] <= item) lo = mid; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ uint64_t mix64(uint64_t h) {
    h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull; h ^= h >> 33;
    return h;
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

// packed 3 x 21-bit weld key of a position (src/cuda/mod.rs:268-273 restated on the 1e-5 quantum); false = outside +-10.48 units
__device__ __forceinline__ bool packed_weld_key(const Point p, uint64_t &key) {
    const long long qx = weld_quant(p.x) + 1048576LL, qy = weld_quant(p.y) + 1048576LL, qz = weld_quant(p.z) + 1048576LL;
    key = (uint64_t) qx | ((uint64_t) qy << 21) | ((uint64_t) qz << 42);
    return (((unsigned long long) qx | (unsigned long long) qy | (unsigned long long) qz) >> 21) == 0ull;
}

}  // namespace sdm
