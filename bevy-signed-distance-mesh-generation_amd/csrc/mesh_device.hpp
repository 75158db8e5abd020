// mesh_device.hpp -- per-voxel device routines shared by libsdmesh.so (kernels.hip) and the
// reference-ABI code object (compute_mesh_generation.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sdmesh/bindings.h"
#include "mc_tables.hpp"
#include "sdf.hpp"

#if !defined(SDM_FAST_MATH)
#pragma clang fp contract(off)       // strict build: no FMA contraction (DESIGN.md "Numerics")
#endif

namespace sdm {

// corner c of a cube: +x iff c%4 in {1,2}, +y iff c%4 >= 2, +z iff c >= 4
// (compute_mesh_generation.cu:80-82)
__device__ __forceinline__ int cox(int c) { return ((c & 3) == 1 || (c & 3) == 2) ? 1 : 0; }
__device__ __forceinline__ int coy(int c) { return (c & 3) >= 2 ? 1 : 0; }
__device__ __forceinline__ int coz(int c) { return c >= 4 ? 1 : 0; }

// A2: the 27 lattice samples a parent's 8 children share -> 8-bit border mask, bit i*4 + j*2 + k
// (compute_mesh_generation.cu:30-60; the early-out of :36-49 does not change the result)
template <int SCENE>
__device__ __forceinline__ uint32_t refine_child_mask(const Point b, const Point cs, const SceneParams &sp) {
    uint32_t m = 0;
    uint32_t bits = 0;   // bit ix*9 + iy*3 + iz = (sd <= 0) at (b + (ix,iy,iz) * cs)
#pragma unroll 1
    for (int ix = 0; ix < 3; ix++) {
        const float x = b.x + (float) ix * cs.x;
#pragma unroll 1
        for (int iy = 0; iy < 3; iy++) {
            const float y = b.y + (float) iy * cs.y;
#pragma unroll
            for (int iz = 0; iz < 3; iz++) {
                const float z = b.z + (float) iz * cs.z;
                const uint32_t in = sdf<SCENE>(x, y, z, sp) <= 0.0f ? 1u : 0u;
                bits |= in << (ix * 9 + iy * 3 + iz);
            }
        }
    }
    // child (i,j,k) is a border voxel iff its 8 corner signs are not all equal
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const int i = c >> 2, j = (c >> 1) & 1, k = c & 1;
        const uint32_t cnt = __popc(bits & (0x361Bu << (i * 9 + j * 3 + k)));
        if (cnt != 0 && cnt != 8) m |= 1u << c;
    }
    return m;
}

__device__ __forceinline__ void nan_pad(Triangle &t) {
    const float qn = __builtin_nanf("");
    t.vertices[0].position.x = qn; t.vertices[0].position.y = qn; t.vertices[0].position.z = qn;
    t.vertices[0].normal.x = qn; t.vertices[0].normal.y = qn; t.vertices[0].normal.z = qn;
#pragma unroll
    for (int q = 1; q < 3; q++) {
        t.vertices[q].position.x = 0.0f; t.vertices[q].position.y = 0.0f; t.vertices[q].position.z = 0.0f;
        t.vertices[q].normal.x = 0.0f; t.vertices[q].normal.y = 0.0f; t.vertices[q].normal.z = 0.0f;
    }
}

// A7 direct path: per-triangle-corner projection exactly as the reference kernel does it
// (compute_mesh_generation.cu:64-120), writing the voxel's 5 raw slots.
template <int SCENE>
__device__ __forceinline__ void voxel_triangles_direct(const Point b, const Point vs, const SceneParams &sp, Triangle *out5) {
    f3 cp[8];
    uint32_t ci = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        cp[c] = mk(b.x + (cox(c) ? vs.x : 0.0f), b.y + (coy(c) ? vs.y : 0.0f), b.z + (coz(c) ? vs.z : 0.0f));
    }
#pragma unroll 1
    for (int c = 0; c < 8; c++) {
        const float x = b.x + (cox(c) ? vs.x : 0.0f), y = b.y + (coy(c) ? vs.y : 0.0f), z = b.z + (coz(c) ? vs.z : 0.0f);
        if (sdf<SCENE>(x, y, z, sp) <= 0.0f) ci |= 1u << c;
    }
    uint64_t row = MC_TRI_PACKED[ci];
    uint32_t nt = 0;
#pragma unroll 1
    while ((row & 0xF) != 0xF) {
        f3 pv[3], nv[3];
#pragma unroll 1
        for (int q = 0; q < 3; q++) {
            const int e = (int) (row & 0xF);
            row >>= 4;
            const int c0 = MC_EDGE_CORNERS[e] & 0xF, c1 = MC_EDGE_CORNERS[e] >> 4;
            f3 a = mk(0, 0, 0), bb = mk(0, 0, 0);
#pragma unroll
            for (int c = 0; c < 8; c++) {   // select without dynamic register indexing
                if (c == c0) a = cp[c];
                if (c == c1) bb = cp[c];
            }
            const f3 mid = mk(a.x * 0.5f + bb.x * 0.5f, a.y * 0.5f + bb.y * 0.5f, a.z * 0.5f + bb.z * 0.5f);
            pv[q] = closest_surface_point<SCENE>(mid, sp);
            nv[q] = empirical_normal<SCENE>(pv[q], sp);
        }
        const bool flip = change_orientation<SCENE>(pv[0], pv[1], pv[2], sp);
        Triangle t;
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int s = flip ? 2 - q : q;
            t.vertices[q].position.x = pv[s].x; t.vertices[q].position.y = pv[s].y; t.vertices[q].position.z = pv[s].z;
            t.vertices[q].normal.x = nv[s].x; t.vertices[q].normal.y = nv[s].y; t.vertices[q].normal.z = nv[s].z;
        }
        out5[nt++] = t;
    }
    Triangle pad;
    nan_pad(pad);
    for (; nt < 5; nt++) out5[nt] = pad;
}

}  // namespace sdm
