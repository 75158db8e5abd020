// sdf.hpp -- scene signed-distance functions and the per-point numerics of the mesh path,
// written for gfx950 (also host-callable so the CPU test-suite can check the formulation).
//
// Semantics follow the reference bit for bit under strict IEEE f32 (no FMA contraction,
// correctly rounded div/sqrt):
//   sd_obj              cuda/modules/common.cu:222-226
//   sd_box_skeleton     cuda/includes/signed_distance.cu:93-113 (incl. the `% 2` bug at :101)
//   sd_line / sd_ray    signed_distance.cu:77-80 / :65-75
//   smooth_min          signed_distance.cu:20-23
//   sd_box              signed_distance.cu:86-91
//   empirical_normal    signed_distance.cu:179-202
//   closest_surface_point signed_distance.cu:227-240 (one Newton round = newton_round())
// GLM arithmetic as fixed in SURVEY.md 8(c): dot = (x+y)+z, length = sqrt(dot),
// normalize = v * (1/sqrt(dot)), min(a,b) = (b<a)?b:a, max(a,b) = (a<b)?b:a.
//
// This is NOT a transliteration.  sd_obj is restructured for the VALU:
//  * all 12 bars of the skeleton are axis-aligned with compile-time end points, so
//    dot(p - b0, dir) is exactly one coordinate difference, the clamp is a select, and the
//    perpendicular squared offsets are shared between bars (11 squares + 20 adds for 12 bars);
//  * min_i (sqrt(d2_i) - lw) == sqrt(min_i d2_i) - lw exactly (correctly rounded sqrt and the
//    subtraction are monotone non-decreasing), so one sqrt replaces twelve.
// Both rewrites are value-preserving for finite inputs; tests/test_host_numerics.py checks the
// result bit-for-bit against the oracle's literal 12-bar loop.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define SDM_HD __host__ __device__ __forceinline__
#define SDM_HDM __host__ __device__ __forceinline__
#else
#define SDM_HD static inline
#define SDM_HDM inline
#endif

#if !defined(SDM_FAST_MATH)
#pragma clang fp contract(off)       // strict build: no FMA contraction (DESIGN.md "Numerics")
#endif

namespace sdm {

enum Scene : int { SCENE_SDOBJ = 0, SCENE_SPHERE = 1, SCENE_CSG3 = 2, SCENE_ANIM = 3, SCENE_PROGRAM = 4, SCENE_BULB = 5, SCENE_COUNT = 6 };

// Runtime SDF program: a linear CSG chain  acc = combine_i(acc, primitive_i(p)).
// Primitives are the reference's (cuda/includes/signed_distance.cu:61-131: sd_ray in both forms, sd_line, sd_box, sd_box_skeleton,
// sd_simple_box, sd_simple_bounding_box) plus a sphere with a radius and a torus; OP_WRAP is the reference's domain repetition
// (signed_distance.cu:9-18) applied to the point every later instruction sees.  Layout == sdmesh_instr.  (sd_mandelbulb, :29-52, is a
// scene of its own, SCENE_BULB below: its acos / atan2 / pow / sin / cos / log are this file's own, see "elementary functions".)
enum SdfOp : uint32_t { OP_SPHERE = 1, OP_BOX = 2, OP_BOX_SKELETON = 3, OP_CAPSULE = 4, OP_TORUS_Y = 5, OP_RAY = 6, OP_SIMPLE_BOX = 7,
                        OP_SIMPLE_BOUNDING_BOX = 8, OP_WRAP = 9, OP_LAST = 9 };
enum SdfCombine : uint32_t { CB_SET = 0, CB_MIN = 1, CB_MAX = 2, CB_SMOOTH_MIN = 3, CB_SUBTRACT = 4 };
#define SDM_MAX_INSTR 16

struct SdfInstr {
    uint32_t op;
    uint32_t combine;
    float k;          // smooth-min radius
    float a[7];       // primitive arguments
};

struct SceneParams {
    float p[4];
    uint32_t ninstr;
    uint32_t pad[3];
    SdfInstr prog[SDM_MAX_INSTR];
};

struct f3 {
    float x, y, z;
};

SDM_HD f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
SDM_HD float gmin(float a, float b) { return (b < a) ? b : a; }      // glm::min / glm::max, exact semantics
SDM_HD float gmax(float a, float b) { return (a < b) ? b : a; }
// Single-instruction min/max/clamp (v_min_f32 / v_max_f32 / v_med3_f32).  They differ from glm::min/max only
// for NaN operands and in the sign of a zero result; they are used only where the operands are finite and a
// -0 cannot occur (sums of squares, distances minus positive constants), so the values are identical.
SDM_HD float fmin1(float a, float b) { return __builtin_fminf(a, b); }
SDM_HD float fmax1(float a, float b) { return __builtin_fmaxf(a, b); }
SDM_HD float clamp1(float t, float lo, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fmed3f(t, lo, hi);
#else
    return __builtin_fminf(__builtin_fmaxf(t, lo), hi);
#endif
}
SDM_HD float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
SDM_HD f3 sub3(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
SDM_HD f3 add3(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
SDM_HD f3 scale3(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
SDM_HD f3 cross3(f3 x, f3 y) {
    return mk(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}

// Correctly rounded square root / reciprocal / third on gfx950 WITHOUT the two-sided residual tests and the scaling hipcc
// expands sqrtf and `/` into.  One Markstein step on the hardware's 1-ulp estimate,
//     y = v_rsq_f32(x); g = x * y; h = y / 2; e = fma(-g, g, x); r = fma(e, h, g)
// is not correctly rounded for an arbitrary 1-ulp estimate, but it IS for this chip's v_rsq_f32: tools/ubench/exact_sqrt.hip
// and tests/test_parity_gpu.py::test_lean_exact_arithmetic_is_exact_for_every_input compare it with hipcc's sqrtf for EVERY float
// in [2^-100, FLT_MAX] (2^31 inputs, a second on the GPU): zero mismatches (profiles/r03_exact_sqrt.jsonl).  Likewise
// fma(fma(-b, y, 1), y, y) with y = v_rcp_f32(b) equals 1.0f / b for every normal b whose reciprocal is normal, and
// fma(fma(-3, q, a), c, q) with c = RN(1/3), q = a * c equals a / 3.0f for every a with a normal quotient.
// Cost per wave and SIMD (same micro-benchmark): 11 ns against 21 ns for round 2's v_sqrt_f32 + two-sided test.
//
// sqrt_rn_normal(x): x == 0, or x >= 2^-95, finite.  The input is nudged by 2^-120 (x + 2^-120 == x for every x >= 2^-95: the
// addend is far below half an ulp) so that x == 0 gives 2^-60 instead of 0 * inf = NaN; for 0 < x < 2^-95 the result is some
// value in [0, 2^-47].  The only callers are expressions  sqrt(x) - c  with |c| >= 0.1: there any root below 2^-30 is absorbed by
// the subtraction and the value equals the reference's bit for bit.  (NaN returns NaN; x = +inf, which no bounded point
// produces, would return NaN instead of +inf.)
#define SDM_SQRT_NUDGE 7.52316384526264e-37f       /* 2^-120 */
SDM_HD float sqrt_rn_normal(float x) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(SDM_FAST_MATH)
    return __builtin_amdgcn_sqrtf(x);          // fast build: the bare v_sqrt_f32 (1 ulp), as nvcc --use_fast_math takes sqrt.approx
#elif defined(__HIP_DEVICE_COMPILE__)
    const float xc = x + SDM_SQRT_NUDGE;
    const float y = __builtin_amdgcn_rsqf(xc);
    const float g = xc * y, h = 0.5f * y;
    const float e = __builtin_fmaf(-g, g, xc);
    return __builtin_fmaf(e, h, g);
#else
    return sqrtf(x);
#endif
}

// sqrtf(x) for ANY x: the lean form inside [2^-90, 2^90] (no nudge needed there), hipcc's expansion outside (zero, denormal,
// huge, negative, NaN) behind a branch.  For the callers that may meet an exact zero whose root matters (sd_box: a point on
// the box's face has u = 0 and the sign of 0 + idst decides) and for runtime programs, whose constants are not known here.
SDM_HD float sqrt_rn(float x) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SDM_FAST_MATH)
    if (__builtin_expect(x >= 8.0779356694631609e-28f && x <= 1.2379400392853803e+27f, 1)) {
        const float y = __builtin_amdgcn_rsqf(x);
        const float g = x * y, h = 0.5f * y;
        return __builtin_fmaf(__builtin_fmaf(-g, g, x), h, g);
    }
#endif
    return sqrtf(x);
}

// 1 / sqrt(d) rounded twice, as glm::normalize takes it (sqrt, then the division), for ANY d: inside [2^-90, 2^90] by the two
// lean exact forms, outside (zero: a degenerate triangle or a vanishing gradient must give inf / NaN exactly as the reference
// does; denormal, huge, NaN) by hipcc's own expansions behind a branch no wave of a sane scene takes.
SDM_HD float inv_length_rn(float d) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SDM_FAST_MATH)
    if (__builtin_expect(d >= 8.0779356694631609e-28f && d <= 1.2379400392853803e+27f, 1)) {
        const float y = __builtin_amdgcn_rsqf(d);
        const float g = d * y, h = 0.5f * y;
        const float r = __builtin_fmaf(__builtin_fmaf(-g, g, d), h, g);          // = sqrtf(d), in [2^-45, 2^45]
        const float q = __builtin_amdgcn_rcpf(r);
        return __builtin_fmaf(__builtin_fmaf(-r, q, 1.0f), q, q);                // = 1.0f / r
    }
#endif
    return 1.0f / sqrtf(d);
}

// a / 3 (the centroid of the orientation test, compute_mesh_generation.cu:104).  Exact for every a whose quotient is normal; a
// zero keeps its magnitude (the sign of a zero centroid coordinate is lost: it only ever enters c + eps), and a quotient below
// 2^-126 may differ in its last denormal bit -- both are absorbed by the + -eps of the stencil it feeds.  (+-inf gives NaN where
// the division gives inf: the stencil around either is NaN, and the orientation test then keeps the triangle as it is.)
SDM_HD float third_rn(float a) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SDM_FAST_MATH)
    const float c = 0.333333343267440796f;
    const float q = a * c;
    return __builtin_fmaf(__builtin_fmaf(-3.0f, q, a), c, q);
#else
    return a / 3.0f;
#endif
}

SDM_HD f3 normalize3(f3 a) { return scale3(a, inv_length_rn(dot3(a, a))); }

SDM_HD float smooth_min(float a, float b, float k) {
    // (fmax1/fmin1: k - |a-b| and a, b are finite; if a == b both forms return the same value, and a zero h has
    // no effect on the result whatever its sign)
    float h = fmax1(k - fabsf(a - b), 0.0f) / k;
    return fmin1(a, b) - h * h * h * k * (1.0f / 6.0f);
}
// smooth_min(a, b, 0.5f) (sd_obj, common.cu:225) with one multiplication fewer, bit for bit: h^3 * 0.5 is exact (a power of two; h is
// 0 or >= 2^-24, so h^3 >= 2^-72 and nothing underflows), hence fl(fl(h^3 * 0.5) * c) = fl(h^3 * (c / 2)) with c = fl(1 / 6), c / 2 exact.
// (The compiler may not reassociate this itself without fast-math; `/ 0.5f` it does turn into an exact doubling.)
#define SDM_SIXTH_HALF 0.0833333358168601989746f          /* fl(1.0f / 6.0f) / 2, exactly */
// On the device h = max(0.5 - |d|, 0) / 0.5 is taken as max(fma(|d|, -2, 1), 0): 2 |d| is exact, so the fma rounds 1 - 2 |d| once, which is
// the doubling of fl(0.5 - |d|) bit for bit (doubling commutes with rounding; 0.5 - |d| is 0 or >= 2^-26, nothing underflows), and
// max(2 t, 0) == 2 max(t, 0).  One v_fma_f32 with an |abs| source modifier instead of v_and + v_sub + v_add per component.
SDM_HD float smooth_min_half(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float h = fmax1(__builtin_fmaf(fabsf(a - b), -2.0f, 1.0f), 0.0f);
#else
    const float h = fmax1(0.5f - fabsf(a - b), 0.0f) / 0.5f;
#endif
    return fmin1(a, b) - h * h * h * SDM_SIXTH_HALF;
}

// Squared residual along a bar's own axis: t = p - b0 (b0 = lo), bar length len.
//   t < 0   -> closest point b0            -> e = p - lo
//   t > len -> closest point b0 + len      -> e = p - (lo + len)   (lo + len is exact)
//   else    -> closest point b0 + t        -> e = p - (lo + t)     (rounding residue, kept)
// All three cases are lo + clamp(t, 0, len): lo + 0 == lo exactly (lo is never -0).
SDM_HD float along_sq(float p, float lo, float len) {
    float t = p - lo;
    float q = lo + clamp1(t, 0.0f, len);
    float e = p - q;
    return e * e;
}

SDM_HD float sq(float v) { return v * v; }

// Minimum over bar groups without forming every bar (value-preserving, exact):
// round-to-nearest addition is monotone non-decreasing in each operand, so for non-negative finite terms
//     min_{i,j} fl(fl(a_i + b) + c_j)  ==  fl(fl(min_i a_i + b) + min_j c_j)
// (the minimum is attained at the smallest operands; ties give equal values).  A group of four parallel bars that
// differ in two perpendicular offsets therefore costs 2 min + 2 add instead of 6 add + 3 min.  Moreover, for offsets
// +-c on one axis, min(sq(p - c), sq(p + c)) == sq(|p| - c) bit for bit: |p| - c is exactly (p - c) or -(p + c),
// whichever has the smaller magnitude (monotonicity again), and squaring discards the sign.
SDM_HD float fmin3(float a, float b, float c) { return fmin1(fmin1(a, b), c); }

// sd_box_skeleton(p, 0, (3, 1, 0.5), 0.1) with the reference's indexing bug:
//   x-bars: y in {-0.5, +0.5},  z in {-0.25, +0.25}, x from -1.5, len 3
//   y-bars: z in {-0.25, +2.75}, x in {-1.5, +1.5},  y from -0.5, len 1    (z offset uses bs[0]=3)
//   z-bars: x in {-1.5, -0.5},  y in {-0.5, +0.5},   z from -0.25, len 0.5 (x offset uses bs[1]=1)
// each d2 is (ex^2 + ey^2) + ez^2 in x, y, z order, as glm::dot evaluates it
struct SkelX { float ax, sx15, mx; };      // along-x residual, min over x in {+-1.5}, min over x in {-1.5, -0.5}
struct SkelY { float ay, sy; };            // along-y residual, min over y in {+-0.5}
struct SkelZ { float az, sz, mz; };        // along-z residual, min over z in {+-0.25}, min over z in {-0.25, +2.75}
SDM_HD SkelX skel_x(float px) {
    SkelX r;
    r.ax = along_sq(px, -1.5f, 3.0f);
    r.sx15 = sq(fabsf(px) - 1.5f);
    r.mx = fmin1(sq(px - (-1.5f)), sq(px - (-0.5f)));
    return r;
}
SDM_HD SkelY skel_y(float py) {
    SkelY r;
    r.ay = along_sq(py, -0.5f, 1.0f);
    r.sy = sq(fabsf(py) - 0.5f);
    return r;
}
SDM_HD SkelZ skel_z(float pz) {
    SkelZ r;
    r.az = along_sq(pz, -0.25f, 0.5f);
    r.sz = sq(fabsf(pz) - 0.25f);
    r.mz = fmin1(sq(pz - (-0.25f)), sq(pz - 2.75f));
    return r;
}
SDM_HD float skel_combine(const SkelX &x, const SkelY &y, const SkelZ &z) {
    const float bx = (x.ax + y.sy) + z.sz;          // the four x-bars
    const float by = (x.sx15 + y.ay) + z.mz;        // the four y-bars
    const float bz = (x.mx + y.sy) + z.az;          // the four z-bars
    return fmin3(bx, by, bz);
}
SDM_HD float skeleton_min_d2(float px, float py, float pz) { return skel_combine(skel_x(px), skel_y(py), skel_z(pz)); }

// sd_box_skeleton(p, 0, (5, 5, 5), lw): the bounding-cube wire frame of the preview (compute_render.cu:8-16).  All
// twelve bars are axis-aligned with ends at +-2.5 (with equal sides the `% 2` of signed_distance.cu:101 changes nothing),
// so the same rewrites apply: unit directions make sd_ray's dot / scale exact, the residual along a bar is along_sq,
// across it sq(|p| - 2.5) is the smaller of the two offsets, and min_i(sqrt(d2_i) - lw) = sqrt(min_i d2_i) - lw.
SDM_HD float skeleton_cube5_min_d2(float px, float py, float pz) {
    const float ax = along_sq(px, -2.5f, 5.0f), ay = along_sq(py, -2.5f, 5.0f), az = along_sq(pz, -2.5f, 5.0f);
    const float sx = sq(fabsf(px) - 2.5f), sy = sq(fabsf(py) - 2.5f), sz = sq(fabsf(pz) - 2.5f);
    return fmin3((ax + sy) + sz, (sx + ay) + sz, (sx + sy) + az);
}

SDM_HD float sd_box(f3 p, f3 bp, f3 bs) {
    f3 d = sub3(p, bp);
    f3 hs = mk(bs.x / 2.0f, bs.y / 2.0f, bs.z / 2.0f);
    f3 q = mk(fabsf(d.x) - hs.x, fabsf(d.y) - hs.y, fabsf(d.z) - hs.z);
    f3 u = mk(gmax(q.x, 0.0f), gmax(q.y, 0.0f), gmax(q.z, 0.0f));
    float udst = sqrt_rn(dot3(u, u));
    f3 m = mk(gmin(q.x, 0.0f), gmin(q.y, 0.0f), gmin(q.z, 0.0f));
    float idst = gmax(gmax(m.x, m.y), m.z);
    return udst + idst;
}

// ---- generic (runtime-parameter) primitives, literal op order of the reference -------------------------
SDM_HD float sd_ray_g(f3 p, f3 bl, f3 bd, float len) {              // signed_distance.cu:65-75
    float d = dot3(sub3(p, bl), bd);
    f3 q;
    if (d < 0) q = bl;
    else if (d > len) q = add3(bl, scale3(bd, len));
    else q = add3(bl, scale3(bd, d));
    f3 e = sub3(p, q);
    return sqrt_rn(dot3(e, e));
}

SDM_HD float sd_line_g(f3 p, f3 b0, f3 b1) {                        // signed_distance.cu:77-80
    f3 d = sub3(b1, b0);
    float len = sqrt_rn(dot3(d, d));
    return sd_ray_g(p, b0, mk(d.x / len, d.y / len, d.z / len), len);
}

SDM_HD float vget3(f3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
SDM_HD void vadd3i(f3 &v, int i, float s) { if (i == 0) v.x += s; else if (i == 1) v.y += s; else v.z += s; }

SDM_HD float sd_box_skeleton_g(f3 p, f3 bp, f3 bs, float lw) {      // signed_distance.cu:93-113 incl. the % 2 at :101
    f3 bpl = sub3(bp, mk(bs.x / 2.0f, bs.y / 2.0f, bs.z / 2.0f));
    float sd = 3.40282347E+38f;
    for (int dir = 0; dir < 3; dir++)
        for (int c0 = 0; c0 < 2; c0++)
            for (int c1 = 0; c1 < 2; c1++) {
                f3 m0 = bpl;
                vadd3i(m0, (dir + 1) % 3, c0 ? vget3(bs, (dir + 1) % 2) : 0.0f);
                vadd3i(m0, (dir + 2) % 3, c1 ? vget3(bs, (dir + 2) % 3) : 0.0f);
                f3 m1 = m0;
                vadd3i(m1, dir, vget3(bs, dir));
                sd = gmin(sd, sd_line_g(p, m0, m1) - lw);
            }
    return sd;
}

SDM_HD float wrap1(float x, float lower, float higher) {            // signed_distance.cu:9-11; glm::mod(x, y) = x - y * floor(x / y)
    const float xx = x - lower, y = higher - lower;
    return lower + (xx - y * floorf(xx / y));
}

SDM_HD float sdf_program(f3 p, const SceneParams &sp) {
    float acc = 3.40282347E+38f;
    const uint32_t n = sp.ninstr < SDM_MAX_INSTR ? sp.ninstr : SDM_MAX_INSTR;
    for (uint32_t i = 0; i < n; i++) {
        const SdfInstr &in = sp.prog[i];
        const f3 c = mk(in.a[0], in.a[1], in.a[2]);
        float d;
        if (in.op == OP_WRAP) {                                     // signed_distance.cu:13-18: the point of every later instruction
            p = mk(wrap1(p.x, in.a[0], in.a[3]), wrap1(p.y, in.a[1], in.a[4]), wrap1(p.z, in.a[2], in.a[5]));
            continue;
        }
        switch (in.op) {
        case OP_SPHERE: { f3 e = sub3(p, c); d = sqrt_rn(dot3(e, e)) - in.a[3]; } break;
        case OP_BOX: d = sd_box(p, c, mk(in.a[3], in.a[4], in.a[5])); break;
        case OP_BOX_SKELETON: d = sd_box_skeleton_g(p, c, mk(in.a[3], in.a[4], in.a[5]), in.a[6]); break;
        case OP_CAPSULE: d = sd_line_g(p, c, mk(in.a[3], in.a[4], in.a[5])) - in.a[6]; break;
        case OP_TORUS_Y: {
            f3 e = sub3(p, c);
            float qx = sqrt_rn(e.x * e.x + e.z * e.z) - in.a[3];
            d = sqrt_rn(qx * qx + e.y * e.y) - in.a[4];
        } break;
        case OP_RAY: {                                              // signed_distance.cu:61-63 (unbounded; a[6] = radius, 0 for the bare distance)
            const f3 bd = mk(in.a[3], in.a[4], in.a[5]);
            const f3 q = add3(c, scale3(bd, dot3(sub3(p, c), bd)));
            const f3 e = sub3(p, q);                                // distance(a, b) = length(b - a)
            d = sqrt_rn(dot3(e, e)) - in.a[6];
        } break;
        case OP_SIMPLE_BOX: {                                       // signed_distance.cu:115-118: the interior term of sd_box alone
            const f3 e = sub3(p, c);
            const f3 q = mk(fabsf(e.x) - in.a[3] / 2.0f, fabsf(e.y) - in.a[4] / 2.0f, fabsf(e.z) - in.a[5] / 2.0f);
            const f3 m = mk(gmin(q.x, 0.0f), gmin(q.y, 0.0f), gmin(q.z, 0.0f));
            d = gmax(gmax(m.x, m.y), m.z);
        } break;
        case OP_SIMPLE_BOUNDING_BOX:                                // signed_distance.cu:120-131: a[0..2] = bb_min, a[3..5] = bb_max
            d = gmax(gmax(in.a[0] - p.x, gmax(in.a[1] - p.y, in.a[2] - p.z)), gmax(p.x - in.a[3], gmax(p.y - in.a[4], p.z - in.a[5])));
            break;
        default: d = 3.40282347E+38f; break;
        }
        switch (in.combine) {
        case CB_SET: acc = d; break;
        case CB_MIN: acc = gmin(acc, d); break;
        case CB_MAX: acc = gmax(acc, d); break;
        case CB_SMOOTH_MIN: acc = smooth_min(acc, d, in.k); break;
        case CB_SUBTRACT: acc = gmax(acc, -d); break;
        default: break;
        }
    }
    return acc;
}


// ---------------------------------------------------------------------------------------------
// Elementary functions for sd_mandelbulb (signed_distance.cu:29-53).  The reference takes acos / atan2 / pow / sin / cos / log from CUDA's
// float library (fast-math variants in its build, build.rs:111-125): no two libraries agree on their last bits, so there is nothing to be
// bit-identical WITH.  What this file fixes instead: each function is evaluated in f64 from +, -, *, /, sqrt and fma only (IEEE, the same
// on the host and on the device), by the argument reductions and series written out below, and rounded ONCE to f32 -- the result is the
// correctly rounded f32 value except within ~1e-7 ulp of a rounding boundary (tests/test_host_numerics.py measures <= 1 ulp against
// f64 libm over each function's domain), and it is the same number on both sides, which is what the oracle comparison needs.
// ---------------------------------------------------------------------------------------------
SDM_HD double e_from_bits(uint64_t b) { double d; __builtin_memcpy(&d, &b, 8); return d; }
SDM_HD uint64_t e_bits(double d) { uint64_t b; __builtin_memcpy(&b, &d, 8); return b; }
SDM_HD double e_inf() { return e_from_bits(0x7FF0000000000000ull); }
SDM_HD double e_nan() { return e_from_bits(0x7FF8000000000000ull); }
#define SDM_FMA(a, b, c) __builtin_fma((a), (b), (c))

// log(x), x = an f32 value (so never an f64 subnormal): x = m 2^e with m in (sqrt(1/2), sqrt(2)], log m = 2 atanh(s), s = (m - 1) / (m + 1),
// |s| <= 0.1716: the series to s^19 leaves 2e-17.
SDM_HD double e_log(double x) {
    if (!(x > 0.0)) return x == 0.0 ? -e_inf() : e_nan();
    if (x == e_inf()) return x;
    const uint64_t b = e_bits(x);
    int e = (int) (b >> 52) - 1023;
    double m = e_from_bits((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);
    if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
    const double s = (m - 1.0) / (m + 1.0), z = s * s;
    double q = 1.0 / 19.0;
    q = SDM_FMA(q, z, 1.0 / 17.0); q = SDM_FMA(q, z, 1.0 / 15.0); q = SDM_FMA(q, z, 1.0 / 13.0); q = SDM_FMA(q, z, 1.0 / 11.0);
    q = SDM_FMA(q, z, 1.0 / 9.0); q = SDM_FMA(q, z, 1.0 / 7.0); q = SDM_FMA(q, z, 1.0 / 5.0); q = SDM_FMA(q, z, 1.0 / 3.0);
    q = SDM_FMA(q, z, 1.0);
    return SDM_FMA((double) e, 0.6931471805599453, 2.0 * s * q);
}

// exp(t): t = k ln2 + r, |r| <= 0.3466 (ln2 in two parts), Taylor to r^13 (r^14 / 14! = 4e-18), scaled by 2^k in two steps
SDM_HD double e_exp(double t) {
    if (t != t) return t;
    if (t > 709.0) return e_inf();
    if (t < -745.0) return 0.0;
    const double k = __builtin_rint(t * 1.4426950408889634);
    double r = SDM_FMA(-k, 6.93147180369123816490e-01, t);
    r = SDM_FMA(-k, 1.90821492927058770002e-10, r);
    double q = 1.0 / 6227020800.0;
    q = SDM_FMA(q, r, 1.0 / 479001600.0); q = SDM_FMA(q, r, 1.0 / 39916800.0); q = SDM_FMA(q, r, 1.0 / 3628800.0); q = SDM_FMA(q, r, 1.0 / 362880.0);
    q = SDM_FMA(q, r, 1.0 / 40320.0); q = SDM_FMA(q, r, 1.0 / 5040.0); q = SDM_FMA(q, r, 1.0 / 720.0); q = SDM_FMA(q, r, 1.0 / 120.0);
    q = SDM_FMA(q, r, 1.0 / 24.0); q = SDM_FMA(q, r, 1.0 / 6.0); q = SDM_FMA(q, r, 0.5); q = SDM_FMA(q, r, 1.0); q = SDM_FMA(q, r, 1.0);
    const int ki = (int) k, k1 = ki / 2, k2 = ki - k1;                    // |k1|, |k2| <= 538: both powers are normal
    return q * e_from_bits((uint64_t) (k1 + 1023) << 52) * e_from_bits((uint64_t) (k2 + 1023) << 52);
}

// x^y for x >= 0 (x = a length), as exp(y log x) with `lx` = e_log(x) shared between the two powers of one iteration
SDM_HD float e_powf_from_log(double lx, float x, float y) {
    if (y == 0.0f) return 1.0f;
    if (x != x || y != y) return x + y;
    return (float) e_exp((double) y * lx);
}

// sin and cos of an f32 angle: k = rint(x 2/pi), r = x - k pi/2 in two parts (|k| < 2^20: the first product is exact), |r| <= pi/4,
// Taylor to r^17 / r^16 (5e-17 left).  |x| >= 1e6 (never: angles here are below 2^10) and non-finite x give NaN.
SDM_HD void e_sincosf(float xf, float *sn, float *cs) {
    const double x = (double) xf;
    if (!(__builtin_fabs(x) < 1.0e6)) { const float n = (float) e_nan(); *sn = n; *cs = n; return; }
    const double k = __builtin_rint(x * 0.63661977236758134308);
    double r = SDM_FMA(-k, 1.57079632673412561417e+00, x);
    r = SDM_FMA(-k, 6.07710050650619224932e-11, r);
    const double z = r * r;
    double a = -1.0 / 355687428096000.0;                                  // sin: r (1 - z/3! + ... - z^8/17!)
    a = SDM_FMA(a, z, 1.0 / 1307674368000.0); a = SDM_FMA(a, z, -1.0 / 6227020800.0); a = SDM_FMA(a, z, 1.0 / 39916800.0);
    a = SDM_FMA(a, z, -1.0 / 362880.0); a = SDM_FMA(a, z, 1.0 / 5040.0); a = SDM_FMA(a, z, -1.0 / 120.0); a = SDM_FMA(a, z, 1.0 / 6.0);
    const double s = SDM_FMA(-a * z, r, r);
    double c = 1.0 / 20922789888000.0;                                    // cos: 1 - z/2! + ... + z^8/16!
    c = SDM_FMA(c, z, -1.0 / 87178291200.0); c = SDM_FMA(c, z, 1.0 / 479001600.0); c = SDM_FMA(c, z, -1.0 / 3628800.0);
    c = SDM_FMA(c, z, 1.0 / 40320.0); c = SDM_FMA(c, z, -1.0 / 720.0); c = SDM_FMA(c, z, 1.0 / 24.0); c = SDM_FMA(c, z, -0.5);
    c = SDM_FMA(c, z, 1.0);
    const int n = (int) k & 3;
    const double sv = (n & 1) ? c : s, cv = (n & 1) ? s : c;
    *sn = (float) ((n & 2) ? -sv : sv);
    *cs = (float) (((n + 1) & 2) ? -cv : cv);
}

// atan(t), 0 <= t <= 1: above tan(pi/8) through atan t = pi/4 + atan((t - 1) / (t + 1)); |w| <= 0.41422: the series to w^37 leaves 2e-16 w
SDM_HD double e_atan01(double t) {
    const bool hi = t > 0.41421356237309503;
    const double w = hi ? (t - 1.0) / (t + 1.0) : t, z = w * w;
    double q = 1.0 / 37.0;
    q = SDM_FMA(q, z, -1.0 / 35.0); q = SDM_FMA(q, z, 1.0 / 33.0); q = SDM_FMA(q, z, -1.0 / 31.0); q = SDM_FMA(q, z, 1.0 / 29.0);
    q = SDM_FMA(q, z, -1.0 / 27.0); q = SDM_FMA(q, z, 1.0 / 25.0); q = SDM_FMA(q, z, -1.0 / 23.0); q = SDM_FMA(q, z, 1.0 / 21.0);
    q = SDM_FMA(q, z, -1.0 / 19.0); q = SDM_FMA(q, z, 1.0 / 17.0); q = SDM_FMA(q, z, -1.0 / 15.0); q = SDM_FMA(q, z, 1.0 / 13.0);
    q = SDM_FMA(q, z, -1.0 / 11.0); q = SDM_FMA(q, z, 1.0 / 9.0); q = SDM_FMA(q, z, -1.0 / 7.0); q = SDM_FMA(q, z, 1.0 / 5.0);
    q = SDM_FMA(q, z, -1.0 / 3.0); q = SDM_FMA(q, z, 1.0);
    const double a = w * q;
    return hi ? 0.78539816339744830962 + a : a;
}

// atan2(y, x) in f64 with C's conventions for zeros and signs; an infinite operand gives NaN (never reached: the iteration has left by then)
SDM_HD double e_atan2(double y, double x) {
    if (x != x || y != y) return e_nan();
    const double ax = __builtin_fabs(x), ay = __builtin_fabs(y);
    const bool xneg = (e_bits(x) >> 63) != 0;
    double a;
    if (ax == 0.0 && ay == 0.0) a = xneg ? 3.14159265358979323846 : 0.0;
    else {
        const bool steep = ay > ax;
        a = e_atan01(steep ? ax / ay : ay / ax);
        if (steep) a = 1.57079632679489661923 - a;
        if (xneg) a = 3.14159265358979323846 - a;
    }
    return __builtin_copysign(a, y);
}
SDM_HD float e_atan2f(float y, float x) { return (float) e_atan2((double) y, (double) x); }
// acos(x) = 2 atan2(sqrt(1 - x), sqrt(1 + x)): NaN outside [-1, 1], as the library function
SDM_HD float e_acosf(float xf) {
    const double x = (double) xf;
    return (float) (2.0 * e_atan2(__builtin_sqrt(1.0 - x), __builtin_sqrt(1.0 + x)));
}
SDM_HD float e_logf(float x) { return (float) e_log((double) x); }

// signed_distance.cu:29-53, statement by statement in f32 (GLM: length = sqrt(dot), dot = (x x + y y) + z z; vec * float per component)
SDM_HD float sd_mandelbulb(f3 p, float time) {
    f3 z = p;
    float dr = 1.0f;
    float r = 0.0f;
    const float power = 7.0f * (1.0f + time * 0.001f);
    for (int i = 0; i < 25; i++) {
        r = sqrt_rn(dot3(z, z));
        if (r > 2.0f) break;
        const float theta = e_acosf(z.z / r) * power;
        const float phi = e_atan2f(z.y, z.x) * power;
        const double lr = e_log((double) r);
        const float zr = e_powf_from_log(lr, r, power);
        dr = e_powf_from_log(lr, r, power - 1.0f) * power * dr + 1.0f;
        float s_theta, c_theta, s_phi, c_phi;
        e_sincosf(theta, &s_theta, &c_theta);
        e_sincosf(phi, &s_phi, &c_phi);
        z = mk(zr * (s_theta * c_phi), zr * (s_phi * s_theta), zr * c_theta);
        z = add3(z, p);
    }
    return 0.5f * e_logf(r) * r / dr;
}

template <int SCENE>
SDM_HD float sdf(float px, float py, float pz, const SceneParams &sp) {
    if (SCENE == SCENE_PROGRAM) {
        return sdf_program(mk(px, py, pz), sp);
    } else if (SCENE == SCENE_BULB) {
        // sd_unit_mandelbulb's form (signed_distance.cu:55-57) with the scale and the time as scene parameters: p[0] = scale, p[1] = time
        return sd_mandelbulb(mk(px / sp.p[0], py / sp.p[0], pz / sp.p[0]), sp.p[1]) * sp.p[0];
    } else if (SCENE == SCENE_SPHERE) {
        return sqrt_rn_normal((px * px + py * py) + pz * pz) - 1.0f;
    } else if (SCENE == SCENE_CSG3) {
        // build-defined scene; mirrors oracle sdf_eval SCENE_CSG3
        float a = sqrt_rn_normal((px * px + py * py) + pz * pz) - 1.0f;
        float b = sd_box(mk(px, py, pz), mk(1.2f, 0.0f, 0.0f), mk(1.2f, 1.2f, 1.2f));
        float qx = sqrt_rn_normal(px * px + pz * pz) - 1.5f;
        float c = sqrt_rn_normal(qx * qx + py * py) - 0.25f;
        return smooth_min(smooth_min(a, b, 0.3f), c, 0.3f);
    } else {
        const float a1 = sqrt_rn_normal(skeleton_min_d2(px, py, pz)) - 0.1f;
        float sy = py;
        if (SCENE == SCENE_ANIM) sy = py - sp.p[0];
        const float a2 = sqrt_rn_normal((px * px + sy * sy) + pz * pz) - 1.0f;
        return smooth_min_half(a1, a2);
    }
}

// ---------------------------------------------------------------------------------------------
// Row context for dense sampling along z (k_sample_dense): everything that depends only on
// (x, y) is computed once per row; eval(z) performs the remaining operations in exactly the
// order sdf<SCENE>() does, so the values are bit-identical.
// ---------------------------------------------------------------------------------------------
template <int SCENE>
struct RowCtx {
    float px, py;
    float a[4];
    SDM_HDM void init(float x, float y, const SceneParams &sp) {
        px = x; py = y;
        if (SCENE == SCENE_SPHERE) {
            a[0] = px * px + py * py;
        } else if (SCENE == SCENE_SDOBJ || SCENE == SCENE_ANIM) {
            const SkelX sx = skel_x(px);
            const SkelY sy = skel_y(py);
            a[0] = sx.ax + sy.sy;            // x-bars, before + sz
            a[1] = sx.sx15 + sy.ay;          // y-bars, before + mz
            a[2] = sx.mx + sy.sy;            // z-bars, before + az
        }
    }
    SDM_HDM float eval(float pz, const SceneParams &sp) const {
        if (SCENE == SCENE_SPHERE) {
            return sqrt_rn_normal(a[0] + pz * pz) - 1.0f;
        } else if (SCENE == SCENE_SDOBJ || SCENE == SCENE_ANIM) {
            const SkelZ z = skel_z(pz);
            const float m = fmin3(a[0] + z.sz, a[1] + z.mz, a[2] + z.az);
            const float a1 = sqrt_rn_normal(m) - 0.1f;
            float sy = py;
            if (SCENE == SCENE_ANIM) sy = py - sp.p[0];
            const float a2 = sqrt_rn_normal((px * px + sy * sy) + pz * pz) - 1.0f;
            return smooth_min_half(a1, a2);
        } else {
            return sdf<SCENE>(px, py, pz, sp);
        }
    }
};

#define SDM_NORMAL_EPS 0.001f

#if defined(__HIP_DEVICE_COMPILE__) && !defined(SDM_NO_PAIRS)
// ---------------------------------------------------------------------------------------------
// Two stencil samples at a time (device only).  The projection and the orientation test are bound by VALU ISSUE
// (DESIGN.md 6): a packed f32 add / mul / fma (v_pk_*_f32) does two operations for 4.3 issue cycles where two scalar ones
// cost 2 x 3.0, and the +-e samples of one axis run the identical instruction sequence on two values.  Everything that
// has a packed form is written on a 2-vector; min / max / med3 / abs / sqrt / compare / select have none and stay per
// component.  Same operations in the same order on each component as sdf<SCENE>() -> the same bits
// (tests: every golden hash goes through this path).
// ---------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v2i __attribute__((ext_vector_type(2)));
SDM_HD v2f mk2(float a, float b) { v2f r; r.x = a; r.y = b; return r; }
SDM_HD v2f min2(v2f a, v2f b) { return mk2(fmin1(a.x, b.x), fmin1(a.y, b.y)); }
SDM_HD v2f max2s(v2f a, float b) { return mk2(fmax1(a.x, b), fmax1(a.y, b)); }
SDM_HD v2f abs2(v2f a) { return mk2(fabsf(a.x), fabsf(a.y)); }
SDM_HD v2f clamp2(v2f t, float lo, float hi) { return mk2(clamp1(t.x, lo, hi), clamp1(t.y, lo, hi)); }
SDM_HD v2f sq2(v2f v) { return v * v; }
SDM_HD v2f along_sq2(v2f p, float lo, float len) {
    const v2f t = p - lo;
    const v2f q = lo + clamp2(t, 0.0f, len);
    const v2f e = p - q;
    return e * e;
}
SDM_HD v2f sqrt_rn_normal2(v2f x) {
#if defined(SDM_FAST_MATH)
    return mk2(__builtin_amdgcn_sqrtf(x.x), __builtin_amdgcn_sqrtf(x.y));
#else
    const v2f xc = x + SDM_SQRT_NUDGE;                                   // sqrt_rn_normal() on both components
    const v2f y = mk2(__builtin_amdgcn_rsqf(xc.x), __builtin_amdgcn_rsqf(xc.y));
    const v2f g = xc * y, h = 0.5f * y;
    const v2f e = __builtin_elementwise_fma(-g, g, xc);
    return __builtin_elementwise_fma(e, h, g);
#endif
}
// the same with the nudge already inside x (the caller added 2^-120 to a term of the sum, see k_sample_dense_q1)
SDM_HD v2f sqrt_rn_normal2_nonudge(v2f xc) {
#if defined(SDM_FAST_MATH)
    return mk2(__builtin_amdgcn_sqrtf(xc.x), __builtin_amdgcn_sqrtf(xc.y));
#else
    const v2f y = mk2(__builtin_amdgcn_rsqf(xc.x), __builtin_amdgcn_rsqf(xc.y));
    const v2f g = xc * y, h = 0.5f * y;
    const v2f e = __builtin_elementwise_fma(-g, g, xc);
    return __builtin_elementwise_fma(e, h, g);
#endif
}
SDM_HD v2f smooth_min_half2(v2f a, v2f b) {                        // smooth_min_half() on both components
    const v2f d = a - b;                                             // (h as in smooth_min_half: one fma per component)
    const v2f h = mk2(fmax1(__builtin_fmaf(fabsf(d.x), -2.0f, 1.0f), 0.0f), fmax1(__builtin_fmaf(fabsf(d.y), -2.0f, 1.0f), 0.0f));
    return min2(a, b) - h * h * h * SDM_SIXTH_HALF;
}
SDM_HD v2f smooth_min2(v2f a, v2f b, float k) {
    const v2f h = max2s(k - abs2(a - b), 0.0f) / k;
    return min2(a, b) - h * h * h * k * (1.0f / 6.0f);
}
// sd_obj at (px.x, py, pz) and (px.y, py, pz) etc.: AXIS = the coordinate the two points differ in; `u` holds its two values
template <int SCENE, int AXIS>
SDM_HD v2f sdobj_pair(v2f u, float px, float py, float pz, const SceneParams &sp) {
    float sy = py;
    if (SCENE == SCENE_ANIM) sy = py - sp.p[0];
    v2f m, s;
    if (AXIS == 0) {
        const SkelY y = skel_y(py); const SkelZ z = skel_z(pz);
        const v2f ax = along_sq2(u, -1.5f, 3.0f), sx15 = sq2(abs2(u) - 1.5f), mx = min2(sq2(u - (-1.5f)), sq2(u - (-0.5f)));
        m = min2(min2((ax + y.sy) + z.sz, (sx15 + y.ay) + z.mz), (mx + y.sy) + z.az);
        s = (u * u + sy * sy) + pz * pz;
    } else if (AXIS == 1) {
        const SkelX x = skel_x(px); const SkelZ z = skel_z(pz);
        const v2f ay = along_sq2(u, -0.5f, 1.0f), sy2 = sq2(abs2(u) - 0.5f);
        m = min2(min2((x.ax + sy2) + z.sz, (x.sx15 + ay) + z.mz), (x.mx + sy2) + z.az);
        v2f sv = u;
        if (SCENE == SCENE_ANIM) sv = u - sp.p[0];
        s = (px * px + sv * sv) + pz * pz;
    } else {
        const SkelX x = skel_x(px); const SkelY y = skel_y(py);
        const v2f az = along_sq2(u, -0.25f, 0.5f), sz = sq2(abs2(u) - 0.25f), mz = min2(sq2(u - (-0.25f)), sq2(u - 2.75f));
        m = min2(min2((x.ax + y.sy) + sz, (x.sx15 + y.ay) + mz), (x.mx + y.sy) + az);
        s = (px * px + sy * sy) + u * u;
    }
    const v2f a1 = sqrt_rn_normal2(m) - 0.1f;
    const v2f a2 = sqrt_rn_normal2(s) - 1.0f;
    return smooth_min_half2(a1, a2);
}
template <int SCENE, int AXIS>
SDM_HD float sdobj_axis_derivative(float c, float px, float py, float pz, const SceneParams &sp) {
    const float e1 = SDM_NORMAL_EPS, e2 = 2.0f * SDM_NORMAL_EPS;
    const v2f o = sdobj_pair<SCENE, AXIS>(mk2(c + e2, c + (-e2)), px, py, pz, sp);      // {f(+2e), f(-2e)}
    const v2f i = sdobj_pair<SCENE, AXIS>(mk2(c + e1, c + (-e1)), px, py, pz, sp);      // {f(+e), f(-e)}
    return (-o.x + 8.0f * i.x - 8.0f * i.y + o.y);
}
// two z samples of one (x, y) row (k_sample_dense_rows): RowCtx::eval() on a 2-vector
template <int SCENE>
SDM_HD v2f row_eval2(const RowCtx<SCENE> &rc, v2f pz, const SceneParams &sp) {
    if (SCENE == SCENE_SPHERE) {
        return sqrt_rn_normal2(rc.a[0] + pz * pz) - 1.0f;
    } else if (SCENE == SCENE_SDOBJ || SCENE == SCENE_ANIM) {
        const v2f az = along_sq2(pz, -0.25f, 0.5f), sz = sq2(abs2(pz) - 0.25f), mz = min2(sq2(pz - (-0.25f)), sq2(pz - 2.75f));
        const v2f m = min2(min2(rc.a[0] + sz, rc.a[1] + mz), rc.a[2] + az);
        const v2f a1 = sqrt_rn_normal2(m) - 0.1f;
        float sy = rc.py;
        if (SCENE == SCENE_ANIM) sy = rc.py - sp.p[0];
        const v2f a2 = sqrt_rn_normal2((rc.px * rc.px + sy * sy) + pz * pz) - 1.0f;
        return smooth_min_half2(a1, a2);
    } else {
        return mk2(rc.eval(pz.x, sp), rc.eval(pz.y, sp));
    }
}
#define SDM_HAVE_PAIRS 1
#endif

// 4th-order central stencil, un-normalised weights, then normalize (signed_distance.cu:179-202)
template <int SCENE>
SDM_HD f3 empirical_normal(f3 p, const SceneParams &sp) {
#if defined(SDM_HAVE_PAIRS)
    if (SCENE == SCENE_SDOBJ || SCENE == SCENE_ANIM) {
        return normalize3(mk(sdobj_axis_derivative<SCENE, 0>(p.x, p.x, p.y, p.z, sp), sdobj_axis_derivative<SCENE, 1>(p.y, p.x, p.y, p.z, sp),
                             sdobj_axis_derivative<SCENE, 2>(p.z, p.x, p.y, p.z, sp)));
    }
#endif
    const float e1 = SDM_NORMAL_EPS, e2 = 2.0f * SDM_NORMAL_EPS;
    float dx = (-sdf<SCENE>(p.x + e2, p.y, p.z, sp) + 8.0f * sdf<SCENE>(p.x + e1, p.y, p.z, sp) -
                8.0f * sdf<SCENE>(p.x + (-e1), p.y, p.z, sp) + sdf<SCENE>(p.x + (-e2), p.y, p.z, sp));
    float dy = (-sdf<SCENE>(p.x, p.y + e2, p.z, sp) + 8.0f * sdf<SCENE>(p.x, p.y + e1, p.z, sp) -
                8.0f * sdf<SCENE>(p.x, p.y + (-e1), p.z, sp) + sdf<SCENE>(p.x, p.y + (-e2), p.z, sp));
    float dz = (-sdf<SCENE>(p.x, p.y, p.z + e2, sp) + 8.0f * sdf<SCENE>(p.x, p.y, p.z + e1, sp) -
                8.0f * sdf<SCENE>(p.x, p.y, p.z + (-e1), sp) + sdf<SCENE>(p.x, p.y, p.z + (-e2), sp));
    return normalize3(mk(dx, dy, dz));
}

// One iteration of closest_surface_point: g <- g - sd(g) * n(g); returns true when the
// PRE-step |sd| <= 1e-5 (the step is applied on the converging iteration too).
template <int SCENE>
SDM_HD bool newton_round(f3 &g, const SceneParams &sp) {
    float sd = sdf<SCENE>(g.x, g.y, g.z, sp);
    f3 n = empirical_normal<SCENE>(g, sp);
    g = sub3(g, scale3(n, sd));
    return fabsf(sd) <= 0.00001f;
}

#define SDM_MAX_NEWTON 10000

template <int SCENE>
SDM_HD f3 closest_surface_point(f3 p, const SceneParams &sp) {
    f3 g = p;
    bool collision = false;
    for (int i = 0; !collision && i < SDM_MAX_NEWTON; i++) collision = newton_round<SCENE>(g, sp);
    return g;
}

// Triangle orientation test of compute_mesh_generation.cu:102-105
template <int SCENE>
SDM_HD bool change_orientation(f3 v0, f3 v1, f3 v2, const SceneParams &sp) {
    f3 tn = normalize3(cross3(sub3(v1, v0), sub3(v2, v0)));
    f3 s = add3(add3(v0, v1), v2);
    f3 c = mk(third_rn(s.x), third_rn(s.y), third_rn(s.z));
    f3 an = empirical_normal<SCENE>(c, sp);
    return dot3(tn, an) <= 0.0f;
}

// Weld quantisation of src/cuda/mod.rs:270: (x * 10e4f32).round() as i64
SDM_HD long long weld_quant(float x) {
    float r = roundf(x * 100000.0f);
    if (!(r == r)) return 0;
    if (r >= 9.2233720368547758e18f) return 0x7fffffffffffffffLL;
    if (r <= -9.2233720368547758e18f) return (long long) 0x8000000000000000ULL;
    return (long long) r;
}

}  // namespace sdm
