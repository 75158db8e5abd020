/*
 * sdmesh_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A scalar, strict-IEEE-f32, plain-C restatement of the mesh-generation path of
 * Meterius/bevy-signed-distance-mesh-generation, written to be read side by side with
 * the reference (every function cites the reference file:line it follows).  It is the
 * checker for the HIP path in tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg.  Nothing under bevy-signed-distance-mesh-generation_amd/ may
 * include, link or call it.
 *
 * PARITY STATUS: "parity unpinned" in the task's sense -- the reference ships no tests,
 * fixtures or golden vectors for this path, and it cannot be built here (needs nvcc,
 * the CUDA headers and GLM 1.0.1, none of which are in the image or vendored in the
 * reference; building it behind stand-in headers is not allowed).  What pins this file
 * instead: (1) the counts recorded in SURVEY.md section 8(c) during the survey phase
 * (tests/golden/survey_8c.json: voxels, emitting voxels, triangles, welded vertices and the
 * tris/voxel histogram for sd_obj k=0..5 and the sphere k=0..4) -- this oracle reproduces
 * every row; the FNV hash columns of that table could not be reproduced by any FNV variant
 * (not even for the scene-independent level-0 field) and are not used;
 * (2) closed-mesh invariants (V - T/2 = chi, every edge shared by exactly two triangles);
 * (3) the DATA the reference itself holds for the path, parsed from its files as text by tools/pin_mc_tables.py into
 * tests/golden/NAME_ref.json (mc_tables, bindings, path_constants) and checked against this file and the product by tests/test_reference_pins.py: the
 * marching-cubes tables, the FFI header (constants, struct fields) and the path's numeric literals (stencil step and
 * weights, Newton threshold and cap, midpoint factor, inside / flip tests, sd_obj's parameters, weld quantum).
 *
 * Arithmetic model: GLM 1.0.1 generic scalar paths, no FMA
 * contraction (compile with -ffp-contract=off), IEEE division and sqrt:
 *   dot(a,b)      = (a.x*b.x + a.y*b.y) + a.z*b.z
 *   length(v)     = sqrt(dot(v,v));  distance(a,b) = length(b - a)
 *   normalize(v)  = v * (1 / sqrt(dot(v,v)))
 *   cross(x,y)    = (x.y*y.z - y.y*x.z, x.z*y.x - y.z*x.x, x.x*y.y - y.x*x.y)
 *   mix(x,y,a)    = x*(1-a) + y*a
 *   min(a,b)      = (b<a)?b:a ;  max(a,b) = (a<b)?b:a
 * The real reference is built with nvcc --use_fast_math (build.rs:111-125), so it is not
 * bit-reproducible off NVIDIA hardware anyway; this IEEE model is the oracle of record.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "mc_tables.h"

/* ---- boundary structs: cuda/includes/bindings.h:43-75 ---------------------------- */
typedef struct { float x, y, z; } Point;
typedef struct { Point position, normal; } Vertex;
typedef struct { Vertex vertices[3]; } Triangle;

_Static_assert(sizeof(Point) == 12, "Point");
_Static_assert(sizeof(Vertex) == 24, "Vertex");
_Static_assert(sizeof(Triangle) == 72, "Triangle");

#define BLOCK_SIZE 128                       /* bindings.h:7  */
#define MESH_GENERATION_INIT_FACTOR 32       /* bindings.h:9  */
#define MESH_GENERATION_BB_SIZE 5.0f         /* bindings.h:10 */

/* ---- scenes ------------------------------------------------------------------------
 * 0 SDOBJ  : the reference scene, cuda/modules/common.cu:222-226
 * 1 SPHERE : length(p) - 1 (the a2 term of sd_obj, common.cu:224)
 * 2 CSG3   : build-defined -- no reference counterpart
 * 3 ANIM   : build-defined: sd_obj with the sphere centre at
 *            (0, params[0], 0); the host passes params[0] = 0.5*sin(t)
 * 5 BULB   : sd_mandelbulb (cuda/includes/signed_distance.cu:29-53) in sd_unit_mandelbulb's form (:55-57):
 *            sd_mandelbulb(p / params[0], params[1]) * params[0]; the reference never calls it
 */
enum { SCENE_SDOBJ = 0, SCENE_SPHERE = 1, SCENE_CSG3 = 2, SCENE_ANIM = 3, SCENE_PROGRAM = 4, SCENE_BULB = 5 };

/* 4 PROGRAM: build-defined runtime SDF program, a linear CSG chain of the reference's
 * primitives: acc = combine_i(acc, primitive_i(p)).  Same instruction layout as include/sdmesh.h sdmesh_instr. */
enum { OP_SPHERE = 1, OP_BOX = 2, OP_BOX_SKELETON = 3, OP_CAPSULE = 4, OP_TORUS_Y = 5, OP_RAY = 6, OP_SIMPLE_BOX = 7, OP_SIMPLE_BOUNDING_BOX = 8,
       OP_WRAP = 9 };
enum { CB_SET = 0, CB_MIN = 1, CB_MAX = 2, CB_SMOOTH_MIN = 3, CB_SUBTRACT = 4 };
typedef struct { uint32_t op, combine; float k; float a[7]; } OrcInstr;
static OrcInstr g_prog[16];
static int g_nprog;
void orc_set_program(const void *instrs, int n) {
    g_nprog = n < 0 ? 0 : (n > 16 ? 16 : n);
    memcpy(g_prog, instrs, sizeof(OrcInstr) * (size_t) g_nprog);
}

typedef struct { float x, y, z; } v3;

static _Thread_local uint64_t tl_evals;
static uint64_t g_evals;

static inline v3 V(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vdivs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
static inline float gmin(float a, float b) { return (b < a) ? b : a; }
static inline float gmax(float a, float b) { return (a < b) ? b : a; }
static inline float gabs(float a) { return fabsf(a); }
static inline float dot3(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float length3(v3 a) { return sqrtf(dot3(a, a)); }
static inline float distance3(v3 a, v3 b) { return length3(vsub(b, a)); }
static inline v3 normalize3(v3 a) { return vscale(a, 1.0f / sqrtf(dot3(a, a))); }
static inline v3 cross3(v3 x, v3 y) {
    return V(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
static inline float vget(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
static inline void vaddi(v3 *a, int i, float s) {
    if (i == 0) a->x += s; else if (i == 1) a->y += s; else a->z += s;
}

/* signed_distance.cu:9-11; glm::mod(x, y) = x - y * floor(x / y) */
static float wrap1(float x, float lower, float higher) {
    float xx = x - lower, y = higher - lower;
    return lower + (xx - y * floorf(xx / y));
}

/* signed_distance.cu:61-63 */
static float sd_ray_unbounded(v3 p, v3 bl, v3 bd) {
    return distance3(vadd(bl, vscale(bd, dot3(vsub(p, bl), bd))), p);
}

/* signed_distance.cu:115-118 */
static float sd_simple_box(v3 p, v3 bp, v3 bs) {
    v3 e = vsub(p, bp);
    v3 q = V(gabs(e.x) - bs.x / 2.0f, gabs(e.y) - bs.y / 2.0f, gabs(e.z) - bs.z / 2.0f);
    v3 m = V(gmin(q.x, 0.0f), gmin(q.y, 0.0f), gmin(q.z, 0.0f));
    return gmax(gmax(m.x, m.y), m.z);
}

/* signed_distance.cu:120-131 */
static float sd_simple_bounding_box(v3 p, v3 bb_min, v3 bb_max) {
    return gmax(gmax(bb_min.x - p.x, gmax(bb_min.y - p.y, bb_min.z - p.z)), gmax(p.x - bb_max.x, gmax(p.y - bb_max.y, p.z - bb_max.z)));
}

/* signed_distance.cu:20-23 */
static float smooth_min(float a, float b, float k) {
    float h = gmax(k - gabs(a - b), 0.0f) / k;
    return gmin(a, b) - h * h * h * k * (1.0f / 6.0f);
}

/* signed_distance.cu:65-75 (the 4-argument overload) */
static float sd_ray(v3 p, v3 bl, v3 bd, float len) {
    float d = dot3(vsub(p, bl), bd);
    if (d < 0) {
        return distance3(bl, p);
    } else if (d > len) {
        return distance3(vadd(bl, vscale(bd, len)), p);
    }
    return distance3(vadd(bl, vscale(bd, d)), p);
}

/* signed_distance.cu:77-80 */
static float sd_line(v3 p, v3 b0, v3 b1) {
    float len = length3(vsub(b1, b0));
    return sd_ray(p, b0, vdivs(vsub(b1, b0), len), len);
}

/* signed_distance.cu:86-91 */
static float sd_box(v3 p, v3 bp, v3 bs) {
    v3 d = vsub(p, bp);
    v3 hs = vdivs(bs, 2.0f);
    v3 q = V(gabs(d.x) - hs.x, gabs(d.y) - hs.y, gabs(d.z) - hs.z);
    float udst = length3(V(gmax(q.x, 0.0f), gmax(q.y, 0.0f), gmax(q.z, 0.0f)));
    v3 m = V(gmin(q.x, 0.0f), gmin(q.y, 0.0f), gmin(q.z, 0.0f));
    float idst = gmax(gmax(m.x, m.y), m.z);   /* utils.cu:20 maximum(vec3) */
    return udst + idst;
}

/* signed_distance.cu:93-113, INCLUDING the (dir + 1) % 2 indexing bug at :101 */
static float sd_box_skeleton(v3 p, v3 bp, v3 bs, float lw) {
    v3 bpl = vsub(bp, vdivs(bs, 2.0f));
    float sd = (float) 3.40282347E+38;        /* utils.cu:10 MAX_POSITIVE_F32 */
    for (int dir = 0; dir < 3; dir++) {
        for (int c0 = 0; c0 < 2; c0++) {
            for (int c1 = 0; c1 < 2; c1++) {
                v3 m0 = bpl;
                vaddi(&m0, (dir + 1) % 3, c0 ? vget(bs, (dir + 1) % 2) : 0.0f);
                vaddi(&m0, (dir + 2) % 3, c1 ? vget(bs, (dir + 2) % 3) : 0.0f);
                v3 m1 = m0;
                vaddi(&m1, dir, vget(bs, dir));
                sd = gmin(sd, sd_line(p, m0, m1) - lw);
            }
        }
    }
    return sd;
}

/* build-defined torus around the y axis: length((length(p.xz)-R, p.y)) - r */
static float sd_torus_y(v3 p, float R, float r) {
    float qx = sqrtf(p.x * p.x + p.z * p.z) - R;
    return sqrtf(qx * qx + p.y * p.y) - r;
}

/* ---- sd_mandelbulb (signed_distance.cu:29-53) ---------------------------------------------------
 * The reference takes acos / atan2 / pow / sin / cos / log from CUDA's float library, built with --use_fast_math
 * (build.rs:111-125): values no other library reproduces.  PARITY UNPINNED for this scene beyond "the same
 * statements in the same order": the elementary functions below are this build's definition of them -- each
 * evaluated in double from + - * / sqrt fma by a fixed reduction and a fixed series, rounded once to float --
 * restated here from the description in csrc/sdf.hpp, coefficient tables and Horner loops instead of unrolled
 * code.  tests/test_host_numerics.py bounds them against glibc's double functions (<= 1 ulp of float). */
static double horner(const double *c, int n, double z) {
    double q = c[0];
    for (int i = 1; i < n; i++) q = fma(q, z, c[i]);
    return q;
}
static double dbl_from_bits(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
static uint64_t dbl_bits(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }

static double el_log(double x) {                 /* x: a float's value, so never a double subnormal */
    static const double C[10] = {1.0 / 19.0, 1.0 / 17.0, 1.0 / 15.0, 1.0 / 13.0, 1.0 / 11.0, 1.0 / 9.0, 1.0 / 7.0, 1.0 / 5.0, 1.0 / 3.0, 1.0};
    if (!(x > 0.0)) return x == 0.0 ? -INFINITY : NAN;
    if (isinf(x)) return x;
    uint64_t b = dbl_bits(x);
    int e = (int) (b >> 52) - 1023;
    double m = dbl_from_bits((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);
    if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
    double s = (m - 1.0) / (m + 1.0);
    return fma((double) e, 0.6931471805599453, 2.0 * s * horner(C, 10, s * s));
}

static double el_exp(double t) {
    static const double C[14] = {1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0,
                                 1.0 / 5040.0, 1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5, 1.0, 1.0};
    if (t != t) return t;
    if (t > 709.0) return INFINITY;
    if (t < -745.0) return 0.0;
    double k = rint(t * 1.4426950408889634);
    double r = fma(-k, 6.93147180369123816490e-01, t);
    r = fma(-k, 1.90821492927058770002e-10, r);
    int ki = (int) k, k1 = ki / 2, k2 = ki - k1;
    return horner(C, 14, r) * dbl_from_bits((uint64_t) (k1 + 1023) << 52) * dbl_from_bits((uint64_t) (k2 + 1023) << 52);
}

static float el_powf_from_log(double lx, float x, float y) {      /* x^y = exp(y log x), x >= 0 */
    if (y == 0.0f) return 1.0f;
    if (x != x || y != y) return x + y;
    return (float) el_exp((double) y * lx);
}

static void el_sincosf(float xf, float *sn, float *cs) {
    static const double S[8] = {-1.0 / 355687428096000.0, 1.0 / 1307674368000.0, -1.0 / 6227020800.0, 1.0 / 39916800.0, -1.0 / 362880.0,
                                1.0 / 5040.0, -1.0 / 120.0, 1.0 / 6.0};
    static const double Cc[9] = {1.0 / 20922789888000.0, -1.0 / 87178291200.0, 1.0 / 479001600.0, -1.0 / 3628800.0, 1.0 / 40320.0,
                                 -1.0 / 720.0, 1.0 / 24.0, -0.5, 1.0};
    double x = (double) xf;
    if (!(fabs(x) < 1.0e6)) { *sn = NAN; *cs = NAN; return; }
    double k = rint(x * 0.63661977236758134308);
    double r = fma(-k, 1.57079632673412561417e+00, x);
    r = fma(-k, 6.07710050650619224932e-11, r);
    double z = r * r;
    double s = fma(-horner(S, 8, z) * z, r, r), c = horner(Cc, 9, z);
    int n = (int) k & 3;
    double sv = (n & 1) ? c : s, cv = (n & 1) ? s : c;
    *sn = (float) ((n & 2) ? -sv : sv);
    *cs = (float) (((n + 1) & 2) ? -cv : cv);
}

static double el_atan01(double t) {               /* 0 <= t <= 1 */
    static const double C[19] = {1.0 / 37.0, -1.0 / 35.0, 1.0 / 33.0, -1.0 / 31.0, 1.0 / 29.0, -1.0 / 27.0, 1.0 / 25.0, -1.0 / 23.0, 1.0 / 21.0,
                                 -1.0 / 19.0, 1.0 / 17.0, -1.0 / 15.0, 1.0 / 13.0, -1.0 / 11.0, 1.0 / 9.0, -1.0 / 7.0, 1.0 / 5.0, -1.0 / 3.0, 1.0};
    int hi = t > 0.41421356237309503;
    double w = hi ? (t - 1.0) / (t + 1.0) : t;
    double a = w * horner(C, 19, w * w);
    return hi ? 0.78539816339744830962 + a : a;
}

static double el_atan2(double y, double x) {
    if (x != x || y != y) return NAN;
    double ax = fabs(x), ay = fabs(y), a;
    int xneg = (int) (dbl_bits(x) >> 63);
    if (ax == 0.0 && ay == 0.0) a = xneg ? 3.14159265358979323846 : 0.0;
    else {
        int steep = ay > ax;
        a = el_atan01(steep ? ax / ay : ay / ax);
        if (steep) a = 1.57079632679489661923 - a;
        if (xneg) a = 3.14159265358979323846 - a;
    }
    return copysign(a, y);
}
static float el_atan2f(float y, float x) { return (float) el_atan2((double) y, (double) x); }
static float el_acosf(float xf) { double x = (double) xf; return (float) (2.0 * el_atan2(sqrt(1.0 - x), sqrt(1.0 + x))); }
static float el_logf(float x) { return (float) el_log((double) x); }

/* the functions above, one by one, for the accuracy test: which = 0 log, 1 exp, 2 sin, 3 cos, 4 atan2(a, b), 5 acos, 6 pow(a, b) */
float orc_elementary(int which, float a, float b) {
    float s, c;
    switch (which) {
    case 0: return el_logf(a);
    case 1: return (float) el_exp((double) a);
    case 2: el_sincosf(a, &s, &c); return s;
    case 3: el_sincosf(a, &s, &c); return c;
    case 4: return el_atan2f(a, b);
    case 5: return el_acosf(a);
    default: return el_powf_from_log(el_log((double) a), a, b);
    }
}

static float sd_mandelbulb(v3 p, float time) {     /* signed_distance.cu:29-53 */
    v3 z = p;
    float dr = 1.0f;
    float r = 0.0f;
    float power = 7.0f * (1.0f + time * 0.001f);   /* POWER * (1.0f + time * 0.001f), :27,34 */
    for (int i = 0; i < 25; i++) {
        r = length3(z);
        if (r > 2.0f) break;
        float theta = el_acosf(z.z / r) * power;
        float phi = el_atan2f(z.y, z.x) * power;
        double lr = el_log((double) r);
        float zr = el_powf_from_log(lr, r, power);
        dr = el_powf_from_log(lr, r, power - 1.0f) * power * dr + 1.0f;
        float st, ct, sp, cp;
        el_sincosf(theta, &st, &ct);
        el_sincosf(phi, &sp, &cp);
        z = V(zr * (st * cp), zr * (sp * st), zr * ct);      /* zr * vec3(s_theta * cos(phi), sin(phi) * s_theta, cos(theta)), :48 */
        z = vadd(z, p);
    }
    return 0.5f * el_logf(r) * r / dr;
}

static float sdf_eval(int scene, const float *params, v3 p) {
    tl_evals++;
    switch (scene) {
    case SCENE_SPHERE:
        return length3(p) - 1.0f;
    case SCENE_CSG3: {
        float a = length3(p) - 1.0f;
        float b = sd_box(p, V(1.2f, 0.0f, 0.0f), V(1.2f, 1.2f, 1.2f));
        float c = sd_torus_y(p, 1.5f, 0.25f);
        return smooth_min(smooth_min(a, b, 0.3f), c, 0.3f);
    }
    case SCENE_PROGRAM: {
        float acc = (float) 3.40282347E+38;
        for (int i = 0; i < g_nprog; i++) {
            const OrcInstr *in = &g_prog[i];
            v3 c = V(in->a[0], in->a[1], in->a[2]);
            float d;
            if (in->op == OP_WRAP) {      /* signed_distance.cu:13-18: wraps the point every later instruction sees */
                p = V(wrap1(p.x, in->a[0], in->a[3]), wrap1(p.y, in->a[1], in->a[4]), wrap1(p.z, in->a[2], in->a[5]));
                continue;
            }
            switch (in->op) {
            case OP_SPHERE: d = length3(vsub(p, c)) - in->a[3]; break;
            case OP_BOX: d = sd_box(p, c, V(in->a[3], in->a[4], in->a[5])); break;
            case OP_BOX_SKELETON: d = sd_box_skeleton(p, c, V(in->a[3], in->a[4], in->a[5]), in->a[6]); break;
            case OP_CAPSULE: d = sd_line(p, c, V(in->a[3], in->a[4], in->a[5])) - in->a[6]; break;
            case OP_TORUS_Y: d = sd_torus_y(vsub(p, c), in->a[3], in->a[4]); break;
            case OP_RAY: d = sd_ray_unbounded(p, c, V(in->a[3], in->a[4], in->a[5])) - in->a[6]; break;
            case OP_SIMPLE_BOX: d = sd_simple_box(p, c, V(in->a[3], in->a[4], in->a[5])); break;
            case OP_SIMPLE_BOUNDING_BOX: d = sd_simple_bounding_box(p, c, V(in->a[3], in->a[4], in->a[5])); break;
            default: d = (float) 3.40282347E+38; break;
            }
            switch (in->combine) {
            case CB_SET: acc = d; break;
            case CB_MIN: acc = gmin(acc, d); break;
            case CB_MAX: acc = gmax(acc, d); break;
            case CB_SMOOTH_MIN: acc = smooth_min(acc, d, in->k); break;
            case CB_SUBTRACT: acc = gmax(acc, -d); break;
            default: break;
            }
        }
        return acc;
    }
    case SCENE_BULB:
        return sd_mandelbulb(V(p.x / params[0], p.y / params[0], p.z / params[0]), params[1]) * params[0];
    case SCENE_ANIM: {
        float a1 = sd_box_skeleton(p, V(0.0f, 0.0f, 0.0f), V(3.0f, 1.0f, 0.5f), 0.1f);
        float a2 = length3(vsub(p, V(0.0f, params[0], 0.0f))) - 1.0f;
        return smooth_min(a1, a2, 0.5f);
    }
    default: { /* SCENE_SDOBJ: common.cu:222-226 */
        float a1 = sd_box_skeleton(p, V(0.0f, 0.0f, 0.0f), V(3.0f, 1.0f, 0.5f), 0.1f);
        float a2 = length3(p) - 1.0f;
        return smooth_min(a1, a2, 0.5f);
    }
    }
}

float orc_sdf(int scene, const float *params, float x, float y, float z) {
    return sdf_eval(scene, params, V(x, y, z));
}

/* signed_distance.cu:179-202 */
#define NORMAL_EPSILON 0.001f
static v3 empirical_normal(int scene, const float *params, v3 p) {
#define F(dx, dy, dz) sdf_eval(scene, params, vadd(p, V(dx, dy, dz)))
    float dx = (-F(2.0f * NORMAL_EPSILON, 0.0f, 0.0f) +
                8.0f * F(NORMAL_EPSILON, 0.0f, 0.0f) -
                8.0f * F(-NORMAL_EPSILON, 0.0f, 0.0f) +
                F(-2.0f * NORMAL_EPSILON, 0.0f, 0.0f));
    float dy = (-F(0.0f, 2.0f * NORMAL_EPSILON, 0.0f) +
                8.0f * F(0.0f, NORMAL_EPSILON, 0.0f) -
                8.0f * F(0.0f, -NORMAL_EPSILON, 0.0f) +
                F(0.0f, -2.0f * NORMAL_EPSILON, 0.0f));
    float dz = (-F(0.0f, 0.0f, 2.0f * NORMAL_EPSILON) +
                8.0f * F(0.0f, 0.0f, NORMAL_EPSILON) -
                8.0f * F(0.0f, 0.0f, -NORMAL_EPSILON) +
                F(0.0f, 0.0f, -2.0f * NORMAL_EPSILON));
#undef F
    return normalize3(V(dx, dy, dz));
}

/* signed_distance.cu:227-240.  Note the step is applied on the converging iteration too. */
static v3 closest_surface_point(int scene, const float *params, v3 p, uint32_t *iters) {
    v3 g = p;
    int collision = 0;
    int i;
    for (i = 0; !collision && i < 10000; i++) {
        float sd = sdf_eval(scene, params, g);
        v3 n = empirical_normal(scene, params, g);
        g = vsub(g, vscale(n, sd));
        collision = gabs(sd) <= 0.00001f;
    }
    if (iters) *iters = (uint32_t) i;
    return g;
}

static inline v3 from_point(Point p) { return V(p.x, p.y, p.z); }   /* utils.cu:26 */
static inline Point to_point(v3 p) { Point r = { p.x, p.y, p.z }; return r; } /* utils.cu:30 */

static void flush_evals(void) {
#pragma omp atomic
    g_evals += tl_evals;
    tl_evals = 0;
}

uint64_t orc_eval_count(int reset) {
    flush_evals();
    uint64_t r = g_evals;
    if (reset) g_evals = 0;
    return r;
}

void orc_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : 1);
#else
    (void) n;
#endif
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---- A1: CudaHandler::create_cuda_voxel_field, src/cuda/mod.rs:105-122 ------------- */
void orc_create_voxel_field(Point *voxels /* 32^3 */, Point *voxel_size) {
    const float SIZE = MESH_GENERATION_BB_SIZE / (float) MESH_GENERATION_INIT_FACTOR;
    voxel_size->x = SIZE; voxel_size->y = SIZE; voxel_size->z = SIZE;
    size_t o = 0;
    for (int x = 0; x < MESH_GENERATION_INIT_FACTOR; x++)
        for (int y = 0; y < MESH_GENERATION_INIT_FACTOR; y++)
            for (int z = 0; z < MESH_GENERATION_INIT_FACTOR; z++) {
                voxels[o].x = ((float) x) * SIZE - MESH_GENERATION_BB_SIZE / 2.0f;
                voxels[o].y = ((float) y) * SIZE - MESH_GENERATION_BB_SIZE / 2.0f;
                voxels[o].z = ((float) z) * SIZE - MESH_GENERATION_BB_SIZE / 2.0f;
                o++;
            }
}

/* ---- A2: compute_refine_voxel_field_by_sdf, compute_mesh_generation.cu:12-62 -------
 * One "thread" per parent id; writes the raw 8n layout, INFINITY for non-border children.
 */
static int obj_contains(int scene, const float *params, v3 p) {   /* :8-10 */
    return sdf_eval(scene, params, p) <= 0.0f;
}

void orc_refine_raw(int scene, const float *params, const Point *in, uint32_t n,
                    Point voxel_size, Point *out /* 8n */) {
    const v3 ovs = vdivs(from_point(voxel_size), (float) 2);
#pragma omp parallel
    {
#pragma omp for schedule(dynamic, 64)
        for (int64_t id = 0; id < (int64_t) n; id++) {
            const v3 base = from_point(in[id]);
            for (int i = 0; i < 2; i++)
                for (int j = 0; j < 2; j++)
                    for (int k = 0; k < 2; k++) {
                        v3 lower = vadd(base, vmul(V((float) i, (float) j, (float) k), ovs));
                        v3 upper = vadd(base, vmul(V((float) (i + 1), (float) (j + 1), (float) (k + 1)), ovs));
                        int is_border = 0;
                        int prev = obj_contains(scene, params, lower);
                        for (int c = 1; c < 8; c++) {
                            v3 q = V(c & 1 ? upper.x : lower.x, c & 2 ? upper.y : lower.y,
                                     c & 4 ? upper.z : lower.z);
                            if (prev != obj_contains(scene, params, q)) { is_border = 1; break; }
                        }
                        size_t n_id = (size_t) id * 8 + i * 4 + j * 2 + k;
                        out[n_id].x = is_border ? lower.x : INFINITY;
                        out[n_id].y = is_border ? lower.y : INFINITY;
                        out[n_id].z = is_border ? lower.z : INFINITY;
                    }
        }
        flush_evals();
    }
}

/* src/cuda/mod.rs:192-193: order-preserving retain of all-finite points */
size_t orc_retain_finite(Point *buf, size_t n) {
    size_t o = 0;
    for (size_t i = 0; i < n; i++)
        if (isfinite(buf[i].x) && isfinite(buf[i].y) && isfinite(buf[i].z)) buf[o++] = buf[i];
    return o;
}

/* ---- A4: marching_cubes.cu:8-43 ---------------------------------------------------- */
typedef struct { v3 vertices[8]; float values[8]; } McCube;

static v3 edge_vertex(const McCube *cube, int i0, int i1) {   /* :13-16 */
    float v = 0.5f;
    v3 x = cube->vertices[i0], y = cube->vertices[i1];
    return vadd(vscale(x, 1.0f - v), vscale(y, v));              /* glm::mix */
}

static unsigned march_cube(const McCube *cube, Triangle *triangles) {   /* :18-43 */
    unsigned cube_index = 0;
    for (int i = 0; i < 8; i++) cube_index |= ((unsigned) (cube->values[i] <= 0.0f)) << i;
    uint64_t row = MC_TRI_PACKED[cube_index];
    unsigned triangle_count = 0;
    while ((row & 0xF) != 0xF) {
        Triangle t;
        memset(&t, 0, sizeof t);
        for (int j = 0; j < 3; j++) {
            int e = (int) (row & 0xF); row >>= 4;
            t.vertices[j].position = to_point(edge_vertex(cube, MC_EDGE_CORNERS[e] & 0xF, MC_EDGE_CORNERS[e] >> 4));
        }
        triangles[triangle_count++] = t;
    }
    return triangle_count;
}

/* ---- A7: compute_surface_triangles_from_voxel_field_by_sdf, compute_mesh_generation.cu:64-120
 * Writes the raw 5n layout; unused slots = { {NaN,NaN,NaN},{NaN,NaN,NaN} } in vertex 0 and
 * zeros in vertices 1..2 (brace elision at :117).  iter_hist (optional, 17 bins) collects the
 * closest_surface_point iteration counts.
 */
void orc_triangles_raw(int scene, const float *params, const Point *vox, uint32_t n,
                       Point voxel_size_p, Triangle *triangles /* 5n */, uint64_t *iter_hist) {
    const v3 voxel_size = from_point(voxel_size_p);
#pragma omp parallel
    {
        uint64_t hist[17];
        memset(hist, 0, sizeof hist);
#pragma omp for schedule(dynamic, 16)
        for (int64_t id = 0; id < (int64_t) n; id++) {
            const size_t triangle_start = 5 * (size_t) id, triangle_end = 5 * ((size_t) id + 1);
            const v3 base = from_point(vox[id]);
            McCube cube;
            for (int c = 0; c < 8; c++) {
                v3 v = base;
                v.x += ((c % 4) == 1 || (c % 4) == 2) ? voxel_size.x : 0.0f;
                v.y += (c % 4) >= 2 ? voxel_size.y : 0.0f;
                v.z += c >= 4 ? voxel_size.z : 0.0f;
                cube.vertices[c] = v;
                cube.values[c] = sdf_eval(scene, params, v);
            }
            unsigned triangle_count = march_cube(&cube, &triangles[triangle_start]);
            for (size_t i = triangle_start; i < triangle_start + triangle_count; i++) {
                v3 v0 = from_point(triangles[i].vertices[0].position);
                v3 v1 = from_point(triangles[i].vertices[1].position);
                v3 v2 = from_point(triangles[i].vertices[2].position);
                uint32_t it0, it1, it2;
                v0 = closest_surface_point(scene, params, v0, &it0);
                v1 = closest_surface_point(scene, params, v1, &it1);
                v2 = closest_surface_point(scene, params, v2, &it2);
                hist[it0 > 16 ? 16 : it0]++; hist[it1 > 16 ? 16 : it1]++; hist[it2 > 16 ? 16 : it2]++;
                v3 n0 = empirical_normal(scene, params, v0);
                v3 n1 = empirical_normal(scene, params, v1);
                v3 n2 = empirical_normal(scene, params, v2);
                const v3 triangle_normal = normalize3(cross3(vsub(v1, v0), vsub(v2, v0)));
                const v3 actual_normal =
                    empirical_normal(scene, params, vdivs(vadd(vadd(v0, v1), v2), 3.0f));
                const int change_orientation = dot3(triangle_normal, actual_normal) <= 0.0f;
                triangles[i].vertices[0].position = to_point(change_orientation ? v2 : v0);
                triangles[i].vertices[0].normal = to_point(change_orientation ? n2 : n0);
                triangles[i].vertices[1].position = to_point(v1);
                triangles[i].vertices[1].normal = to_point(n1);
                triangles[i].vertices[2].position = to_point(change_orientation ? v0 : v2);
                triangles[i].vertices[2].normal = to_point(change_orientation ? n0 : n2);
            }
            for (size_t i = triangle_start + triangle_count; i < triangle_end; i++) {
                memset(&triangles[i], 0, sizeof(Triangle));
                triangles[i].vertices[0].position.x = NAN; triangles[i].vertices[0].position.y = NAN;
                triangles[i].vertices[0].position.z = NAN; triangles[i].vertices[0].normal.x = NAN;
                triangles[i].vertices[0].normal.y = NAN; triangles[i].vertices[0].normal.z = NAN;
            }
        }
        if (iter_hist) {
            for (int b = 0; b < 17; b++) {
#pragma omp atomic
                iter_hist[b] += hist[b];
            }
        }
        flush_evals();
    }
}

/* ---- A8: host weld, src/cuda/mod.rs:263-296 -----------------------------------------
 * key = [(x * 10e4f32).round() as i64; 3]; first-seen vertex keeps position AND normal;
 * index = insertion rank.  Slots whose vertices[0].position.x is not finite are skipped.
 */
static int64_t f32_discrete(float x) {
    float r = roundf(x * 10e4f);                     /* f32 multiply, round half away */
    if (r != r) return 0;                            /* Rust `as i64`: NaN -> 0, saturating */
    if (r >= 9223372036854775807.0f) return INT64_MAX;
    if (r <= -9223372036854775808.0f) return INT64_MIN;
    return (int64_t) r;
}

typedef struct { int64_t k[3]; uint32_t val; uint32_t used; } WeldSlot;

static uint64_t mix64(uint64_t h) {
    h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull; h ^= h >> 33;
    return h;
}

/* Returns the welded vertex count; *n_tris = emitted triangle count.
 * pos/nrm capacity: 9 * nslots floats worst case; idx capacity 3 * nslots. */
size_t orc_weld(const Triangle *raw, size_t nslots, float *pos, float *nrm, uint32_t *idx,
                size_t *n_tris) {
    size_t cap = 64;
    while (cap < nslots * 6 + 16) cap <<= 1;
    WeldSlot *tab = (WeldSlot *) calloc(cap, sizeof(WeldSlot));
    size_t nv = 0, nt = 0;
    for (size_t t = 0; t < nslots; t++) {
        if (!isfinite(raw[t].vertices[0].position.x)) continue;
        for (int j = 0; j < 3; j++) {
            const Vertex *v = &raw[t].vertices[j];
            int64_t k0 = f32_discrete(v->position.x), k1 = f32_discrete(v->position.y),
                    k2 = f32_discrete(v->position.z);
            uint64_t h = mix64((uint64_t) k0 * 0x9E3779B97F4A7C15ull ^ mix64((uint64_t) k1) ^
                               (mix64((uint64_t) k2) << 1));
            size_t s = (size_t) h & (cap - 1);
            for (;;) {
                if (!tab[s].used) {
                    tab[s].used = 1; tab[s].k[0] = k0; tab[s].k[1] = k1; tab[s].k[2] = k2;
                    tab[s].val = (uint32_t) nv;
                    pos[3 * nv] = v->position.x; pos[3 * nv + 1] = v->position.y; pos[3 * nv + 2] = v->position.z;
                    nrm[3 * nv] = v->normal.x; nrm[3 * nv + 1] = v->normal.y; nrm[3 * nv + 2] = v->normal.z;
                    nv++;
                    break;
                }
                if (tab[s].k[0] == k0 && tab[s].k[1] == k1 && tab[s].k[2] == k2) break;
                s = (s + 1) & (cap - 1);
            }
            idx[3 * nt + j] = tab[s].val;
        }
        nt++;
    }
    free(tab);
    *n_tris = nt;
    return nv;
}

/* FNV-1a-64 as used for the SURVEY 8(c) table */
uint64_t orc_fnv1a64(const void *data, size_t n) {
    const unsigned char *p = (const unsigned char *) data;
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}

/* ---- conventional dense CPU marching cubes (sample -> classify -> emit midpoints) ----
 * No projection, no normals, no weld: the "plain CPU MC" comparison line the north star asks
 * for.  Returns the triangle count; N = cells per axis.
 */
uint64_t orc_dense_mc_count(int scene, const float *params, uint32_t N) {
    const float h = MESH_GENERATION_BB_SIZE / (float) N;
    const size_t S = (size_t) N + 1;
    uint8_t *sign = (uint8_t *) malloc(S * S * S);
    uint64_t tris = 0;
#pragma omp parallel
    {
#pragma omp for schedule(static)
        for (int64_t x = 0; x < (int64_t) S; x++)
            for (size_t y = 0; y < S; y++)
                for (size_t z = 0; z < S; z++) {
                    v3 p = V((float) x * h - 2.5f, (float) y * h - 2.5f, (float) z * h - 2.5f);
                    sign[((size_t) x * S + y) * S + z] = sdf_eval(scene, params, p) <= 0.0f;
                }
        uint64_t local = 0;
#pragma omp for schedule(static)
        for (int64_t x = 0; x < (int64_t) N; x++)
            for (size_t y = 0; y < N; y++)
                for (size_t z = 0; z < N; z++) {
                    unsigned ci = 0;
                    for (int c = 0; c < 8; c++) {
                        size_t cx = x + (((c % 4) == 1 || (c % 4) == 2) ? 1 : 0);
                        size_t cy = y + ((c % 4) >= 2 ? 1 : 0);
                        size_t cz = z + (c >= 4 ? 1 : 0);
                        ci |= (unsigned) sign[(cx * S + cy) * S + cz] << c;
                    }
                    local += MC_TRI_COUNT[ci];
                }
#pragma omp atomic
        tris += local;
        flush_evals();
    }
    free(sign);
    return tris;
}

/* ---- point-wise probes for the numerics tests (sd, normal, projection, orientation) ---- */
void orc_eval_points(int scene, const float *params, const float *pts, long n, float *sd, float *nrm,
                     float *proj, int *flip) {
    for (long i = 0; i < n; i++) {
        v3 p = V(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
        sd[i] = sdf_eval(scene, params, p);
        if (nrm) {
            v3 nn = empirical_normal(scene, params, p);
            nrm[3 * i] = nn.x; nrm[3 * i + 1] = nn.y; nrm[3 * i + 2] = nn.z;
        }
        if (proj) {
            v3 g = closest_surface_point(scene, params, p, NULL);
            proj[3 * i] = g.x; proj[3 * i + 1] = g.y; proj[3 * i + 2] = g.z;
        }
    }
    if (flip) {
        for (long i = 0; i + 2 < n; i += 3) {
            v3 v0 = V(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
            v3 v1 = V(pts[3 * i + 3], pts[3 * i + 4], pts[3 * i + 5]);
            v3 v2 = V(pts[3 * i + 6], pts[3 * i + 7], pts[3 * i + 8]);
            v3 tn = normalize3(cross3(vsub(v1, v0), vsub(v2, v0)));
            v3 an = empirical_normal(scene, params, vdivs(vadd(vadd(v0, v1), v2), 3.0f));
            flip[i / 3] = dot3(tn, an) <= 0.0f;
        }
    }
}

long long orc_weld_quant(float x) { return (long long) f32_discrete(x); }


/* ===========================================================================================
 * Row N4: the ray-march preview, cuda/modules/compute_render.cu:21-97 with
 * common.cu:13-215 (coordinate helpers, pixel cone), ray_marching.cu:15-49 (sphere tracing) and
 * color.cu:7-22 (ACES tone map).  One deliberate difference, stated in DESIGN.md: tan(fov / 2)
 * (common.cu:86, the same value for every pixel) is evaluated ONCE by the caller and passed in as
 * fov_fac, because libm's and the GPU's tanf are not bit-identical; everything else follows the
 * reference's operation order.  "parity unpinned" like the rest of this file (no reference outputs).
 * =========================================================================================== */
typedef struct { unsigned long long tick; float time; unsigned int render_texture_size[2]; float render_screen_size[2]; } GlobalsBuffer;
typedef struct { float position[3], forward[3], up[3], right[3]; float fov; } CameraBuffer;

#define RAY_MARCH_STEP_LIMIT 256
#define RAY_MARCH_DEPTH_LIMIT 500.0f
#define RAY_MARCH_COLLISION_DISTANCE 0.001f
#define SQRT_INV 0.7071067811865475f

static float sd_scene(int scene, const float *params, v3 p) {          /* compute_render.cu:3-19 */
    float sd = 3.40282347E+38f;
    sd = gmin(sd, sdf_eval(scene, params, p));
    sd = gmin(sd, sd_box_skeleton(p, V(0.0f, 0.0f, 0.0f), V(5.0f, 5.0f, 5.0f), 0.05f));
    return sd;
}

static v3 scene_normal(int scene, const float *params, v3 p) {          /* signed_distance.cu:179-202 on sd_scene */
#define F(dx, dy, dz) sd_scene(scene, params, vadd(p, V(dx, dy, dz)))
    float dx = (-F(2.0f * NORMAL_EPSILON, 0.0f, 0.0f) + 8.0f * F(NORMAL_EPSILON, 0.0f, 0.0f) -
                8.0f * F(-NORMAL_EPSILON, 0.0f, 0.0f) + F(-2.0f * NORMAL_EPSILON, 0.0f, 0.0f));
    float dy = (-F(0.0f, 2.0f * NORMAL_EPSILON, 0.0f) + 8.0f * F(0.0f, NORMAL_EPSILON, 0.0f) -
                8.0f * F(0.0f, -NORMAL_EPSILON, 0.0f) + F(0.0f, -2.0f * NORMAL_EPSILON, 0.0f));
    float dz = (-F(0.0f, 0.0f, 2.0f * NORMAL_EPSILON) + 8.0f * F(0.0f, 0.0f, NORMAL_EPSILON) -
                8.0f * F(0.0f, 0.0f, -NORMAL_EPSILON) + F(0.0f, 0.0f, -2.0f * NORMAL_EPSILON));
#undef F
    return normalize3(V(dx, dy, dz));
}

/* texture coordinate -> unit ray direction (common.cu:15-17, 73-92) */
static v3 pixel_dir(float px, float py, float tw, float th, const GlobalsBuffer *g, const CameraBuffer *c, float fov_fac) {
    const float nx = (px + 0.5f) / tw, ny = (py + 0.5f) / th;                       /* texture_to_ndc */
    const float cx = (2.0f * nx - 1.0f) * (tw / th), cy = 1.0f - 2.0f * ny;         /* ndc_to_camera(p, texture size) */
    const float sx = g->render_screen_size[0], sy = g->render_screen_size[1];
    const float gx = (float) g->render_texture_size[0], gy = (float) g->render_texture_size[1];
    const float width_factor = (sx / gx) * (gy / sy);
    const v3 fw = V(c->forward[0], c->forward[1], c->forward[2]), up = V(c->up[0], c->up[1], c->up[2]),
             rt = V(c->right[0], c->right[1], c->right[2]);
    return normalize3(vadd(vadd(fw, vscale(up, cy * fov_fac)), vscale(rt, cx * fov_fac * width_factor)));
}

static v3 aces(v3 hdr) {                                                 /* color.cu:7-22, column-major mat3x3 */
    const v3 v = V(0.59719f * hdr.x + 0.35458f * hdr.y + 0.04823f * hdr.z,
                   0.07600f * hdr.x + 0.90834f * hdr.y + 0.01566f * hdr.z,
                   0.02840f * hdr.x + 0.13383f * hdr.y + 0.83777f * hdr.z);
    const v3 a = V(v.x * (v.x + 0.0245786f) - 0.000090537f, v.y * (v.y + 0.0245786f) - 0.000090537f,
                   v.z * (v.z + 0.0245786f) - 0.000090537f);
    const v3 b = V(v.x * (0.983729f * v.x + 0.4329510f) + 0.238081f, v.y * (0.983729f * v.y + 0.4329510f) + 0.238081f,
                   v.z * (0.983729f * v.z + 0.4329510f) + 0.238081f);
    const v3 q = V(a.x / b.x, a.y / b.y, a.z / b.z);
    const v3 r = V(1.60475f * q.x + -0.53108f * q.y + -0.07367f * q.z,
                   -0.10208f * q.x + 1.10813f * q.y + -0.00605f * q.z,
                   -0.00327f * q.x + -0.07276f * q.y + 1.07602f * q.z);
    return V(gmin(gmax(r.x, 0.0f), 1.0f), gmin(gmax(r.y, 0.0f), 1.0f), gmin(gmax(r.z, 0.0f), 1.0f));
}

/* probe for the host numerics test: the literal 12-bar loop on the preview's bounding-cube frame (compute_render.cu:8-16) */
void orc_cube_frame(const float *pts, long n, float *out) {
    for (long i = 0; i < n; i++)
        out[i] = sd_box_skeleton(V(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]), V(0.0f, 0.0f, 0.0f), V(5.0f, 5.0f, 5.0f), 0.05f);
}

float orc_fov_fac(float fov) { return tanf(fov / 2); }                   /* common.cu:86 */

void orc_render(int scene, const float *params, const void *globals, const void *camera, float fov_fac,
                uint32_t width, uint32_t height, uint8_t *rgba /* width * height * 4 */) {
    const GlobalsBuffer *g = (const GlobalsBuffer *) globals;
    const CameraBuffer *c = (const CameraBuffer *) camera;
    const float tw = (float) width, th = (float) height;
    const v3 light = normalize3(V(1.0f, 1.0f, 1.0f));
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t y = 0; y < (int64_t) height; y++)
        for (uint32_t x = 0; x < width; x++) {
            const v3 dir = pixel_dir((float) x, (float) y, tw, th, g, c, fov_fac);
            /* get_pixel_cone_radius (common.cu:97-178) */
            const v3 b0 = pixel_dir((float) x - SQRT_INV, (float) y - SQRT_INV, tw, th, g, c, fov_fac);
            const v3 b1 = pixel_dir((float) x - SQRT_INV, (float) y + SQRT_INV, tw, th, g, c, fov_fac);
            const v3 b2 = pixel_dir((float) x + SQRT_INV, (float) y - SQRT_INV, tw, th, g, c, fov_fac);
            const v3 b3 = pixel_dir((float) x + SQRT_INV, (float) y + SQRT_INV, tw, th, g, c, fov_fac);
            const float cone = gmax(gmax(length3(vsub(dir, b0)), length3(vsub(dir, b1))),
                                    gmax(length3(vsub(dir, b2)), length3(vsub(dir, b3))));
            /* ray_march (ray_marching.cu:15-49) */
            v3 pos = V(c->position[0], c->position[1], c->position[2]);
            float depth = 0.0f;
            int outcome = 1;                                             /* StepLimit */
            for (int steps = 0; steps < RAY_MARCH_STEP_LIMIT; steps++) {
                const float cd = cone * depth;
                const float d = sd_scene(scene, params, pos);
                if (d <= cd + RAY_MARCH_COLLISION_DISTANCE) { outcome = 0; break; }
                depth += (d - cd);
                pos = vadd(pos, vscale(dir, d - cd));
                if (depth > RAY_MARCH_DEPTH_LIMIT) { outcome = 2; break; }
            }
            v3 color = V(0.0f, 0.0f, 0.0f);
            if (outcome == 0) {
                const v3 n = scene_normal(scene, params, pos);
                const float t = (dot3(n, light) + 1.0f) / 2.0f;
                const v3 c1 = V(19.0f / 255.0f, 9.0f / 255.0f, 130.0f / 255.0f), c2 = V(240.0f / 255.0f, 103.0f / 255.0f, 24.0f / 255.0f);
                color = vadd(vscale(c1, 1.0f - t), vscale(c2, t));       /* glm::mix */
            } else if (outcome == 1) {
                color = V(1.0f, 1.0f, 1.0f);
            }
            color = aces(color);
            uint8_t *o = rgba + 4 * ((size_t) y * width + x);
            o[0] = (uint8_t) (gmin(gmax(color.x, 0.0f), 1.0f) * 255.0f);
            o[1] = (uint8_t) (gmin(gmax(color.y, 0.0f), 1.0f) * 255.0f);
            o[2] = (uint8_t) (gmin(gmax(color.z, 0.0f), 1.0f) * 255.0f);
            o[3] = 0xFF;
        }
    flush_evals();
}
