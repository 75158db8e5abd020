// Exhaustive check + issue cost of candidate CORRECTLY ROUNDED f32 square roots / reciprocals on gfx950.
// The hardware's v_sqrt_f32 / v_rsq_f32 / v_rcp_f32 are 1-ulp approximations; the strict build needs the IEEE result.
// A candidate is usable only if it returns the bits of the reference (hipcc's correctly rounded sqrtf / division) for EVERY
// input of its domain -- 2^31 inputs is a fraction of a second on the GPU, so this is a proof by enumeration for this chip.
//   ref        hipcc's sqrtf (v_sqrt_f32 + two-sided fma residual test + denormal scaling)
//   A          sqrt_rn_normal of round 2: v_sqrt_f32, y -+ 1 ulp, two fma residuals, two compare/select pairs (9 instr)
//   B          x' = max(x, 2^-100); y = v_rsq_f32(x'); g = x' * y; h = 0.5 * y; e = fma(-g, g, x'); r = fma(e, h, g)   (6 instr)
//   C          B without the clamp (5 instr; x = 0 -> NaN)
//   D          B with h refined once: t = fma(-g, h, 0.5); h = fma(h, t, h)   (8 instr)
//   E          g = v_sqrt_f32(x); h = 0.5 * v_rsq_f32(x'); e = fma(-g, g, x); r = fma(e, h, g)
//   R          reciprocal: y = v_rcp_f32(b); e = fma(-b, y, 1); r = fma(e, y, y)  vs  1.0f / b
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off exact_sqrt.hip -o exact_sqrt
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#pragma clang fp contract(off)

__device__ __forceinline__ float sqrt_A(float x) {
    const float y = __builtin_amdgcn_sqrtf(x);
    const float ydn = __builtin_bit_cast(float, __builtin_bit_cast(int, y) - 1);
    const float yup = __builtin_bit_cast(float, __builtin_bit_cast(int, y) + 1);
    const float rdn = __builtin_fmaf(-ydn, y, x);
    const float rup = __builtin_fmaf(-yup, y, x);
    float r = rdn <= 0.0f ? ydn : y;
    r = rup > 0.0f ? yup : r;
    return r;
}
#define TINY 7.888609052210118e-31f      /* 2^-100 */
__device__ __forceinline__ float sqrt_B(float x) {
    const float xc = __builtin_fmaxf(x, TINY);
    const float y = __builtin_amdgcn_rsqf(xc);
    const float g = xc * y, h = 0.5f * y;
    const float e = __builtin_fmaf(-g, g, xc);
    return __builtin_fmaf(e, h, g);
}
__device__ __forceinline__ float sqrt_C(float x) {
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y, h = 0.5f * y;
    const float e = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(e, h, g);
}
__device__ __forceinline__ float sqrt_D(float x) {
    const float xc = __builtin_fmaxf(x, TINY);
    const float y = __builtin_amdgcn_rsqf(xc);
    const float g = xc * y;
    float h = 0.5f * y;
    const float t = __builtin_fmaf(-g, h, 0.5f);
    h = __builtin_fmaf(h, t, h);
    const float e = __builtin_fmaf(-g, g, xc);
    return __builtin_fmaf(e, h, g);
}
__device__ __forceinline__ float sqrt_E(float x) {
    const float xc = __builtin_fmaxf(x, TINY);
    const float g = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rsqf(xc);
    const float e = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(e, h, g);
}
// F: A's structure with the one-sided information of the plain residual: e = x - y*y (exact for |y - sqrt x| <= 1 ulp);
// the sign of e says on which side the root lies, so only ONE neighbour has to be tested
__device__ __forceinline__ float sqrt_F(float x) {
    const float y = __builtin_amdgcn_sqrtf(x);
    const float e = __builtin_fmaf(-y, y, x);
    const int step = e < 0.0f ? -1 : 1;
    const float yn = __builtin_bit_cast(float, __builtin_bit_cast(int, y) + step);
    // root between y and yn: choose by the sign of x - y*yn  (= x - (midpoint^2 - ulp^2/4))
    const float rn = __builtin_fmaf(-yn, y, x);
    const bool take = e < 0.0f ? (rn <= 0.0f) : (rn > 0.0f);
    return take ? yn : y;
}
// B2: B with the clamp as an ADD: x + 2^-120 == x for every x >= 2^-95 (2^-120 is below half an ulp), 0 becomes 2^-120
#define TINY2 7.52316384526264e-37f      /* 2^-120 */
__device__ __forceinline__ float sqrt_B2(float x) {
    const float xc = x + TINY2;
    const float y = __builtin_amdgcn_rsqf(xc);
    const float g = xc * y, h = 0.5f * y;
    const float e = __builtin_fmaf(-g, g, xc);
    return __builtin_fmaf(e, h, g);
}
// a / 3 (the centroid of the orientation test): q0 = a * RN(1/3); r = fma(-3, q0, a); q = fma(r, RN(1/3), q0)
__device__ __forceinline__ float div3_M(float a) {
    const float c = 0.333333343267440796f;
    const float q0 = a * c;
    const float r = __builtin_fmaf(-3.0f, q0, a);
    return __builtin_fmaf(r, c, q0);
}
__device__ __forceinline__ float rcp_R(float b) {
    const float y = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, y, 1.0f);
    return __builtin_fmaf(e, y, y);
}

template <int V> __device__ __forceinline__ float cand(float x) {
    if (V == 0) return sqrt_A(x);
    if (V == 1) return sqrt_B(x);
    if (V == 2) return sqrt_C(x);
    if (V == 3) return sqrt_D(x);
    if (V == 4) return sqrt_E(x);
    if (V == 5) return sqrt_F(x);
    if (V == 6) return rcp_R(x);
    if (V == 7) return sqrt_B2(x);
    if (V == 8) return div3_M(x);
    return 0.0f;
}
template <int V> __device__ __forceinline__ float refv(float x) {
    if (V == 6) return 1.0f / x;
    if (V == 8) return x / 3.0f;
    return sqrtf(x);
}

// bits in [lo, hi): count mismatches against the reference, keep the first few
template <int V>
__global__ __launch_bounds__(256) void k_check(uint32_t lo, uint32_t hi, unsigned long long *nbad, uint32_t *bad, float *maxtiny) {
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
    for (uint64_t b = lo + (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; b < hi; b += stride) {
        const float x = __builtin_bit_cast(float, (uint32_t) b);
        const float c = cand<V>(x), r = refv<V>(x);
        if (__builtin_bit_cast(uint32_t, c) != __builtin_bit_cast(uint32_t, r)) {
            const unsigned long long n = atomicAdd(nbad, 1ULL);
            if (n < 16) { bad[3 * n] = (uint32_t) b; bad[3 * n + 1] = __builtin_bit_cast(uint32_t, c); bad[3 * n + 2] = __builtin_bit_cast(uint32_t, r); }
        }
    }
}
// inputs below the checked domain (zero, denormals, tiny normals): largest |result| -- must stay far below ulp(0.1)/2
template <int V>
__global__ __launch_bounds__(256) void k_tiny(uint32_t lo, uint32_t hi, uint32_t *maxbits, unsigned long long *nnan) {
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
    for (uint64_t b = lo + (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; b < hi; b += stride) {
        const float c = cand<V>(__builtin_bit_cast(float, (uint32_t) b));
        if (!(c == c)) atomicAdd(nnan, 1ULL);
        else atomicMax(maxbits, __builtin_bit_cast(uint32_t, fabsf(c)));
    }
}

// issue cost: 8 independent chains per lane, x <- cand(x) + 1 (keeps the values in range, adds one v_add per sqrt)
template <int V>
__global__ __launch_bounds__(256) void k_rate(float *out, int iters) {
    float a[8];
#pragma unroll
    for (int c = 0; c < 8; c++) a[c] = 1.5f + threadIdx.x * 1e-3f + c;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < 8; c++) a[c] = cand<V>(a[c]) + 1.0f;
    }
    float acc = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) acc += a[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int V> void run(const char *name, uint32_t lo, uint32_t hi, uint32_t tlo, uint32_t thi) {
    unsigned long long *nbad; uint32_t *bad; float *out; uint32_t *mx;
    hipMalloc(&nbad, 16); hipMalloc(&bad, 16 * 3 * 4); hipMalloc(&out, 256 * 5 * 256 * 4); hipMalloc(&mx, 4);
    hipMemset(nbad, 0, 16); hipMemset(bad, 0, 16 * 3 * 4); hipMemset(mx, 0, 4);
    hipLaunchKernelGGL((k_check<V>), dim3(256 * 8), dim3(256), 0, 0, lo, hi, nbad, bad, (float *) nullptr);
    if (thi > tlo) hipLaunchKernelGGL((k_tiny<V>), dim3(256 * 8), dim3(256), 0, 0, tlo, thi, mx, nbad + 1);
    unsigned long long hn[2]; uint32_t hb[48], hmx;
    hipMemcpy(hn, nbad, 16, hipMemcpyDeviceToHost); hipMemcpy(hb, bad, sizeof hb, hipMemcpyDeviceToHost); hipMemcpy(&hmx, mx, 4, hipMemcpyDeviceToHost);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, grid = 256 * 5;
    hipLaunchKernelGGL((k_rate<V>), dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_rate<V>), dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e-3 / ((double) grid * 4 * iters * 8) * 1024;        // seconds of one SIMD per wave-level call
    float mxf; memcpy(&mxf, &hmx, 4);
    printf("{\"variant\": \"%s\", \"checked_lo\": \"0x%08x\", \"checked_hi\": \"0x%08x\", \"mismatches\": %llu, \"below_domain_max_abs\": %.3e, \"below_domain_nan\": %llu, "
           "\"ns_per_wave_call_per_simd_incl_add\": %.3f, \"first_bad\": [", name, lo, hi, hn[0], mxf, hn[1], per * 1e9);
    for (unsigned long long i = 0; i < hn[0] && i < 8; i++) printf("%s[\"0x%08x\", \"0x%08x\", \"0x%08x\"]", i ? ", " : "", hb[3 * i], hb[3 * i + 1], hb[3 * i + 2]);
    printf("]}\n"); fflush(stdout);
    hipFree(nbad); hipFree(bad); hipFree(out); hipFree(mx);
}

int main() {
    // sqrt: every positive float from 2^-100 up to FLT_MAX (0x0d800000 = 2^-100, 0x7f800000 = +inf exclusive); below-domain: [0, 2^-100)
    const uint32_t LO = 0x0d800000u, HI = 0x7f800000u;
    run<0>("A_sqrt_two_sided", 0x00800000u, HI, 0, 0x00800000u);
    run<1>("B_rsq_markstein_clamped", LO, HI, 0, LO);
    run<2>("C_rsq_markstein", LO, HI, 0, LO);
    run<3>("D_rsq_markstein_refined_h", LO, HI, 0, LO);
    run<4>("E_sqrt_plus_rsq", LO, HI, 0, LO);
    run<5>("F_sqrt_one_sided", 0x00800000u, HI, 0, 0x00800000u);
    // reciprocal: every normal b of either sign whose reciprocal is normal: |b| in [2^-126, 2^126]
    run<6>("R_rcp_markstein_pos", 0x00800000u, 0x7e800001u, 0, 0);
    run<6>("R_rcp_markstein_neg", 0x80800000u, 0xfe800001u, 0, 0);
    // B2: exact from 2^-95 up (in [2^-96, 2^-95) the nudge is a rounding tie); below that only "small and not NaN" is needed
    run<7>("B2_rsq_markstein_add_clamp", 0x10000000u, HI, 0, 0x10000000u);
    // a / 3 for every finite a whose quotient is normal (|a| >= 2^-124), both signs
    run<8>("div3_markstein_pos", 0x01800000u, 0x7f800000u, 0, 0);
    run<8>("div3_markstein_neg", 0x81800000u, 0xff800000u, 0, 0);
    return 0;
}
