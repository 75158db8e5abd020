// micro-benchmark: scalar f32 VALU ops vs packed (v_pk_*_f32) ops, same number of flops, 4 waves/SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#pragma clang fp contract(off)
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float s) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    if (MODE == 0) {            // 8 independent scalar chains: mul + add per step (contract off -> 2 instrs)
        for (int i = 0; i < iters; i++) {
            a0 = a0 * s + 1.0f; a1 = a1 * s + 1.0f; a2 = a2 * s + 1.0f; a3 = a3 * s + 1.0f;
            a4 = a4 * s + 1.0f; a5 = a5 * s + 1.0f; a6 = a6 * s + 1.0f; a7 = a7 * s + 1.0f;
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    } else if (MODE == 1) {     // 4 independent packed chains
        v2f b0 = {a0, a1}, b1 = {a2, a3}, b2 = {a4, a5}, b3 = {a6, a7}, ss = {s, s}, one = {1.0f, 1.0f};
        for (int i = 0; i < iters; i++) { b0 = b0 * ss + one; b1 = b1 * ss + one; b2 = b2 * ss + one; b3 = b3 * ss + one; }
        out[blockIdx.x * blockDim.x + threadIdx.x] = b0.x + b0.y + b1.x + b1.y + b2.x + b2.y + b3.x + b3.y;
    } else if (MODE == 2) {     // scalar: compare + select + min mix (what sd_obj is made of)
        for (int i = 0; i < iters; i++) {
            a0 = fminf(a0 * s, a1); a1 = (a1 < a2) ? a1 + 1.0f : a2; a2 = __builtin_amdgcn_fmed3f(a2, 0.0f, a3) + s; a3 = fmaxf(a3 - s, a0);
            a4 = fminf(a4 * s, a5); a5 = (a5 < a6) ? a5 + 1.0f : a6; a6 = __builtin_amdgcn_fmed3f(a6, 0.0f, a7) + s; a7 = fmaxf(a7 - s, a4);
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    } else {                    // sqrt chains (IEEE)
        for (int i = 0; i < iters; i++) { a0 = sqrtf(a0 + s); a1 = sqrtf(a1 + s); a2 = sqrtf(a2 + s); a3 = sqrtf(a3 + s); }
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
    }
}
template <int MODE> void run(const char *name, float *d, int iters, double ops_per_iter) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(256), 0, 0, d, iters, 0.999f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(256), 0, 0, d, iters, 0.999f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double waveinstr = 1024.0 * 4 * iters * ops_per_iter;          // wave-instructions issued
    printf("%-28s %.3f ms  -> %.2f cycles per wave-instr per SIMD (2.4 GHz, 1024 SIMDs)\n", name, ms,
           ms * 1e-3 * 2.4e9 * 1024 / waveinstr);
}
int main() {
    float *d; hipMalloc(&d, 1024 * 256 * 4);
    run<0>("scalar mul+add x8", d, 4000, 16);
    run<1>("packed mul+add x4", d, 4000, 8);
    run<2>("min/select/med3 mix", d, 4000, 18);
    run<3>("IEEE sqrt x4 (+add)", d, 1000, 4 * 14);
    return 0;
}
