// micro-benchmark: MACHINE-instruction issue rate of the gfx950 VALU, one instruction kind at a time, at 1..8 waves per SIMD.
// Every operation is one inline-asm instruction, so the compiler can neither pack two scalar operations into a v_pk_* nor
// contract or reorder anything: what is counted is what SQ_INSTS_VALU counts.  (valu_rate.hip, the round-2 ceiling, was
// written in C: hipcc's SLP vectoriser turned its independent mul/add chains into v_pk_mul_f32 / v_pk_add_f32, so its
// "2.55-2.85 cycles per instruction" were cycles per SOURCE operation -- about 4.2 cycles per machine instruction.)
// Build: hipcc --offload-arch=gfx950 -O3 valu_issue.hip -o valu_issue
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f2 __attribute__((ext_vector_type(2)));

enum Kind { K_ADD = 0, K_MUL, K_MIN, K_FMA, K_CMP_CND, K_PK_ADD, K_PK_MUL, K_PK_FMA, K_SQRT, K_RCP, K_MIX,
            K_SUB, K_MAX, K_MED3, K_MIN3, K_ADDU, K_MOV, K_ADD_ABS, K_MUL3, K_CMP_CND64 };
static const char *NAMES[] = {"v_add_f32", "v_mul_f32", "v_min_f32", "v_fma_f32", "v_cmp_lt_f32 vcc + v_cndmask_b32 vcc (2 instr)", "v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32",
                              "v_sqrt_f32", "v_rcp_f32", "mix: 3 add + 2 mul + min + cmp + cndmask",
                              "v_sub_f32", "v_max_f32", "v_med3_f32", "v_min3_f32", "v_add_u32", "v_mov_b32", "v_add_f32 |abs| (VOP3)", "v_mul_f32 d, a, b (3 registers)",
                              "v_cmp_lt_f32 sgpr pair + v_cndmask_b32 sgpr pair (2 instr, VOP3)"};

template <int KIND, int CHAINS>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *clk, int iters, float s) {
    f2 a[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) { a[c].x = threadIdx.x * 1e-3f + c; a[c].y = a[c].x + 0.5f; }
    f2 s2; s2.x = s; s2.y = s;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int rep = 0; rep < 4; rep++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                if (KIND == K_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[c].x) : "v"(s));
                if (KIND == K_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[c].x) : "v"(s));
                if (KIND == K_MIN) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[c].x) : "v"(s));
                if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[c].x) : "v"(s));
                if (KIND == K_CMP_CND) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[c].x) : "v"(s) : "vcc");
                if (KIND == K_CMP_CND64) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n\tv_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(a[c].x) : "v"(s) : "s20", "s21");
                if (KIND == K_SUB) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[c].x) : "v"(s));
                if (KIND == K_MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[c].x) : "v"(s));
                if (KIND == K_MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[c].x) : "v"(s), "v"(s2.y));
                if (KIND == K_MIN3) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[c].x) : "v"(s), "v"(s2.y));
                if (KIND == K_ADDU) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[c].x) : "v"(s));
                if (KIND == K_MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(a[c].x) : "v"(s));
                if (KIND == K_ADD_ABS) asm volatile("v_add_f32_e64 %0, |%0|, %1" : "+v"(a[c].x) : "v"(s));
                if (KIND == K_MUL3) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[c].y) : "v"(a[c].x), "v"(s));
                if (KIND == K_PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[c]) : "v"(s2));
                if (KIND == K_PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[c]) : "v"(s2));
                if (KIND == K_PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[c]) : "v"(s2));
                if (KIND == K_SQRT) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[c].x));
                if (KIND == K_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[c].x));
            }
        }
        if (KIND == K_MIX) {                              // 8 instructions per chain, the SDF's kinds in roughly its proportions
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                asm volatile("v_add_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_min_f32 %0, %0, %1\n\t"
                             "v_mul_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %1\n\tv_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc"
                             : "+v"(a[c].x) : "v"(s) : "vcc");
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float acc = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) acc += a[c].x + a[c].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int KIND, int CHAINS> void run(float *d, unsigned long long *dclk, int waves_per_simd, int iters) {
    const int grid = 256 * waves_per_simd;              // one 4-wave workgroup per CU and per wave-per-SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, CHAINS>), dim3(grid), dim3(256), 0, 0, d, dclk, iters, 0.999f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, CHAINS>), dim3(grid), dim3(256), 0, 0, d, dclk, iters, 0.999f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long h[2 * 256 * 8];
    hipMemcpy(h, dclk, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost);
    double ghz = 0; for (int b = 0; b < grid; b++) ghz += (double) h[2 * b] / (double) h[2 * b + 1] * 0.1; ghz /= grid;
    const double per_iter = KIND == K_MIX ? 8.0 * CHAINS : ((KIND == K_CMP_CND || KIND == K_CMP_CND64) ? 8.0 * CHAINS : 4.0 * CHAINS);
    const double winstr = (double) grid * 4 * iters * per_iter;
    printf("{\"instr\": \"%s\", \"chains\": %d, \"waves_per_simd\": %d, \"ms\": %.4f, \"machine_wave_instr_per_s\": %.4e, "
           "\"cycles_per_machine_instr_per_simd\": %.3f, \"clock_ghz\": %.3f}\n",
           NAMES[KIND], CHAINS, waves_per_simd, ms, winstr / (ms * 1e-3), ms * 1e-3 * ghz * 1e9 * 1024 / winstr, ghz);
    fflush(stdout);
}

template <int KIND> void sweep(float *d, unsigned long long *dclk, int iters) {
    for (int w : {1, 2, 5, 8}) run<KIND, 8>(d, dclk, w, iters);
}

int main() {
    float *d; unsigned long long *dclk;
    hipMalloc(&d, 256 * 8 * 256 * 4); hipMalloc(&dclk, 2 * 256 * 8 * 8);
    sweep<K_ADD>(d, dclk, 2000); sweep<K_MUL>(d, dclk, 2000); sweep<K_MIN>(d, dclk, 2000); sweep<K_FMA>(d, dclk, 2000);
    sweep<K_CMP_CND>(d, dclk, 1000); sweep<K_CMP_CND64>(d, dclk, 1000); sweep<K_SUB>(d, dclk, 2000); sweep<K_MAX>(d, dclk, 2000); sweep<K_MED3>(d, dclk, 2000);
    sweep<K_MIN3>(d, dclk, 2000); sweep<K_ADDU>(d, dclk, 2000); sweep<K_MOV>(d, dclk, 2000); sweep<K_ADD_ABS>(d, dclk, 2000); sweep<K_MUL3>(d, dclk, 2000);
    sweep<K_PK_ADD>(d, dclk, 2000); sweep<K_PK_MUL>(d, dclk, 2000); sweep<K_PK_FMA>(d, dclk, 2000);
    sweep<K_SQRT>(d, dclk, 1000); sweep<K_RCP>(d, dclk, 1000); sweep<K_MIX>(d, dclk, 1000);
    for (int w : {1, 2, 3, 5}) run<K_MIX, 1>(d, dclk, w, 8000);           // ONE dependent chain per wave: the latency-bound end
    return 0;
}
