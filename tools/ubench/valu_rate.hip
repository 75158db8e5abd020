// micro-benchmark: what f32 VALU issue rate does gfx950 sustain at 1..8 waves per SIMD?
// Independent mul/add/min chains (no FMA contraction, like the product build), 1024 SIMDs, 256-thread workgroups.
// Reports wave64 instructions per second (the unit of bench.py's roofline_step) and the shader clock the run held
// (s_memtime ticks over s_memrealtime's 100 MHz).  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off valu_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang fp contract(off)

template <int CHAINS>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *clk, int iters, float s) {
    float a[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) a[c] = threadIdx.x * 1e-3f + c;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) { a[c] = a[c] * s; a[c] = a[c] + 1.0f; a[c] = fminf(a[c], 3.0f); a[c] = a[c] - s; }   // 4 VALU each
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float acc = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) acc += a[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int CHAINS> void run(float *d, unsigned long long *dclk, int waves_per_simd, int iters) {
    const int grid = 256 * waves_per_simd;              // one 4-wave workgroup per CU and per wave-per-SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<CHAINS>, dim3(grid), dim3(256), 0, 0, d, dclk, iters, 0.999f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CHAINS>, dim3(grid), dim3(256), 0, 0, d, dclk, iters, 0.999f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long h[2 * 256 * 8];
    hipMemcpy(h, dclk, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost);
    double ghz = 0; for (int b = 0; b < grid; b++) ghz += (double) h[2 * b] / (double) h[2 * b + 1] * 0.1; ghz /= grid;
    const double winstr = (double) grid * 4 * iters * CHAINS * 4;
    printf("{\"chains\": %d, \"waves_per_simd\": %d, \"ms\": %.4f, \"wave_instr_per_s\": %.4e, \"cycles_per_instr_per_simd_at_measured_clock\": %.3f, \"clock_ghz\": %.3f}\n",
           CHAINS, waves_per_simd, ms, winstr / (ms * 1e-3), ms * 1e-3 * ghz * 1e9 * 1024 / winstr, ghz);
}

int main() {
    float *d; unsigned long long *dclk;
    hipMalloc(&d, 256 * 8 * 256 * 4); hipMalloc(&dclk, 2 * 256 * 8 * 8);
    for (int w : {1, 2, 4, 5, 8}) { run<8>(d, dclk, w, 4000); }
    for (int w : {1, 2, 4, 5, 8}) { run<2>(d, dclk, w, 16000); }
    return 0;
}
