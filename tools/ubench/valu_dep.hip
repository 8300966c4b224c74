// Dependent-issue micro-benchmark for gfx950: cycles per wave64 VALU instruction as a function of the number of independent
// dependency chains inside a wavefront (1, 2, 4, 8) and of the wavefronts per SIMD (1, 2, 4).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITERS 4096
template <int CHAINS, int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed)
{
    uint32_t x[CHAINS]; uint32_t y = seed + threadIdx.x;
    for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x * 7u + c + seed;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int u = 0; u < 8 / CHAINS; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (OP == 0) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[c]) : "v"(y));
                else asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(x[c]) : "v"(y));
            }
    }
    uint32_t s = 0; for (int c = 0; c < CHAINS; ++c) s += x[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CHAINS, int OP> static void run(uint32_t *buf, int waves_per_simd)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * waves_per_simd;                   // 256-thread workgroups: one wavefront on each SIMD of a CU
    hipLaunchKernelGGL((k<CHAINS, OP>), dim3(blocks), dim3(256), 0, 0, buf, 1u);
    hipEventRecord(e0); hipLaunchKernelGGL((k<CHAINS, OP>), dim3(blocks), dim3(256), 0, 0, buf, 1u); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)waves_per_simd * ITERS * 8;
    printf("%s chains %d waves/SIMD %d: %7.3f ms -> %.2f cycles per instruction per SIMD (2400 MHz)\n", OP ? "v_bcnt" : "v_xor ", CHAINS, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / instr_per_simd);
}
int main()
{
    uint32_t *buf; hipMalloc(&buf, 256 * 8 * 256 * 4);
    for (int w = 1; w <= 4; w *= 2) { run<1, 0>(buf, w); run<2, 0>(buf, w); run<4, 0>(buf, w); run<8, 0>(buf, w); }
    for (int w = 1; w <= 4; w *= 2) { run<1, 1>(buf, w); run<2, 1>(buf, w); run<4, 1>(buf, w); run<8, 1>(buf, w); }
    run<8, 0>(buf, 8); run<8, 1>(buf, 8);
    return 0;
}
