// VALU issue-rate micro-benchmark for gfx950: cycles per wave64 instruction of a few opcodes used on the hot path.
// Each kernel runs 8 independent dependency chains of one opcode, 4 waves per SIMD, and reports shader cycles / instruction / wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITERS 2048
#define CHAINS 8
#define DEFK(name, stmt)                                                                 \
__global__ __launch_bounds__(256) void name(uint32_t *out, uint32_t seed) {              \
    uint32_t x[CHAINS]; uint32_t y = seed + threadIdx.x, z = seed * 3u + 1u;             \
    for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x * 7u + c + seed;                 \
    for (int i = 0; i < ITERS; ++i) {                                                    \
        _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) { stmt; }                     \
    }                                                                                    \
    uint32_t s = 0; for (int c = 0; c < CHAINS; ++c) s += x[c];                          \
    out[blockIdx.x * 256 + threadIdx.x] = s;                                             \
}
DEFK(k_xor,   asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[c]) : "v"(y)))
DEFK(k_bcnt,  asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(x[c]) : "v"(y)))
DEFK(k_perm,  asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(y), "v"(z)))
DEFK(k_align, asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(x[c]) : "v"(y)))
DEFK(k_dot2,  asm volatile("v_dot2c_i32_i16 %0, %1, %2" : "+v"(x[c]) : "v"(y), "v"(z)))
DEFK(k_dot2v3, asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(x[c]) : "v"(y), "v"(z)))
DEFK(k_mad24, asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(x[c]) : "v"(y), "v"(z)))
DEFK(k_mul_lo, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[c]) : "v"(y)))
DEFK(k_add3,  asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(y), "v"(z)))
DEFK(k_min3,  asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(y), "v"(z)))
DEFK(k_fma32, asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[c]) : "v"(y), "v"(z)))
DEFK(k_rdlane, { uint32_t s_; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s_) : "v"(x[c])); asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[c]) : "s"(s_)); })
DEFK(k_dpp,   asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[c])))
__global__ __launch_bounds__(256) void k_fma64(double *out, double seed) {
    double x[CHAINS]; const double y = seed + threadIdx.x * 1e-9, z = seed * 0.5;
    for (int c = 0; c < CHAINS; ++c) x[c] = c + seed;
    for (int i = 0; i < ITERS; ++i) {
        _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(x[c]) : "v"(y), "v"(z));
    }
    double s = 0; for (int c = 0; c < CHAINS; ++c) s += x[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename K, typename P, typename S> static void run(const char *name, K kern, P *buf, S seed, int per_iter)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 4;                       // 4 workgroups of 4 waves per CU -> 4 waves per SIMD
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, buf, seed);
    hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, buf, seed); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const double instr_per_simd = 4.0 * ITERS * CHAINS * per_iter;       // 4 waves per SIMD
    printf("%-10s %8.3f ms  -> %.2f cycles per wave64 instruction (at %d MHz)\n", name, ms, ms * 1e-3 * clk_khz * 1e3 / instr_per_simd, clk_khz / 1000);
}
int main()
{
    uint32_t *buf; hipMalloc(&buf, 256 * 4 * 256 * 8);
    run("v_xor", k_xor, buf, 1u, 1); run("v_bcnt", k_bcnt, buf, 1u, 1); run("v_perm", k_perm, buf, 1u, 1); run("v_alignbyte", k_align, buf, 1u, 1);
    run("v_dot2c", k_dot2, buf, 1u, 1); run("v_dot2 vop3p", k_dot2v3, buf, 1u, 1); run("v_mad_i24", k_mad24, buf, 1u, 1); run("v_mul_lo", k_mul_lo, buf, 1u, 1);
    run("v_add3", k_add3, buf, 1u, 1); run("v_min3", k_min3, buf, 1u, 1); run("v_fma_f32", k_fma32, buf, 1u, 1);
    run("readlane+add", k_rdlane, buf, 1u, 2); run("v_add dpp", k_dpp, buf, 1u, 1);
    run("v_fma_f64", k_fma64, (double *)buf, 1.0, 1);
    return 0;
}
