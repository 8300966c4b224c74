// VALU issue-rate ceiling of gfx950 per opcode class, measured three ways so that the result does not hang on the nominal clock:
//   (a) wall time (HIP events)  -> wave64 instructions per second per SIMD,
//   (b) s_memtime (clock64) and s_memrealtime (wall_clock64) read inside the kernel -> ticks per instruction and the tick rates,
//   (c) the implied shader clock = clock64 ticks / wall seconds, if clock64 follows the shader clock on this part.
// MI355X_MICROARCH.md: a SIMD-32 issues a wave64 VALU instruction over 2 cycles (157.3 TFLOP/s FP32 = 256 CU x 4 SIMD x 32 lanes x
// 2 flop x 2.4 GHz).  Every kernel runs CHAINS independent dependency chains of ONE opcode per wavefront, W wavefronts per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_peak valu_peak.hip ; run on the MI355X: ./valu_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITERS 4096
#define CHAINS 8
struct Clk { unsigned long long c0, c1, w0, w1; };
#define DEFK(name, T, init, stmt)                                                        \
__global__ __launch_bounds__(256) void name(T *out, T seed, Clk *clk) {                  \
    T x[CHAINS]; T y = seed + (T)threadIdx.x, z = seed * (T)3 + (T)1;                    \
    for (int c = 0; c < CHAINS; ++c) x[c] = init;                                        \
    const unsigned long long c0 = clock64(), w0 = wall_clock64();                        \
    for (int i = 0; i < ITERS; ++i) {                                                    \
        _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) { stmt; }                     \
    }                                                                                    \
    const unsigned long long c1 = clock64(), w1 = wall_clock64();                        \
    T s = 0; for (int c = 0; c < CHAINS; ++c) s += x[c];                                 \
    out[blockIdx.x * 256 + threadIdx.x] = s;                                             \
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk->c0 = c0; clk->c1 = c1; clk->w0 = w0; clk->w1 = w1; } \
}
#define UI (uint32_t)(threadIdx.x * 7u + c + seed)
#define FI (float)(threadIdx.x * 7u + c) * 1e-3f + seed
#define DI (double)(threadIdx.x * 7u + c) * 1e-3 + seed
DEFK(k_add_f32, float, FI, asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[c]) : "v"(y)))
DEFK(k_fma_f32, float, FI, asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[c]) : "v"(y), "v"(z)))
DEFK(k_mul_f32, float, FI, asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[c]) : "v"(y)))
DEFK(k_pk_fma_f32, double, DI, asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x[c]) : "v"(y), "v"(z)))
DEFK(k_xor, uint32_t, UI, asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[c]) : "v"(y)))
DEFK(k_add_u32, uint32_t, UI, asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[c]) : "v"(y)))
DEFK(k_and, uint32_t, UI, asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[c]) : "v"(y)))
DEFK(k_lshl, uint32_t, UI, asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x[c])))
DEFK(k_min_u32, uint32_t, UI, asm volatile("v_min_u32 %0, %0, %1" : "+v"(x[c]) : "v"(y)))
DEFK(k_bcnt, uint32_t, UI, asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(x[c]) : "v"(y)))
DEFK(k_perm, uint32_t, UI, asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(y), "v"(z)))
DEFK(k_alignbyte, uint32_t, UI, asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(x[c]) : "v"(y)))
DEFK(k_dot2, uint32_t, UI, asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(x[c]) : "v"(y), "v"(z)))
DEFK(k_mad_i24, uint32_t, UI, asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(x[c]) : "v"(y), "v"(z)))
DEFK(k_mad_u32_u24, uint32_t, UI, asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(x[c]) : "v"(y), "v"(z)))
DEFK(k_mul_lo, uint32_t, UI, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[c]) : "v"(y)))
DEFK(k_add3, uint32_t, UI, asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(y), "v"(z)))
DEFK(k_lshl_add, uint32_t, UI, asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x[c]) : "v"(y)))
DEFK(k_bfe, uint32_t, UI, asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(x[c])))
DEFK(k_cvt_f32_i32, uint32_t, UI, asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(x[c])))
DEFK(k_sad_u8, uint32_t, UI, asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(x[c]) : "v"(y), "v"(z)))
DEFK(k_dpp_add, uint32_t, UI, asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[c])))
DEFK(k_fma_f64, double, DI, asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(x[c]) : "v"(y), "v"(z)))
DEFK(k_add_f64, double, DI, asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[c]) : "v"(y)))
DEFK(k_mul_f64, double, DI, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[c]) : "v"(y)))

template <typename K, typename T> static void run(const char *name, K kern, T seed, int waves_per_simd, double flop_per_lane)
{
    static void *buf = nullptr; static Clk *clk = nullptr;
    if (!buf) { hipMalloc(&buf, (size_t)256 * 8 * 256 * 8); hipMalloc((void **)&clk, sizeof(Clk)); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int ncu = 256; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    const int blocks = ncu * waves_per_simd;                   // 256-thread workgroups = one wavefront on each of the 4 SIMDs of a CU
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, (T *)buf, seed, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, (T *)buf, seed, clk); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    Clk h; hipMemcpy(&h, clk, sizeof(h), hipMemcpyDeviceToHost);
    const double instr_per_simd = (double)waves_per_simd * ITERS * CHAINS;
    const double rate = instr_per_simd / (ms * 1e-3) / 1e9;                        // G wave64-instructions / s / SIMD
    const double ticks_c = (double)(h.c1 - h.c0), ticks_w = (double)(h.w1 - h.w0);
    printf("%-14s W=%d  %8.3f ms  %6.3f Ginstr/s/SIMD  %5.2f cyc@2.4GHz  clock64 %6.2f ticks/instr  wall_clock64 %7.4f ticks/instr", name,
           waves_per_simd, ms, rate, 2.4 / rate, ticks_c / instr_per_simd, ticks_w / instr_per_simd);
    if (flop_per_lane > 0) printf("  -> %6.1f TFLOP/s chip", rate * 1e9 * 4 * ncu * 64 * flop_per_lane / 1e12);
    printf("\n");
}
int main()
{
    int clk_khz = 0, wall_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    printf("# device clock attribute %d MHz, wall clock rate %d kHz; %d chains per wavefront, %d iterations\n", clk_khz / 1000, wall_khz, CHAINS, ITERS);
    for (int W = 4; W <= 8; W += 4) {
        run("v_add_f32", k_add_f32, 1.0f, W, 1); run("v_mul_f32", k_mul_f32, 1.0f, W, 1); run("v_fma_f32", k_fma_f32, 1.0f, W, 2);
        run("v_pk_fma_f32", k_pk_fma_f32, 1.0, W, 4);
        run("v_xor_b32", k_xor, 1u, W, 0); run("v_add_u32", k_add_u32, 1u, W, 0); run("v_and_b32", k_and, 1u, W, 0); run("v_lshlrev_b32", k_lshl, 1u, W, 0);
        run("v_min_u32", k_min_u32, 1u, W, 0); run("v_bcnt_u32", k_bcnt, 1u, W, 0); run("v_perm_b32", k_perm, 1u, W, 0);
        run("v_alignbyte", k_alignbyte, 1u, W, 0); run("v_dot2_i32_i16", k_dot2, 1u, W, 0); run("v_mad_i32_i24", k_mad_i24, 1u, W, 0);
        run("v_mad_u32_u24", k_mad_u32_u24, 1u, W, 0); run("v_mul_lo_u32", k_mul_lo, 1u, W, 0); run("v_add3_u32", k_add3, 1u, W, 0);
        run("v_lshl_add_u32", k_lshl_add, 1u, W, 0); run("v_bfe_u32", k_bfe, 1u, W, 0); run("v_cvt_f32_i32", k_cvt_f32_i32, 1u, W, 0);
        run("v_sad_u8", k_sad_u8, 1u, W, 0); run("v_add_u32 dpp", k_dpp_add, 1u, W, 0);
        run("v_fma_f64", k_fma_f64, 1.0, W, 2); run("v_add_f64", k_add_f64, 1.0, W, 1); run("v_mul_f64", k_mul_f64, 1.0, W, 1);
    }
    return 0;
}
