// How much VALU work hides behind v_mfma_i32_32x32x32_i8 on one SIMD?  Per MFMA, NV independent VALU ops of one kind are placed in the
// same wavefront's stream (2 accumulator chains, W wavefronts per SIMD).  cycles are nominal 2.4 GHz per MFMA per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define ITERS 1024

template <int NV, int KIND>
__global__ __launch_bounds__(256) void k(int *out, int seed)
{
    v4i a = { seed, seed + 1, seed + 2, seed + 3 }, b = { seed ^ 5, seed ^ 6, seed ^ 7, seed ^ 8 };
    v16i c0 = {}, c1 = {};
    unsigned x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 2654435761u + i + seed;
    for (int it = 0; it < ITERS; ++it) {
        c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (KIND == 0) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[v & 7]) : "v"(x[(v + 1) & 7]));
            else if (KIND == 1) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x[v & 7]) : "v"(x[(v + 1) & 7]));
            else asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x[v & 7]) : "v"(x[(v + 1) & 7]));
        }
        c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (KIND == 0) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[v & 7]) : "v"(x[(v + 1) & 7]));
            else if (KIND == 1) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x[v & 7]) : "v"(x[(v + 1) & 7]));
            else asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x[v & 7]) : "v"(x[(v + 1) & 7]));
        }
    }
    int s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    for (int i = 0; i < 8; ++i) s += (int)x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, int KIND>
static void run(int *d, int wps, const char *name)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * wps;
    for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); k<NV, KIND><<<blocks, 256>>>(d, rep); hipEventRecord(e1); hipEventSynchronize(e1); }
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma = (double)blocks * 4 * ITERS * 2;
    printf("%-14s x%2d per MFMA, %d waves/SIMD: %7.3f ms  %6.1f cycles per MFMA per SIMD\n", name, NV, wps, ms, ms * 1e-3 * 2.4e9 / (n_mfma / 1024.0));
}

int main()
{
    int *d; (void)hipMalloc(&d, 4096 * 256 * 4);
    for (int wps = 1; wps <= 4; wps *= 2) {
        run<0, 0>(d, wps, "none");
        run<4, 0>(d, wps, "v_and"); run<8, 0>(d, wps, "v_and"); run<12, 0>(d, wps, "v_and"); run<16, 0>(d, wps, "v_and");
        run<4, 1>(d, wps, "v_min_u32"); run<8, 1>(d, wps, "v_min_u32"); run<12, 1>(d, wps, "v_min_u32");
        run<4, 2>(d, wps, "v_lshl_add_u32"); run<8, 2>(d, wps, "v_lshl_add_u32");
    }
    return 0;
}
