// Issue rate of the int8 MFMA shapes of gfx950: 4 independent accumulator chains per wavefront, W wavefronts per SIMD, 256 CUs.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_i8_rate mfma_i8_rate.hip ; run on the MI355X.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define ITERS 2048

template <int SHAPE>
__global__ __launch_bounds__(256) void k(int *out, int seed)
{
    v4i a = { seed, seed + 1, seed + 2, seed + 3 }, b = { seed ^ 5, seed ^ 6, seed ^ 7, seed ^ 8 };
    if (SHAPE == 0) {
        v16i c0 = {}, c1 = {}, c2 = {}, c3 = {};
        for (int i = 0; i < ITERS; ++i) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
        }
        int s = 0;
        for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else {
        v4i c0 = {}, c1 = {}, c2 = {}, c3 = {};
        for (int i = 0; i < ITERS; ++i) {
            c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0, 0, 0);
        }
        int s = 0;
        for (int i = 0; i < 4; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}

int main()
{
    int *d; hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int shape = 0; shape < 2; ++shape)
        for (int wps = 1; wps <= 4; wps *= 2) {
            const int blocks = 256 * wps;                       // 4 waves per block = 1 per SIMD; wps blocks per CU
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (shape == 0) k<0><<<blocks, 256>>>(d, rep); else k<1><<<blocks, 256>>>(d, rep);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double n_mfma = (double)blocks * 4 * ITERS * 4, ops = n_mfma * (shape == 0 ? 65536.0 : 32768.0);
            printf("%-12s waves/SIMD %d: %.3f ms  %.1f TOP/s  %.1f cycles @2.4GHz per MFMA per SIMD\n", shape == 0 ? "32x32x32_i8" : "16x16x64_i8", wps, ms,
                   ops / ms / 1e9, ms * 1e-3 * 2.4e9 / (n_mfma / 1024.0));
        }
    return 0;
}
