// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 (e2m1) operands on gfx950, for the 256-bit Hamming matcher:
//  (1) semantics: D[row][col] = sum over the 64 K-slots of A[row][k] * B[col][k] * 2^(sa - 127) * 2^(sb - 127) + C, with a lane (l31, half)
//      holding K-slots [32 half, 32 half + 32) of row / column l31 as 32 nibbles in 4 registers; C / D map as the 32x32 integer MFMAs;
//  (2) rate: issue cycles per instruction against v_mfma_i32_32x32x32_i8 (half the K per instruction).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f4_probe mfma_f4_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_sem(const uint32_t *a, const uint32_t *b, float *d, int sa, int sb, float c0)
{
    const int lane = threadIdx.x;
    v8i A = { (int)a[4 * lane], (int)a[4 * lane + 1], (int)a[4 * lane + 2], (int)a[4 * lane + 3], 0, 0, 0, 0 };
    v8i B = { (int)b[4 * lane], (int)b[4 * lane + 1], (int)b[4 * lane + 2], (int)b[4 * lane + 3], 0, 0, 0, 0 };
    v16f c;
    for (int i = 0; i < 16; ++i) c[i] = c0;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c, 4, 4, 0, sa, 0, sb);
    for (int i = 0; i < 16; ++i) d[16 * lane + i] = c[i];
}

template <int F4>
__global__ void k_rate(float *out, long long *cyc, int iters)
{
    const int lane = threadIdx.x & 63;
    v8i A = { lane, lane * 3, lane * 5, lane * 7, 0, 0, 0, 0 }, B = { lane * 11, lane * 13, lane * 17, lane * 19, 0, 0, 0, 0 };
    v4i A4 = { lane, lane * 3, lane * 5, lane * 7 }, B4 = { lane * 11, lane * 13, lane * 17, lane * 19 };
    v16f c0 = {}, c1 = {}, c2 = {}, c3 = {};
    v16i i0 = {}, i1 = {}, i2 = {}, i3 = {};
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (F4) {
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c0, 4, 4, 0, 133, 0, 127);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c1, 4, 4, 0, 133, 0, 127);
            c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c2, 4, 4, 0, 133, 0, 127);
            c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c3, 4, 4, 0, 133, 0, 127);
        } else {
            i0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A4, B4, i0, 0, 0, 0);
            i1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A4, B4, i1, 0, 0, 0);
            i2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A4, B4, i2, 0, 0, 0);
            i3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A4, B4, i3, 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i] + (float)(i0[i] + i1[i] + i2[i] + i3[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

static float f4val(int nib) { static const float t[8] = { 0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f }; const float v = t[nib & 7]; return (nib & 8) ? -v : v; }

int main()
{
    std::vector<uint32_t> ha(256), hb(256);
    srand(7);
    // A: every nibble +1 (0x2) or -1 (0xA); B: every nibble 0 or 1 (0x2)
    for (int i = 0; i < 256; ++i) { uint32_t x = 0, y = 0; for (int n = 0; n < 8; ++n) { x |= ((rand() & 1) ? 0xAu : 0x2u) << (4 * n); y |= ((rand() & 1) ? 0x2u : 0x0u) << (4 * n); } ha[i] = x; hb[i] = y; }
    uint32_t *da, *db; float *dd; long long *dc;
    CHK(hipMalloc(&da, 1024)); CHK(hipMalloc(&db, 1024)); CHK(hipMalloc(&dd, 64 * 16 * 4)); CHK(hipMalloc(&dc, 8));
    CHK(hipMemcpy(da, ha.data(), 1024, hipMemcpyHostToDevice)); CHK(hipMemcpy(db, hb.data(), 1024, hipMemcpyHostToDevice));
    for (int sa : { 127, 133 }) {
        k_sem<<<1, 64>>>(da, db, dd, sa, 127, 5.f);
        std::vector<float> hd(1024);
        CHK(hipMemcpy(hd.data(), dd, 4096, hipMemcpyDeviceToHost));
        // expected with the assumed maps
        int bad = 0; float first_got = 0, first_exp = 0;
        for (int lane = 0; lane < 64; ++lane) for (int r = 0; r < 16; ++r) {
            const int col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            float s = 0;
            for (int half = 0; half < 2; ++half) for (int q = 0; q < 4; ++q) for (int n = 0; n < 8; ++n)
                s += f4val((ha[4 * (row + 32 * half) + q] >> (4 * n)) & 15) * f4val((hb[4 * (col + 32 * half) + q] >> (4 * n)) & 15);
            const float e = s * (sa == 133 ? 64.f : 1.f) + 5.f;
            if (hd[16 * lane + r] != e) { if (!bad) { first_got = hd[16 * lane + r]; first_exp = e; } ++bad; }
        }
        printf("semantics scale_a = %d: %d of 1024 entries differ from the assumed layout (first: got %g, expected %g)\n", sa, bad, first_got, first_exp);
    }
    float *dout; CHK(hipMalloc(&dout, 1024 * 256 * 4));
    for (int f4 = 0; f4 < 2; ++f4) for (int waves : { 1, 2, 4 }) {
        const int iters = 4096;
        if (f4) k_rate<1><<<1024, 64 * waves>>>(dout, dc, iters); else k_rate<0><<<1024, 64 * waves>>>(dout, dc, iters);
        CHK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        CHK(hipEventRecord(e0));
        if (f4) k_rate<1><<<1024, 64 * waves>>>(dout, dc, iters); else k_rate<0><<<1024, 64 * waves>>>(dout, dc, iters);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        long long cyc; CHK(hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost));
        const double n_mfma = 1024.0 * waves * iters * 4, macs = n_mfma * 32 * 32 * (f4 ? 64 : 32);
        printf("%s, %d waves per block (1024 blocks): %.3f ms, %.1f shader cycles per MFMA per wave, %.2f P-op/s (2 x MAC)\n", f4 ? "f4 32x32x64" : "i8 32x32x32",
               waves, ms, (double)cyc / (iters * 4), 2 * macs / (ms * 1e-3) / 1e15);
    }
    return 0;
}
