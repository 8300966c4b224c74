// A1 -- BGR->gray and the 5x5 binomial pyramid (replaces Frame::InitFrame / CreateImagePyramid,
// src/Basic/Frame.cpp:22-40, i.e. cv::cvtColor(CV_BGR2GRAY) + cv::pyrDown).  Integer arithmetic,
// bit-exact with oracle/image.c.  HBM-bound: every source byte is read once, every result
// byte written once; 16-byte accesses per lane.
#include "ygz_internal.h"

// gray = (1868 B + 9617 G + 4899 R + 8192) >> 14   [OpenCV 3.1 RGB2Gray<uchar>]
__device__ __forceinline__ uint32_t gray1(uint32_t b, uint32_t g, uint32_t r)
{
    return (1868u * b + 9617u * g + 4899u * r + 8192u) >> 14;
}

// 16 pixels per lane: 3 x 16-byte loads, 1 x 16-byte store.  npix % 16 == 0.
__global__ __launch_bounds__(256) void k_bgr2gray16(const uint8_t *__restrict__ bgr, uint8_t *__restrict__ gray,
                                                    unsigned npix, int slot_begin)
{
    const unsigned i = blockIdx.x * 256u + threadIdx.x;          // group of 16 pixels
    if (i * 16u >= npix) return;
    const size_t slot = (size_t)(slot_begin + blockIdx.y);
    const uint4 *src = reinterpret_cast<const uint4 *>(bgr + slot * npix * 3u) + (size_t)i * 3u;
    const uint4 a = src[0], b = src[1], c = src[2];
    const uint32_t w[12] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w };
    uint32_t out[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {        // 4 pixels = 12 bytes = 3 words
        const uint32_t w0 = w[3 * q], w1 = w[3 * q + 1], w2 = w[3 * q + 2];
        const uint32_t g0 = gray1(w0 & 255u, (w0 >> 8) & 255u, (w0 >> 16) & 255u);
        const uint32_t g1 = gray1(w0 >> 24, w1 & 255u, (w1 >> 8) & 255u);
        const uint32_t g2 = gray1((w1 >> 16) & 255u, w1 >> 24, w2 & 255u);
        const uint32_t g3 = gray1((w2 >> 8) & 255u, (w2 >> 16) & 255u, w2 >> 24);
        out[q] = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
    }
    reinterpret_cast<uint4 *>(gray + slot * npix)[i] = make_uint4(out[0], out[1], out[2], out[3]);
}

__global__ __launch_bounds__(256) void k_bgr2gray1(const uint8_t *__restrict__ bgr, uint8_t *__restrict__ gray,
                                                   unsigned npix, int slot_begin)
{
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= npix) return;
    const size_t slot = (size_t)(slot_begin + blockIdx.y);
    const uint8_t *s = bgr + (slot * npix + i) * 3u;
    gray[slot * npix + i] = (uint8_t)gray1(s[0], s[1], s[2]);
}

__device__ __forceinline__ int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = (i < 0) ? -i : 2 * (n - 1) - i;
    return i;
}

// cv::pyrDown 8u: separable [1 4 6 4 1], dst = (sum + 128) >> 8, BORDER_REFLECT_101.
// One block = 64 x 32 output pixels; the 160 x 67 source window is staged in LDS with 16-byte loads (rows are 16-byte aligned when
// sw % 16 == 0), borders reflected on the fly.
#define PD_TW 64
#define PD_TH 32
#define PD_RPT (PD_TH / 4)   // output rows per thread
#define PD_SW 160          // staged row: source x in [2*ox0 - 16, 2*ox0 + 144): ten 16-byte vectors, 16-byte aligned when sw % 16 == 0
#define PD_SH (2 * PD_TH + 3)   // source y in [2*oy0 - 2, 2*oy0 + 2*PD_TH + 1)
#define PD_X0 16           // staged column of source x = 2*ox0
__global__ __launch_bounds__(256) void k_pyr_down(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                  int sw, int sh, int dw, int dh, int slot_begin)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[PD_SH][PD_SW];
    const size_t slot = (size_t)(slot_begin + blockIdx.z);
    const uint8_t *s = src + slot * (size_t)sw * sh;
    uint8_t *d = dst + slot * (size_t)dw * dh;
    const int ox0 = blockIdx.x * PD_TW, oy0 = blockIdx.y * PD_TH;
    const int sx0 = 2 * ox0 - PD_X0, sy0 = 2 * oy0 - 2;
    const bool aligned = (sw & 15) == 0;
    // 16 bytes per lane and load (4-byte loads ran the kernel at a quarter of the HBM rate), and all of a thread's staging loads go
    // out before the first LDS store
    constexpr int PD_RW = PD_SW / 16, PD_TOT = PD_SH * PD_RW, PD_N = (PD_TOT + 255) / 256;
    uint4 stage[PD_N];
#pragma unroll
    for (int k = 0; k < PD_N; ++k) {
        const int i = threadIdx.x + 256 * k;
        const int r = i / PD_RW, c16 = (i % PD_RW) * 16;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (i < PD_TOT) {
            const int sy = reflect101(sy0 + r, sh), sx = sx0 + c16;
            const uint8_t *row = s + (size_t)sy * sw;
            if (aligned && sx >= 0 && sx + 15 < sw) v = *reinterpret_cast<const uint4 *>(row + sx);
            else {
                uint32_t w4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    w4[q] = (uint32_t)row[reflect101(sx + 4 * q, sw)] | ((uint32_t)row[reflect101(sx + 4 * q + 1, sw)] << 8) |
                            ((uint32_t)row[reflect101(sx + 4 * q + 2, sw)] << 16) | ((uint32_t)row[reflect101(sx + 4 * q + 3, sw)] << 24);
                v = make_uint4(w4[0], w4[1], w4[2], w4[3]);
            }
        }
        stage[k] = v;
    }
#pragma unroll
    for (int k = 0; k < PD_N; ++k) {
        const int i = threadIdx.x + 256 * k;
        if (i < PD_TOT) *reinterpret_cast<uint4 *>(&tile[i / PD_RW][(i % PD_RW) * 16]) = stage[k];
    }
    __syncthreads();
    // lane = output column, each thread PD_RPT consecutive output rows (they share source rows)
    const int tx = threadIdx.x & 63, ty0 = (threadIdx.x >> 6) * PD_RPT;
    const int ox = ox0 + tx;
    int hrow[2 * PD_RPT + 3];                       // horizontal sums of source rows 2*ty0-2 .. 2*ty0+2*PD_RPT
#pragma unroll
    for (int r = 0; r < 2 * PD_RPT + 3; ++r) {
        // the five taps start at source x = 2*ox - 2, byte 2*tx + PD_X0 - 2 of the staged row: two aligned LDS dwords, the first four taps
        // with their weights 1 4 6 4 as ONE v_dot4_u32_u8 (five byte reads and eight ALU operations per row before)
        const uint32_t *tw = reinterpret_cast<const uint32_t *>(&tile[2 * ty0 + r][(2 * tx + PD_X0 - 2) & ~3]);
        const uint32_t d0 = tw[0], d1 = tw[1], sh = (uint32_t)((2 * tx + PD_X0 - 2) & 3);
        const uint32_t v4 = __builtin_amdgcn_alignbyte(d1, d0, sh);
        hrow[r] = (int)__builtin_amdgcn_udot4(v4, 0x04060401u, (d1 >> (8 * sh)) & 255u, false);
    }
    if (ox < dw) {
#pragma unroll
        for (int k = 0; k < PD_RPT; ++k) {
            const int oy = oy0 + ty0 + k;
            if (oy < dh) {
                const int v = hrow[2 * k] + 4 * hrow[2 * k + 1] + 6 * hrow[2 * k + 2] + 4 * hrow[2 * k + 3] + hrow[2 * k + 4];
                d[(size_t)oy * dw + ox] = (uint8_t)((v + 128) >> 8);
            }
        }
    }
}

int ygz_launch_gray_pyramid(ygz_hip_ctx *ctx, int slot_begin, int n_slots, int from_bgr, int up_to_level)
{
    int rc = ygz_ensure_levels(ctx, up_to_level);
    if (rc != YGZ_OK) return rc;
    const unsigned npix = (unsigned)ctx->lw[0] * (unsigned)ctx->lh[0];
    if (from_bgr) {
        if ((npix & 15u) == 0)
            YGZ_LAUNCH(ctx, KID_BGR2GRAY, k_bgr2gray16, dim3(ygz_div_up((int)(npix / 16), 256), n_slots), dim3(256),
                               ctx->bgr, ctx->lvl[0], npix, slot_begin);
        else
            YGZ_LAUNCH(ctx, KID_BGR2GRAY, k_bgr2gray1, dim3(ygz_div_up((int)npix, 256), n_slots), dim3(256),
                               ctx->bgr, ctx->lvl[0], npix, slot_begin);
    }
    for (int L = 1; L < up_to_level; ++L) {
        const int sw = ctx->lw[L - 1], sh = ctx->lh[L - 1], dw = ctx->lw[L], dh = ctx->lh[L];
        YGZ_LAUNCH(ctx, KID_PYR_DOWN, k_pyr_down, dim3(ygz_div_up(dw, PD_TW), ygz_div_up(dh, PD_TH), n_slots), dim3(256), ctx->lvl[L - 1], ctx->lvl[L], sw, sh, dw, dh, slot_begin);
    }
    YGZ_HIPCHK(ctx, hipGetLastError());
    return YGZ_OK;
}
