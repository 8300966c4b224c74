// A1 -- BGR->gray and the 5x5 binomial pyramid (replaces Frame::InitFrame / CreateImagePyramid,
// src/Basic/Frame.cpp:22-40, i.e. cv::cvtColor(CV_BGR2GRAY) + cv::pyrDown).  Integer arithmetic,
// bit-exact with oracle/image.c.  HBM-bound: every source byte is read once, every result
// byte written once; 16-byte accesses per lane.
#include "ygz_internal.h"
#include <limits.h>

// gray = (1868 B + 9617 G + 4899 R + 8192) >> 14   [OpenCV 3.1 RGB2Gray<uchar>]
__device__ __forceinline__ uint32_t gray1(uint32_t b, uint32_t g, uint32_t r)
{
    return (1868u * b + 9617u * g + 4899u * r + 8192u) >> 14;
}

// 16 pixels per lane: 3 x 16-byte loads, 1 x 16-byte store.  npix % 16 == 0.
// pad != nullptr (needs w % 16 == 0): the same 16 pixels also go into the interior of the tracker's framed copy of level 0
// (klt.hip; 8-byte aligned there), so that no separate copy kernel has to re-read the level.
typedef uint32_t img_u32x4 __attribute__((ext_vector_type(4)));
typedef img_u32x4 __attribute__((aligned(4))) img_u32x4u;
typedef img_u32x4 __attribute__((aligned(1))) img_u32x4b;      // 16 bytes at any byte address
typedef uint32_t __attribute__((aligned(1))) img_u32b;
__global__ __launch_bounds__(256) void k_bgr2gray16(const uint8_t *__restrict__ bgr, uint8_t *__restrict__ gray,
                                                    unsigned npix, int slot_begin, uint8_t *__restrict__ pad, int width, int height)
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
    if (pad) {
        // the framed copy: the 16 pixels at (y, x .. x + 15) of the interior, their BORDER_REFLECT_101 images in the left / right frame columns
        // (only the first two and the last two lanes of a row hold pixels 1 .. 24 / w - 25 .. w - 2: byte-reversed dwords), and the same again in
        // the rows above / below the image that mirror row y (y in 1 .. 24 / h - 25 .. h - 2).  Needs w % 16 == 0, w >= 64, h >= 50.
        const unsigned p0 = i * 16u, y = p0 / (unsigned)width, x = p0 - y * (unsigned)width;
        const size_t pw = (size_t)KLT_PW(width), ph = (size_t)height + 2 * KLT_B;
        uint8_t *base = pad + slot * pw * ph;
        img_u32x4 v; v.x = out[0]; v.y = out[1]; v.z = out[2]; v.w = out[3];
        const uint32_t r0 = __builtin_amdgcn_perm(out[0], out[0], 0x00010203u), r1 = __builtin_amdgcn_perm(out[1], out[1], 0x00010203u),
                       r2 = __builtin_amdgcn_perm(out[2], out[2], 0x00010203u), r3 = __builtin_amdgcn_perm(out[3], out[3], 0x00010203u);
        img_u32x4 rv; rv.x = r3; rv.y = r2; rv.z = r1; rv.w = r0;                 // the 16 pixels in reverse order
        auto store_row = [&](int r) {
            uint8_t *row = base + (size_t)r * pw;
            *reinterpret_cast<img_u32x4u *>(row + x + KLT_B) = v;
            if (x == 0u) *reinterpret_cast<img_u32x4b *>(row + KLT_B - 15) = rv;                       // pixels 15 .. 0 -> columns 9 .. 24
            else if (x == 16u) {                                                                        // pixels 24 .. 16 -> columns 0 .. 8
                *reinterpret_cast<img_u32b *>(row + 5) = r0; *reinterpret_cast<img_u32b *>(row + 1) = r1; row[0] = (uint8_t)(out[2] & 255u);
            }
            if ((int)x == width - 16) *reinterpret_cast<img_u32x4b *>(row + KLT_B + width - 1) = rv;    // pixels w-1 .. w-16 -> columns 24+w-1 .. 24+w+14
            else if ((int)x == width - 32) {                                                            // pixels w-17 .. w-25 -> columns 24+w+15 .. 24+w+23
                *reinterpret_cast<img_u32b *>(row + KLT_B + width + 15) = r3; *reinterpret_cast<img_u32b *>(row + KLT_B + width + 19) = r2;
                row[KLT_B + width + 23] = (uint8_t)(out[1] >> 24);
            }
        };
        store_row((int)y + KLT_B);
        if (y >= 1u && y <= (unsigned)KLT_B) store_row(KLT_B - (int)y);
        if ((int)y >= height - 1 - KLT_B && (int)y <= height - 2) store_row(KLT_B + 2 * (height - 1) - (int)y);
    }
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
// pad != nullptr: every result byte also goes into the interior of the tracker's framed copy of the destination level.
__global__ __launch_bounds__(256) void k_pyr_down(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                  int sw, int sh, int dw, int dh, int slot_begin, uint8_t *__restrict__ pad, int frame_here)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[PD_SH][PD_SW];
    const size_t slot = (size_t)(slot_begin + blockIdx.z);
    const uint8_t *s = src + slot * (size_t)sw * sh;
    uint8_t *d = dst + slot * (size_t)dw * dh;
    const int dpw = KLT_PW(dw);
    uint8_t *pd = pad ? pad + slot * (size_t)dpw * (dh + 2 * KLT_B) + (size_t)KLT_B * dpw + KLT_B : nullptr;
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
                const uint8_t o8 = (uint8_t)((v + 128) >> 8);
                d[(size_t)oy * dw + ox] = o8;
                if (pd) {
                    // the framed copy: the interior pixel and, near the image's edges, its BORDER_REFLECT_101 images in the frame (frame_here:
                    // dw, dh > KLT_B + 1, so that one reflection reaches every frame pixel; smaller levels leave the frame to k_klt_frame)
                    pd[(size_t)oy * dpw + ox] = o8;
                    if (frame_here) {
                        // (on a small level a pixel mirrors into the left AND the right frame, or the top AND the bottom one)
                        const bool fl = ox >= 1 && ox <= KLT_B, fr = ox >= dw - 1 - KLT_B && ox <= dw - 2;
                        const bool ft = oy >= 1 && oy <= KLT_B, fb = oy >= dh - 1 - KLT_B && oy <= dh - 2;
                        if (fl | fr | ft | fb) {
                            const ptrdiff_t xl = -ox, xr = 2 * (dw - 1) - ox, yt = (ptrdiff_t)(-oy) * dpw, yb = (ptrdiff_t)(2 * (dh - 1) - oy) * dpw, y0 = (ptrdiff_t)oy * dpw;
                            if (fl) pd[y0 + xl] = o8;
                            if (fr) pd[y0 + xr] = o8;
                            if (ft) { pd[yt + ox] = o8; if (fl) pd[yt + xl] = o8; if (fr) pd[yt + xr] = o8; }
                            if (fb) { pd[yb + ox] = o8; if (fl) pd[yb + xl] = o8; if (fr) pd[yb + xr] = o8; }
                        }
                    }
                }
            }
        }
    }
}

// The BORDER_REFLECT_101 frame of the tracker's working images (what buildOpticalFlowPyramid makes with copyMakeBorder), for the levels
// of a range of slots in ONE launch: the interiors were written by the kernels above, a lane fills one dword of the frame
// (top and bottom bands, left and right columns of the middle rows; the dwords of all levels form one flat item space, blockIdx.y = slot).  With full != 0 the level-0
// interior is copied too (level 0 came from a gray upload, not from k_bgr2gray16).
struct FrameArgs {
    const uint8_t *img[YGZ_MAX_LEVELS]; uint8_t *pad[YGZ_MAX_LEVELS];
    int w[YGZ_MAX_LEVELS], h[YGZ_MAX_LEVELS];
    int first[YGZ_MAX_LEVELS], items[YGZ_MAX_LEVELS];      // first block of each level, its number of items (dwords of frame)
    int n_levels, slot_begin, full0;
};
typedef uint32_t __attribute__((aligned(1))) img_u32u;      // 4 adjacent bytes at any address: one (unaligned) global_load_dword
__global__ __launch_bounds__(256) void k_klt_frame(FrameArgs A)
{
    // a block belongs to ONE level (first[] counts blocks), so the level is wave-uniform
    int L = 0;
#pragma unroll
    for (int k = 1; k < YGZ_MAX_LEVELS; ++k) if (k < A.n_levels && (int)blockIdx.x >= A.first[k]) L = k;
    const int item = ((int)blockIdx.x - A.first[L]) * 256 + (int)threadIdx.x;
    if (item >= A.items[L]) return;
    const int w = A.w[L], h = A.h[L], pw = KLT_PW(w), ph = h + 2 * KLT_B, pw4 = pw >> 2;
    int y, dc;
    if (A.full0 && L == 0) { y = item / pw4; dc = item - y * pw4; }      // every dword of the framed level
    else {
        const int r0 = (KLT_B + w) >> 2, nr = pw4 - r0, band = KLT_B * pw4, mid = (KLT_B >> 2) + nr;
        if (item < band) { y = item / pw4; dc = item - y * pw4; }
        else if (item < 2 * band) { const int j = item - band; y = j / pw4; dc = j - y * pw4; y += KLT_B + h; }
        else {
            const int j = item - 2 * band;
            y = j / mid;
            const int c = j - y * mid;
            y += KLT_B;
            dc = c < (KLT_B >> 2) ? c : r0 + (c - (KLT_B >> 2));
        }
    }
    const size_t slot = (size_t)(A.slot_begin + (int)blockIdx.y);
    const uint8_t *row = A.img[L] + slot * (size_t)w * h + (size_t)reflect101(y - KLT_B, h) * w;
    const int x = 4 * dc - KLT_B;
    uint32_t v = 0;
    if (x >= 0 && x + 3 < w) v = *reinterpret_cast<const img_u32u *>(row + x);      // inside the row: the bands above and below the image
    else {
#pragma unroll
        for (int k = 0; k < 4; ++k) v |= (uint32_t)row[reflect101(x + k, w)] << (8 * k);
    }
    *reinterpret_cast<uint32_t *>(A.pad[L] + slot * (size_t)pw * ph + (size_t)y * pw + 4 * dc) = v;
}

int ygz_launch_gray_pyramid(ygz_hip_ctx *ctx, int slot_begin, int n_slots, int from_bgr, int up_to_level)
{
    int rc = ygz_ensure_levels(ctx, up_to_level);
    if (rc != YGZ_OK) return rc;
    const unsigned npix = (unsigned)ctx->lw[0] * (unsigned)ctx->lh[0];
    // the tracker's framed copies are written along with the levels once its buffers exist (first LK call): no copy kernel per step
    int fuse_levels = 0;                                    // the leading levels that have a framed buffer (the tracker allocates the levels it uses)
    while (fuse_levels < up_to_level && ctx->klt_pad[fuse_levels]) ++fuse_levels;
    const bool fuse = fuse_levels > 0;
    // level 0 comes framed out of k_bgr2gray16 when its 16-pixel groups line up with the rows and the frame is one reflection away;
    // level L >= 1 out of k_pyr_down when both its sides exceed the frame (one reflection reaches every frame pixel); k_klt_frame does the rest
    // (small levels; level 0 of a gray upload: interior + frame)
    const bool l0_framed = fuse && from_bgr && (npix & 15u) == 0 && (ctx->lw[0] & 15) == 0 && ctx->lw[0] >= 64 && ctx->lh[0] >= 2 * KLT_B + 2;
    bool framed[YGZ_MAX_LEVELS];
    for (int L = 0; L < YGZ_MAX_LEVELS; ++L) framed[L] = L == 0 ? l0_framed : (L < fuse_levels && ctx->lw[L] > KLT_B + 1 && ctx->lh[L] > KLT_B + 1);
    if (from_bgr) {
        if ((npix & 15u) == 0)
            YGZ_LAUNCH(ctx, KID_BGR2GRAY, k_bgr2gray16, dim3(ygz_div_up((int)(npix / 16), 256), n_slots), dim3(256),
                               ctx->bgr, ctx->lvl[0], npix, slot_begin, l0_framed ? ctx->klt_pad[0] : (uint8_t *)nullptr, ctx->lw[0], ctx->lh[0]);
        else
            YGZ_LAUNCH(ctx, KID_BGR2GRAY, k_bgr2gray1, dim3(ygz_div_up((int)npix, 256), n_slots), dim3(256),
                               ctx->bgr, ctx->lvl[0], npix, slot_begin);
    }
    for (int L = 1; L < up_to_level; ++L) {
        const int sw = ctx->lw[L - 1], sh = ctx->lh[L - 1], dw = ctx->lw[L], dh = ctx->lh[L];
        YGZ_LAUNCH(ctx, KID_PYR_DOWN, k_pyr_down, dim3(ygz_div_up(dw, PD_TW), ygz_div_up(dh, PD_TH), n_slots), dim3(256), ctx->lvl[L - 1], ctx->lvl[L], sw, sh, dw, dh, slot_begin,
                   L < fuse_levels ? ctx->klt_pad[L] : (uint8_t *)nullptr, framed[L] ? 1 : 0);
    }
    if (fuse) {
        FrameArgs F;
        for (int L = 0; L < YGZ_MAX_LEVELS; ++L) { F.img[L] = nullptr; F.pad[L] = nullptr; F.w[L] = F.h[L] = 0; F.first[L] = 0; F.items[L] = 0; }
        int total = 0;                                  // blocks
        for (int L = 0; L < fuse_levels; ++L) {
            const int w = ctx->lw[L], h = ctx->lh[L], pw4 = KLT_PW(w) >> 2;
            F.img[L] = ctx->lvl[L]; F.pad[L] = ctx->klt_pad[L]; F.w[L] = w; F.h[L] = h;
            F.first[L] = total;
            // level 0 that k_bgr2gray16 did not write (gray upload, odd width, small frame): interior and frame; a small level L >= 1: its frame
            F.items[L] = framed[L] ? 0 : (L == 0 ? pw4 * (h + 2 * KLT_B) : 2 * KLT_B * pw4 + h * ((KLT_B >> 2) + pw4 - ((KLT_B + w) >> 2)));
            total += ygz_div_up(F.items[L], 256);
        }
        F.n_levels = fuse_levels; F.slot_begin = slot_begin; F.full0 = 1;
        if (total > 0) YGZ_LAUNCH(ctx, KID_KLT_PAD, k_klt_frame, dim3(total, n_slots), dim3(256), F);
    }
    if ((int)ctx->pad_levels.size() >= slot_begin + n_slots)
        for (int s = slot_begin; s < slot_begin + n_slots; ++s) ctx->pad_levels[s] = (uint8_t)fuse_levels;
    YGZ_HIPCHK(ctx, hipGetLastError());
    return YGZ_OK;
}
