// L4 -- pyramidal Lucas-Kanade tracker.
// Replaces cv::calcOpticalFlowPyrLK as called by Tracker::TrackKLT (src/Algorithm/Tracker.cpp:92-98:
// 21x21 window, maxLevel 4, 30 iterations / eps 1e-3, OPTFLOW_USE_INITIAL_FLOW).  Arithmetic follows
// the frozen specification in oracle/klt.c (OpenCV generic path: int16 Scharr derivatives, 14-bit
// fixed-point bilinear weights, int16 patch <<5, float normal equations).
//
//  k_scharr   one lane per pixel: 3/10/3 Scharr of the previous image per level (reflect-101 inside
//             the image exactly as calcSharrDeriv), 2 x int16 per pixel, 4-byte stores.
//  k_klt      one wavefront per point, all pyramid levels inside one launch (points are independent,
//             so no per-level launch boundary): the 441-pixel window is spread over the 64 lanes
//             (7 pixels per lane); the previous-image patch and its derivatives (3 x int16 per pixel)
//             stay in LDS for the whole iteration loop; every iteration gathers the moving 22x22
//             window of the next image, forms the two mismatch sums and reduces them with wave
//             shuffles in a FIXED order.  The integer terms are identical to the oracle's; only the
//             float summation order differs (tree vs raster), well inside the 1e-5 track tolerance.
#include "ygz_internal.h"

#define KLT_MAXWIN 21
#define KLT_NPIX   (KLT_MAXWIN * KLT_MAXWIN)

__device__ __forceinline__ int refl101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = (i < 0) ? -i : 2 * (n - 1) - i;
    return i;
}

// calcSharrDeriv: dx = smooth_v(x+1) - smooth_v(x-1), dy = 3/10/3 smooth_h of (row+1 - row-1)
__global__ __launch_bounds__(256) void k_scharr(const uint8_t *__restrict__ img, int16_t *__restrict__ deriv, int w, int h)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const int y0 = y > 0 ? y - 1 : (h > 1 ? 1 : 0), y2 = y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0);
    const int xm = x > 0 ? x - 1 : (w > 1 ? 1 : 0), xp = x < w - 1 ? x + 1 : (w > 1 ? w - 2 : 0);
    const uint8_t *r0 = img + (size_t)y0 * w, *r1 = img + (size_t)y * w, *r2 = img + (size_t)y2 * w;
    // trow0 = (s0+s2)*3 + s1*10 ; trow1 = s2 - s0   (as int16 in the reference; values fit)
    const int t0m = (r0[xm] + r2[xm]) * 3 + r1[xm] * 10, t0p = (r0[xp] + r2[xp]) * 3 + r1[xp] * 10;
    const int t1m = r2[xm] - r0[xm], t1c = r2[x] - r0[x], t1p = r2[xp] - r0[xp];
    const int dx = (int16_t)(t0p - t0m);
    const int dy = (int16_t)((t1p + t1m) * 3 + t1c * 10);
    reinterpret_cast<uint32_t *>(deriv)[(size_t)y * w + x] = ((uint32_t)(uint16_t)dx) | ((uint32_t)(uint16_t)dy << 16);
}

struct KltArgs {
    const uint8_t *prev[YGZ_MAX_LEVELS], *next[YGZ_MAX_LEVELS];   // level images of the two slots
    const int16_t *deriv[YGZ_MAX_LEVELS];                          // Scharr of prev
    int w[YGZ_MAX_LEVELS], h[YGZ_MAX_LEVELS];
    int max_level, win, max_count, use_initial_flow;
    double epsilon;            // already squared
    float min_eig_thr;
    const float *prev_pts; float *next_pts; uint8_t *status; float *err; int n;
};

__device__ __forceinline__ float wave_sum_f(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = __fadd_rn(v, __shfl_xor(v, off));
    return v;
}

__device__ __forceinline__ int cv_round_f(float v) { return __float2int_rn(v); }
#define KLT_DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

__global__ __launch_bounds__(256) void k_klt(KltArgs A)
{
    __shared__ int16_t sI[4][KLT_NPIX + 7];
    __shared__ uint32_t sD[4][KLT_NPIX + 7];      // (ix | iy<<16)
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + wv;
    if (p >= A.n) return;                          // wave-uniform
    const int win = A.win, npix = win * win;
    const float half = (float)(win - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (float)(1 << 20);
    int16_t *IWin = sI[wv]; uint32_t *DWin = sD[wv];
    const float ppx = A.prev_pts[2 * p], ppy = A.prev_pts[2 * p + 1];
    float outx = A.use_initial_flow ? A.next_pts[2 * p] : ppx;
    float outy = A.use_initial_flow ? A.next_pts[2 * p + 1] : ppy;
    bool status = true;
    float errv = 0.f;

    for (int level = A.max_level; level >= 0; --level) {
        const int w = A.w[level], h = A.h[level];
        const uint8_t *I = A.prev[level], *J = A.next[level];
        const uint32_t *D = reinterpret_cast<const uint32_t *>(A.deriv[level]);
        const float s = (float)(1. / (double)(1 << level));
        float prevx = __fmul_rn(ppx, s), prevy = __fmul_rn(ppy, s);
        float nx, ny;
        if (level == A.max_level) { nx = __fmul_rn(outx, s); ny = __fmul_rn(outy, s); }
        else { nx = __fmul_rn(outx, 2.f); ny = __fmul_rn(outy, 2.f); }
        outx = nx; outy = ny;
        prevx = __fsub_rn(prevx, half); prevy = __fsub_rn(prevy, half);
        const int ipx = (int)floorf(prevx), ipy = (int)floorf(prevy);
        if (ipx < -win || ipx >= w || ipy < -win || ipy >= h) {
            if (level == 0) { status = false; errv = 0.f; }
            continue;
        }
        float a = __fsub_rn(prevx, (float)ipx), b = __fsub_rn(prevy, (float)ipy);
        int iw00 = cv_round_f(__fmul_rn(__fmul_rn(__fsub_rn(1.f, a), __fsub_rn(1.f, b)), 16384.f));
        int iw01 = cv_round_f(__fmul_rn(__fmul_rn(a, __fsub_rn(1.f, b)), 16384.f));
        int iw10 = cv_round_f(__fmul_rn(__fmul_rn(__fsub_rn(1.f, a), b), 16384.f));
        int iw11 = 16384 - iw00 - iw01 - iw10;
        float sA11 = 0.f, sA12 = 0.f, sA22 = 0.f;
        for (int i = lane; i < npix; i += 64) {
            const int yy = i / win, xx = i - yy * win;
            const int X = ipx + xx, Y = ipy + yy;
            const int X0 = refl101(X, w), X1 = refl101(X + 1, w), Y0 = refl101(Y, h), Y1 = refl101(Y + 1, h);
            const int ival = KLT_DESCALE((int)I[(size_t)Y0 * w + X0] * iw00 + (int)I[(size_t)Y0 * w + X1] * iw01 +
                                         (int)I[(size_t)Y1 * w + X0] * iw10 + (int)I[(size_t)Y1 * w + X1] * iw11, 9);
            const bool x0in = X >= 0 && X < w, x1in = X + 1 >= 0 && X + 1 < w, y0in = Y >= 0 && Y < h, y1in = Y + 1 >= 0 && Y + 1 < h;
            const uint32_t d00 = (x0in && y0in) ? D[(size_t)Y * w + X] : 0u, d01 = (x1in && y0in) ? D[(size_t)Y * w + X + 1] : 0u;
            const uint32_t d10 = (x0in && y1in) ? D[(size_t)(Y + 1) * w + X] : 0u, d11 = (x1in && y1in) ? D[(size_t)(Y + 1) * w + X + 1] : 0u;
            const int ixval = KLT_DESCALE((int)(int16_t)(d00 & 0xFFFF) * iw00 + (int)(int16_t)(d01 & 0xFFFF) * iw01 +
                                          (int)(int16_t)(d10 & 0xFFFF) * iw10 + (int)(int16_t)(d11 & 0xFFFF) * iw11, 14);
            const int iyval = KLT_DESCALE((int)(int16_t)(d00 >> 16) * iw00 + (int)(int16_t)(d01 >> 16) * iw01 +
                                          (int)(int16_t)(d10 >> 16) * iw10 + (int)(int16_t)(d11 >> 16) * iw11, 14);
            IWin[i] = (int16_t)ival;
            DWin[i] = ((uint32_t)(uint16_t)(int16_t)ixval) | ((uint32_t)(uint16_t)(int16_t)iyval << 16);
            sA11 = __fadd_rn(sA11, (float)(ixval * ixval));
            sA12 = __fadd_rn(sA12, (float)(ixval * iyval));
            sA22 = __fadd_rn(sA22, (float)(iyval * iyval));
        }
        const float A11 = __fmul_rn(wave_sum_f(sA11), FLT_SCALE), A12 = __fmul_rn(wave_sum_f(sA12), FLT_SCALE),
                    A22 = __fmul_rn(wave_sum_f(sA22), FLT_SCALE);
        float Dd = __fsub_rn(__fmul_rn(A11, A22), __fmul_rn(A12, A12));
        const float dif = __fsub_rn(A11, A22);
        const float minEig = __fdiv_rn(__fsub_rn(__fadd_rn(A22, A11),
                                                 ygz_sqrtf_cr(__fadd_rn(__fmul_rn(dif, dif), __fmul_rn(__fmul_rn(4.f, A12), A12)))),
                                       (float)(2 * win * win));
        if (minEig < A.min_eig_thr || Dd < 1.192092896e-07f) {
            if (level == 0) status = false;
            continue;
        }
        Dd = __fdiv_rn(1.f, Dd);
        nx = __fsub_rn(nx, half); ny = __fsub_rn(ny, half);
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < A.max_count; ++j) {
            const int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (inx < -win || inx >= w || iny < -win || iny >= h) {
                if (level == 0) status = false;
                break;
            }
            a = __fsub_rn(nx, (float)inx); b = __fsub_rn(ny, (float)iny);
            iw00 = cv_round_f(__fmul_rn(__fmul_rn(__fsub_rn(1.f, a), __fsub_rn(1.f, b)), 16384.f));
            iw01 = cv_round_f(__fmul_rn(__fmul_rn(a, __fsub_rn(1.f, b)), 16384.f));
            iw10 = cv_round_f(__fmul_rn(__fmul_rn(__fsub_rn(1.f, a), b), 16384.f));
            iw11 = 16384 - iw00 - iw01 - iw10;
            float sb1 = 0.f, sb2 = 0.f;
            for (int i = lane; i < npix; i += 64) {
                const int yy = i / win, xx = i - yy * win;
                const int X0 = refl101(inx + xx, w), X1 = refl101(inx + xx + 1, w), Y0 = refl101(iny + yy, h), Y1 = refl101(iny + yy + 1, h);
                const int diff = KLT_DESCALE((int)J[(size_t)Y0 * w + X0] * iw00 + (int)J[(size_t)Y0 * w + X1] * iw01 +
                                             (int)J[(size_t)Y1 * w + X0] * iw10 + (int)J[(size_t)Y1 * w + X1] * iw11, 9) - (int)IWin[i];
                const uint32_t d = DWin[i];
                sb1 = __fadd_rn(sb1, (float)(diff * (int)(int16_t)(d & 0xFFFF)));
                sb2 = __fadd_rn(sb2, (float)(diff * (int)(int16_t)(d >> 16)));
            }
            const float b1 = __fmul_rn(wave_sum_f(sb1), FLT_SCALE), b2 = __fmul_rn(wave_sum_f(sb2), FLT_SCALE);
            const float dx = __fmul_rn(__fsub_rn(__fmul_rn(A12, b2), __fmul_rn(A22, b1)), Dd);
            const float dy = __fmul_rn(__fsub_rn(__fmul_rn(A12, b1), __fmul_rn(A11, b2)), Dd);
            nx = __fadd_rn(nx, dx); ny = __fadd_rn(ny, dy);
            outx = __fadd_rn(nx, half); outy = __fadd_rn(ny, half);
            if ((double)dx * (double)dx + (double)dy * (double)dy <= A.epsilon) break;
            if (j > 0 && (double)fabsf(__fadd_rn(dx, pdx)) < 0.01 && (double)fabsf(__fadd_rn(dy, pdy)) < 0.01) {
                outx = __fsub_rn(outx, __fmul_rn(dx, 0.5f)); outy = __fsub_rn(outy, __fmul_rn(dy, 0.5f));
                break;
            }
            pdx = dx; pdy = dy;
        }
        if (status && level == 0) {
            const float qx = __fsub_rn(outx, half), qy = __fsub_rn(outy, half);
            const int inx = (int)floorf(qx), iny = (int)floorf(qy);
            if (inx < -win || inx >= w || iny < -win || iny >= h) { status = false; continue; }
            const float aa = __fsub_rn(qx, (float)inx), bb = __fsub_rn(qy, (float)iny);
            iw00 = cv_round_f(__fmul_rn(__fmul_rn(__fsub_rn(1.f, aa), __fsub_rn(1.f, bb)), 16384.f));
            iw01 = cv_round_f(__fmul_rn(__fmul_rn(aa, __fsub_rn(1.f, bb)), 16384.f));
            iw10 = cv_round_f(__fmul_rn(__fmul_rn(__fsub_rn(1.f, aa), bb), 16384.f));
            iw11 = 16384 - iw00 - iw01 - iw10;
            float se = 0.f;
            for (int i = lane; i < npix; i += 64) {
                const int yy = i / win, xx = i - yy * win;
                const int X0 = refl101(inx + xx, w), X1 = refl101(inx + xx + 1, w), Y0 = refl101(iny + yy, h), Y1 = refl101(iny + yy + 1, h);
                const int diff = KLT_DESCALE((int)J[(size_t)Y0 * w + X0] * iw00 + (int)J[(size_t)Y0 * w + X1] * iw01 +
                                             (int)J[(size_t)Y1 * w + X0] * iw10 + (int)J[(size_t)Y1 * w + X1] * iw11, 9) - (int)IWin[i];
                se = __fadd_rn(se, fabsf((float)diff));
            }
            errv = __fdiv_rn(__fmul_rn(wave_sum_f(se), 1.f), (float)(32 * win * win));
        }
    }
    if (lane == 0) {
        A.next_pts[2 * p] = outx; A.next_pts[2 * p + 1] = outy;
        A.status[p] = (uint8_t)status; A.err[p] = errv;
    }
}

extern "C" {

void ygz_hip_default_klt_params(ygz_klt_params *p)
{
    p->win = 21; p->max_level = 4; p->max_iter = 30;      // Tracker.h:25-26, Tracker.cpp:97
    p->eps = 0.001; p->min_eig_threshold = 1e-4;          // Tracker.h:27, OpenCV default
    p->use_initial_flow = 1;
}

int ygz_hip_klt_track(ygz_hip_ctx *ctx, int prev_slot, int cur_slot, const float *prev_pts, float *next_pts, int n,
                      const ygz_klt_params *prm, uint8_t *status, float *err)
{
    if (!ctx || !prm || n < 0 || prev_slot < 0 || prev_slot >= ctx->prm.max_frames || cur_slot < 0 || cur_slot >= ctx->prm.max_frames)
        return YGZ_E_INVALID;
    if (prm->win < 3 || prm->win > KLT_MAXWIN || prm->max_level < 0 || prm->max_level >= YGZ_MAX_LEVELS) return YGZ_E_INVALID;
    if (n == 0) return YGZ_OK;
    if (!prev_pts || !next_pts || !status) return YGZ_E_INVALID;
    if (!ctx->pyr_valid[prev_slot] || !ctx->pyr_valid[cur_slot]) return YGZ_E_STATE;
    // buildOpticalFlowPyramid: stop when the next level would not exceed the window
    int max_level = prm->max_level, sw = ctx->lw[0], sh = ctx->lh[0];
    for (int level = 0; level <= prm->max_level; ++level) {
        sw = (sw + 1) / 2; sh = (sh + 1) / 2;
        if (sw <= prm->win || sh <= prm->win) { max_level = level; break; }
    }
    int rc = ygz_ensure_levels(ctx, max_level + 1);
    if (rc != YGZ_OK) return rc;
    // levels beyond the frame pyramid (same cv::pyrDown) for both slots
    if (max_level + 1 > ctx->prm.pyramid_levels) {
        if ((rc = ygz_launch_gray_pyramid(ctx, prev_slot, 1, 0, max_level + 1)) != YGZ_OK) return rc;
        if (cur_slot != prev_slot && (rc = ygz_launch_gray_pyramid(ctx, cur_slot, 1, 0, max_level + 1)) != YGZ_OK) return rc;
    }
    KltArgs A;
    for (int L = 0; L <= max_level; ++L) {
        const size_t npix = (size_t)ctx->lw[L] * ctx->lh[L];
        if (!ctx->deriv[L]) YGZ_HIPCHK(ctx, hipMalloc((void **)&ctx->deriv[L], npix * 4 + 64));   // one slot's worth, reused
        A.prev[L] = ctx->lvl[L] + (size_t)prev_slot * npix;
        A.next[L] = ctx->lvl[L] + (size_t)cur_slot * npix;
        A.deriv[L] = ctx->deriv[L];
        A.w[L] = ctx->lw[L]; A.h[L] = ctx->lh[L];
        hipLaunchKernelGGL(k_scharr, dim3(ygz_div_up(A.w[L], 64), ygz_div_up(A.h[L], 4)), dim3(256), 0, ctx->stream,
                           A.prev[L], ctx->deriv[L], A.w[L], A.h[L]);
    }
    for (int L = max_level + 1; L < YGZ_MAX_LEVELS; ++L) { A.prev[L] = A.next[L] = nullptr; A.deriv[L] = nullptr; A.w[L] = A.h[L] = 0; }
    const size_t N = (size_t)n;
    uint8_t *buf = nullptr;
    rc = ygz_scratch(ctx, SCR_KLT_PTS, N * (8 + 8 + 4 + 1) + 64, (void **)&buf);
    if (rc != YGZ_OK) return rc;
    float *d_prev = (float *)buf, *d_next = d_prev + 2 * N, *d_err = d_next + 2 * N; uint8_t *d_st = (uint8_t *)(d_err + N);
    YGZ_HIPCHK(ctx, hipMemcpyAsync(d_prev, prev_pts, N * 8, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(d_next, next_pts, N * 8, hipMemcpyHostToDevice, ctx->stream));
    A.max_level = max_level; A.win = prm->win;
    A.max_count = prm->max_iter < 0 ? 0 : (prm->max_iter > 100 ? 100 : prm->max_iter);
    double eps = prm->eps < 0 ? 0 : (prm->eps > 10 ? 10 : prm->eps);
    A.epsilon = eps * eps;
    A.min_eig_thr = (float)prm->min_eig_threshold;
    A.use_initial_flow = prm->use_initial_flow;
    A.prev_pts = d_prev; A.next_pts = d_next; A.status = d_st; A.err = d_err; A.n = n;
    hipLaunchKernelGGL(k_klt, dim3(ygz_div_up(n, 4)), dim3(256), 0, ctx->stream, A);
    YGZ_HIPCHK(ctx, hipGetLastError());
    YGZ_HIPCHK(ctx, hipMemcpyAsync(next_pts, d_next, N * 8, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(status, d_st, N, hipMemcpyDeviceToHost, ctx->stream));
    if (err) YGZ_HIPCHK(ctx, hipMemcpyAsync(err, d_err, N * 4, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return YGZ_OK;
}

}  // extern "C"
