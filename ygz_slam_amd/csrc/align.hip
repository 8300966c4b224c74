// L1-L2 -- 8x8 inverse-compositional patch alignment + the direct-projection search around it.
// Replaces cvutils::Align2D (src/Algorithm/CVUtils.cpp:186-318) and Matcher::FindDirectProjection
// (src/Algorithm/Matcher.cpp:356-466: GetWarpAffineMatrix, GetBestSearchLevel, WarpAffine).
//
// Mapping: one lane = one candidate patch, 64 candidates per workgroup.  The reference sums its 64
// float residual terms in raster order and branches on the result (convergence at 0.03 px,
// chi2 < 20000), so the per-patch arithmetic is kept in exactly that order (no FMA contraction,
// explicit _rn ops): results are bit-identical to oracle/align.c.  The affine-warped 10x10
// reference patch of every lane lives in LDS as pwb[k][lane] (byte k of 64 lanes is contiguous:
// conflict-free), gradients are re-derived from it instead of being stored; current-image
// pixels come straight from HBM/L2 (9x9 window per iteration, one new column per step).
#include "ygz_internal.h"
#include <cstring>
#include <vector>
#include "../../include/ygz_exp.h"
#include "se3_dev.h"

struct Cam { float fx, fy, cx, cy; };

__device__ __forceinline__ void pixel2camera_d(const Cam &c, const double px[2], double depth, double out[3])
{   // Basic/Camera.h:53-59
    out[0] = (px[0] - c.cx) * depth / c.fx;
    out[1] = (px[1] - c.cy) * depth / c.fy;
    out[2] = depth;
}
__device__ __forceinline__ void camera2pixel_d(const Cam &c, const double p[3], double out[2])
{   // Basic/Camera.h:46-51
    out[0] = c.fx * p[0] / p[2] + c.cx;
    out[1] = c.fy * p[1] / p[2] + c.cy;
}

__device__ __forceinline__ float cof3(const float m[9], int i, int j)
{
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return __fsub_rn(__fmul_rn(m[3 * i1 + j1], m[3 * i2 + j2]), __fmul_rn(m[3 * i1 + j2], m[3 * i2 + j1]));
}

// The neighbourhood of the start pixel is staged ONCE per candidate: STG_ROWS rows x 16 bytes of the current level image around
// floor(u0, v0) go into the lane's LDS column (dword e of the lane at stg[e * 64]: consecutive lanes hit consecutive banks), and the
// Gauss-Newton iterations read their 9 x 9 windows from there as long as the window stays inside (+-3 px in x, +-2 px in y; Align2D
// converges at 0.03 px or bails out).  Before, every iteration gathered 27 dwords per lane from L2 / HBM -- 64 different cache lines
// per load -- and the counters showed 10.7 x the algorithmic bytes (profiles/traffic.json, r01).  A window that leaves the staged
// block falls back to the global loads, so the arithmetic and the results are unchanged.
#define STG_ROWS 13
#define STG_DWORDS (STG_ROWS * 4)

// cvutils::Align2D core.  pwb: LDS, element k of this lane at pwb[k * 64]; stg: the lane's staging column (or nullptr).
__device__ bool align2d_core(const uint8_t *__restrict__ cur, int w, int h, const uint8_t *pwb, uint32_t *stg, int n_iter,
                             double *pu, double *pv, float *chi2_out)
{
    // gradient Hessian: J = (0.5*(I[x+1]-I[x-1]), 0.5*(I[y+1]-I[y-1]), 1); every partial sum is a
    // multiple of 0.25 below 2^22, hence exact in float in any order
    float H[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    for (int y = 0; y < 8; ++y)
        for (int x = 0; x < 8; ++x) {
            const int k = (y + 1) * 10 + (x + 1);
            const float jx = 0.5f * (float)((int)pwb[(k + 1) * 64] - (int)pwb[(k - 1) * 64]);
            const float jy = 0.5f * (float)((int)pwb[(k + 10) * 64] - (int)pwb[(k - 10) * 64]);
            H[0] = __fadd_rn(H[0], __fmul_rn(jx, jx)); H[1] = __fadd_rn(H[1], __fmul_rn(jx, jy)); H[2] = __fadd_rn(H[2], jx);
            H[4] = __fadd_rn(H[4], __fmul_rn(jy, jy)); H[5] = __fadd_rn(H[5], jy); H[8] = __fadd_rn(H[8], 1.0f);
        }
    H[3] = H[1]; H[6] = H[2]; H[7] = H[5];
    float Hinv[9];
    {
        const float c0 = cof3(H, 0, 0), c1 = cof3(H, 1, 0), c2 = cof3(H, 2, 0);
        const float det = __fadd_rn(__fadd_rn(__fmul_rn(c0, H[0]), __fmul_rn(c1, H[3])), __fmul_rn(c2, H[6]));
        const float invdet = __fdiv_rn(1.0f, det);
        Hinv[0] = __fmul_rn(c0, invdet); Hinv[1] = __fmul_rn(c1, invdet); Hinv[2] = __fmul_rn(c2, invdet);
        Hinv[3] = __fmul_rn(cof3(H, 0, 1), invdet); Hinv[4] = __fmul_rn(cof3(H, 1, 1), invdet); Hinv[5] = __fmul_rn(cof3(H, 2, 1), invdet);
        Hinv[6] = __fmul_rn(cof3(H, 0, 2), invdet); Hinv[7] = __fmul_rn(cof3(H, 1, 2), invdet); Hinv[8] = __fmul_rn(cof3(H, 2, 2), invdet);
    }
    float mean_diff = 0.f;
    float u = (float)*pu, v = (float)*pv;
    bool stg_ok = false;
    int sx0 = 0, sy0 = 0;
    if (stg && u == u && v == v && w >= 16 && h >= STG_ROWS) {
        const int u0 = (int)floorf(u), v0 = (int)floorf(v);
        if (!(u0 < 4 || v0 < 4 || u0 >= w - 4 || v0 >= h - 4)) {                 // otherwise the first iteration bails out anyway
            sx0 = min(max(u0 - 7, 0), w - 16); sy0 = min(max(v0 - 6, 0), h - STG_ROWS);
            const uint8_t *p0 = cur + (size_t)sy0 * w + sx0;
#pragma unroll
            for (int r = 0; r < STG_ROWS; ++r) {
                ygz_gptr32u q = (ygz_gptr32u)(p0 + (size_t)r * w);
#pragma unroll
                for (int d = 0; d < 4; ++d) stg[(r * 4 + d) * 64] = q[d];
            }
            stg_ok = true;
        }
    }
    const float min_update_squared = (float)(0.03 * 0.03);
    float chi2 = 0.f;
    bool converged = false;
    for (int iter = 0; iter < n_iter; ++iter) {
        chi2 = 0.f;
        if (u != u || v != v) break;
        const int u_r = (int)floorf(u), v_r = (int)floorf(v);
        if (u_r < 4 || v_r < 4 || u_r >= w - 4 || v_r >= h - 4) break;
        const float sx = __fsub_rn(u, (float)u_r), sy = __fsub_rn(v, (float)v_r);
        const float wTL = (float)((1.0 - (double)sx) * (1.0 - (double)sy));
        const float wTR = (float)((double)sx * (1.0 - (double)sy));
        const float wBL = (float)((1.0 - (double)sx) * (double)sy);
        const float wBR = __fmul_rn(sx, sy);
        float J0 = 0.f, J1 = 0.f, J2 = 0.f;
        // the 9 x 9 window rows v_r-4..v_r+4, columns u_r-4..u_r+4: three aligned dword loads + v_alignbyte per row instead of
        // 18 byte gathers (a wave-wide byte gather costs the address unit as much as a dword load)
        uint32_t wl[9], wh[9], w8[9];
        const int ox = u_r - 4 - sx0, oy = v_r - 4 - sy0;
        if (stg_ok && ox >= 0 && ox <= 7 && oy >= 0 && oy <= STG_ROWS - 9) {       // the window lies inside the staged block
            const uint32_t sh = (uint32_t)(ox & 3);
            const uint32_t *sp = stg + (oy * 4 + (ox >> 2)) * 64;
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                const uint32_t d0 = sp[(r * 4) * 64], d1 = sp[(r * 4 + 1) * 64], d2 = sp[(r * 4 + 2) * 64];
                wl[r] = __builtin_amdgcn_alignbyte(d1, d0, sh); wh[r] = __builtin_amdgcn_alignbyte(d2, d1, sh); w8[r] = (d2 >> (8 * sh)) & 255u;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 9; ++r) {                                         // 9 bytes per row: an unaligned 8-byte load and a byte
                const uint8_t *rp = cur + (size_t)(v_r + r - 4) * w + (u_r - 4);
                ygz_load8(rp, wl[r], wh[r]); w8[r] = (uint32_t)rp[8];
            }
        }
#define WIN(r, c) ((c) < 8 ? YGZ_BYTE(wl[r], wh[r], (c) & 7) : (int)w8[r])
#pragma unroll
        for (int y = 0; y < 8; ++y) {
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const float tl = (float)WIN(y, x), tr = (float)WIN(y, x + 1), bl = (float)WIN(y + 1, x), br = (float)WIN(y + 1, x + 1);
                const float sp = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(wTL, tl), __fmul_rn(wTR, tr)), __fmul_rn(wBL, bl)), __fmul_rn(wBR, br));
                const int k = (y + 1) * 10 + (x + 1);
                const float refv = (float)pwb[k * 64];
                const float res = __fadd_rn(__fsub_rn(sp, refv), mean_diff);
                const float jx = 0.5f * (float)((int)pwb[(k + 1) * 64] - (int)pwb[(k - 1) * 64]);
                const float jy = 0.5f * (float)((int)pwb[(k + 10) * 64] - (int)pwb[(k - 10) * 64]);
                J0 = __fsub_rn(J0, __fmul_rn(res, jx));
                J1 = __fsub_rn(J1, __fmul_rn(res, jy));
                J2 = __fsub_rn(J2, res);
                chi2 = __fadd_rn(chi2, __fmul_rn(res, res));
            }
        }
#undef WIN
        const float up0 = __fadd_rn(__fadd_rn(__fmul_rn(Hinv[0], J0), __fmul_rn(Hinv[1], J1)), __fmul_rn(Hinv[2], J2));
        const float up1 = __fadd_rn(__fadd_rn(__fmul_rn(Hinv[3], J0), __fmul_rn(Hinv[4], J1)), __fmul_rn(Hinv[5], J2));
        const float up2 = __fadd_rn(__fadd_rn(__fmul_rn(Hinv[6], J0), __fmul_rn(Hinv[7], J1)), __fmul_rn(Hinv[8], J2));
        u = __fadd_rn(u, up0); v = __fadd_rn(v, up1); mean_diff = __fadd_rn(mean_diff, up2);
        if (__fadd_rn(__fmul_rn(up0, up0), __fmul_rn(up1, up1)) < min_update_squared) { converged = true; break; }
    }
    *pu = (double)u; *pv = (double)v;
    if (chi2_out) *chi2_out = chi2;
    return converged && chi2 < 20000.f;
}

// Matcher::WarpAffine (Matcher.cpp:438-466; the legacy utils::WarpAffine, src/utils.cpp:66-98, is the same code) with half_patch_size 5:
// the 10 x 10 affine-warped reference patch into the lane's LDS column, bilinear samples truncated to uchar (GetBilateralInterpUchar)
static __device__ __forceinline__ void warp_affine_lds(const double Am[4], const uint8_t *img, int rw, int rh, const double px_ref[2], int Lr, int sl,
                                                       uint8_t *pwb)
{
    const double det = Am[0] * Am[3] - Am[2] * Am[1];
    const double invdet = 1.0 / det;
    const double R0 = Am[3] * invdet, R1 = -Am[1] * invdet, R2 = -Am[2] * invdet, R3 = Am[0] * invdet;
    const double rx = px_ref[0] / (double)(1 << Lr), ry = px_ref[1] / (double)(1 << Lr);
    const double sc = (double)(1 << sl);
    // A row of ten samples at a time: the ten sample positions first, then the twenty 2-byte gathers as ONE batch (out-of-image
    // samples read the image origin and are discarded), then the interpolation.  Sample by sample the loop was a chain of 100
    // dependent memory latencies.  (Copying the bounding box of the warped patch into the lane's LDS column first -- 64 dword loads
    // instead of 200 gathers, then 400 LDS reads -- was measured slower: 0.47 against 0.43 ms per 512 pairs.)
    for (int y = 0; y < 10; ++y) {
        double xx[10], yy[10];
        uint32_t top[10], bot[10];
        bool inb[10];
#pragma unroll
        for (int x = 0; x < 10; ++x) {
            double ppx = (double)(x - 5), ppy = (double)(y - 5);
            ppx *= sc; ppy *= sc;
            const double qx = (R0 * ppx + R1 * ppy) + rx, qy = (R2 * ppx + R3 * ppy) + ry;
            inb[x] = !(qx < 0 || qy < 0 || qx >= rw - 1 || qy >= rh - 1);
            // cvutils::GetBilateralInterpUchar (CVUtils.h:59-71)
            xx[x] = qx - floor(qx); yy[x] = qy - floor(qy);
            const uint8_t *d = inb[x] ? img + (size_t)((int)qy) * rw + (int)qx : img;
            top[x] = *reinterpret_cast<const ygz_u16u *>(d); bot[x] = *reinterpret_cast<const ygz_u16u *>(d + rw);      // 2 gathers, not 4
        }
#pragma unroll
        for (int x = 0; x < 10; ++x) {
            const int d00 = (int)(top[x] & 255u), d01 = (int)(top[x] >> 8), d10 = (int)(bot[x] & 255u), d11 = (int)(bot[x] >> 8);
            const uint8_t val = (uint8_t)((1 - xx[x]) * (1 - yy[x]) * d00 + xx[x] * (1 - yy[x]) * d01 + (1 - xx[x]) * yy[x] * d10 + xx[x] * yy[x] * d11);
            pwb[(y * 10 + x) * 64] = inb[x] ? val : (uint8_t)0;
        }
    }
}

struct FdpArgs {
    const uint8_t *lvl[YGZ_MAX_LEVELS];
    size_t sstride[YGZ_MAX_LEVELS];               // bytes from one slot's level image to the next slot's (w * h in a context's frame store; the row size in a keyframe store)
    int w[YGZ_MAX_LEVELS], h[YGZ_MAX_LEVELS];
    int n_levels, cells, n_pairs;
    Cam cam;
    const int32_t *pair_q, *pair_t, *trk_n;       // cur slot, ref slot, candidates per pair
    const double *pair_T;                         // [pairs][2][7] (T_ref, T_cur)
    const double *trk_px, *trk_depth; const int32_t *trk_level;
    double *px_cur; int32_t *search_level; uint8_t *ok;      // [pairs][cells]
    const uint8_t *cand;                                     // [pairs][cells] or nullptr: 0 = not a candidate (FindCandidates dropped it)
    int prio;                                                // != 0: raise the wavefronts' issue priority (gather-latency bound)
};

// the body shared by both Matcher::FindDirectProjection overloads (Matcher.cpp:356-417) from Pixel2Camera(px_ref, depth) on:
// GetWarpAffineMatrix, GetBestSearchLevel, WarpAffine into the lane's LDS column, Align2D, rescale.  px_cur in (prediction) /
// out (refined, level-0 pixels); returns success && InFrame(px_cur, 10).
static __device__ __forceinline__ bool fdp_core(const FdpArgs &A, int ref_slot, int cur_slot, const double *Tr7, const double *Tc7,
                                                const double px_ref[2], double depth, int Lr, uint8_t *pwb, uint32_t *stg, double px_cur[2],
                                                int *sl_out)
{
    double pt_ref[3];
    pixel2camera_d(A.cam, px_ref, depth, pt_ref);
    Se3 T_ref, T_cur, Tri, TCR;
    for (int k = 0; k < 4; ++k) { T_ref.q[k] = Tr7[k]; T_cur.q[k] = Tc7[k]; }
    for (int k = 0; k < 3; ++k) { T_ref.t[k] = Tr7[4 + k]; T_cur.t[k] = Tc7[4 + k]; }
    se3_inv_d(&T_ref, &Tri);
    se3_mul_d(&T_cur, &Tri, &TCR);
    // GetWarpAffineMatrix (Matcher.cpp:420-436)
    double Am[4];
    {
        double pw[3], pdu[3], pdv[3], q[3], pc[2], pu[2], pv[2];
        se3_act_d(&Tri, pt_ref, pw);
        const double s = (double)(1 << Lr);
        const double pxu[2] = { px_ref[0] + 4.0 * s, px_ref[1] + 0.0 * s };
        const double pxv[2] = { px_ref[0] + 0.0 * s, px_ref[1] + 4.0 * s };
        pixel2camera_d(A.cam, pxu, pt_ref[2], pdu);
        pixel2camera_d(A.cam, pxv, pt_ref[2], pdv);
        se3_act_d(&TCR, pw, q);  camera2pixel_d(A.cam, q, pc);
        se3_act_d(&TCR, pdu, q); camera2pixel_d(A.cam, q, pu);
        se3_act_d(&TCR, pdv, q); camera2pixel_d(A.cam, q, pv);
        Am[0] = (pu[0] - pc[0]) / 4; Am[2] = (pu[1] - pc[1]) / 4;
        Am[1] = (pv[0] - pc[0]) / 4; Am[3] = (pv[1] - pc[1]) / 4;
    }
    // GetBestSearchLevel (Matcher.h:123-134)
    int sl = 0;
    {
        double D = Am[0] * Am[3] - Am[2] * Am[1];
        while (D > 3.0 && sl < A.n_levels - 1) { sl += 1; D *= 0.25; }
    }
    // WarpAffine (Matcher.cpp:438-466), half_patch_size 5
    warp_affine_lds(Am, A.lvl[Lr] + (size_t)ref_slot * A.sstride[Lr], A.w[Lr], A.h[Lr], px_ref, Lr, sl, pwb);
    const int cw = A.w[sl], ch = A.h[sl];
    const uint8_t *cur = A.lvl[sl] + (size_t)cur_slot * A.sstride[sl];
    double u = px_cur[0] / (double)(1 << sl), v = px_cur[1] / (double)(1 << sl);
    const bool good = align2d_core(cur, cw, ch, pwb, stg, 10, &u, &v, nullptr);
    const double ox = u * (double)(1 << sl), oy = v * (double)(1 << sl);
    px_cur[0] = ox; px_cur[1] = oy;
    *sl_out = sl;
    const bool inframe = ox >= 10 && ox < A.w[0] - 10 && oy >= 10 && oy < A.h[0] - 10;     // Frame::InFrame(px,10)
    return inframe && good;
}

// Matcher::FindDirectProjection (Feature* overload, Matcher.cpp:385-417)
__global__ __launch_bounds__(64) void k_find_direct_projection(FdpArgs A)
{
    __shared__ uint8_t pwb_all[100 * 64];
    __shared__ uint32_t stg_all[STG_DWORDS * 64];
    int bx, pair;
    ygz_raise_prio(A.prio);
    if (!ygz_xcd_remap(A.n_pairs, bx, pair)) return;
    const int ii = bx * 64 + threadIdx.x;
    if (ii >= A.trk_n[pair]) return;
    const size_t i = (size_t)pair * A.cells + ii;
    const double depth = A.trk_depth[i];
    if (depth < 0 || (A.cand && !A.cand[i])) { A.ok[i] = 0; A.search_level[i] = 0; return; }
    const double px_ref[2] = { A.trk_px[2 * i], A.trk_px[2 * i + 1] };
    double px_cur[2] = { A.px_cur[2 * i], A.px_cur[2 * i + 1] };
    int sl;
    const bool ok = fdp_core(A, A.pair_t[pair], A.pair_q[pair], A.pair_T + 14 * (size_t)pair, A.pair_T + 14 * (size_t)pair + 7,
                             px_ref, depth, A.trk_level[i], pwb_all + threadIdx.x, stg_all + threadIdx.x, px_cur, &sl);
    A.px_cur[2 * i] = px_cur[0]; A.px_cur[2 * i + 1] = px_cur[1];
    A.search_level[i] = sl;
    A.ok[i] = (uint8_t)ok;
}

// ---- LocalMapping::FindCandidates + ProjectMapPoints (src/Module/LocalMapping.cpp:47-120) ----
struct LmapArgs {
    FdpArgs F;                                     // levels, sizes, camera (the pair / track members are unused)
    int cur_slot, P, C, K;
    const double *T_cur;                           // [7]
    const double *pos_world; const uint8_t *point_bad;              // [P][3], [P]
    const int32_t *kf_slot; const double *kf_T;                      // [K], [K][7]
    const int32_t *cand_point, *cand_kf, *cand_level; const double *cand_px_ref;      // [C]
    uint8_t *in_view; double *px_proj; int32_t *match_cand; double *px_match; int32_t *match_level;   // [P]
    double *cand_px; int32_t *cand_sl; uint8_t *cand_ok;             // [C]
    int given_px;                                  // != 0: px_proj holds the caller's predictions and every point counts as in view (no FindCandidates step)
};

// FindCandidates (:47-79), lane = map point: World2Camera, Camera2Pixel, z < 0 or !InFrame(px, 20) -> not in view
__global__ __launch_bounds__(256) void k_lmap_project(LmapArgs A)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= A.P) return;
    A.match_cand[p] = 0x7fffffff; A.match_level[p] = 0;
    A.px_match[2 * p] = 0.0; A.px_match[2 * p + 1] = 0.0;
    if (A.given_px) { A.in_view[p] = 1; return; }
    double px[2] = { 0.0, 0.0 };
    bool vis = false;
    if (!(A.point_bad && A.point_bad[p])) {
        Se3 T;
        for (int k = 0; k < 4; ++k) T.q[k] = A.T_cur[k];
        for (int k = 0; k < 3; ++k) T.t[k] = A.T_cur[4 + k];
        const double pw[3] = { A.pos_world[3 * (size_t)p], A.pos_world[3 * (size_t)p + 1], A.pos_world[3 * (size_t)p + 2] };
        double pc[3];
        se3_act_d(&T, pw, pc);
        camera2pixel_d(A.F.cam, pc, px);
        vis = !(pc[2] < 0) && px[0] >= 20 && px[0] < A.F.w[0] - 20 && px[1] >= 20 && px[1] < A.F.h[0] - 20;
    }
    A.px_proj[2 * p] = px[0]; A.px_proj[2 * p + 1] = px[1];
    A.in_view[p] = (uint8_t)vis;
}

// ProjectMapPoints (:81-120), lane = candidate: every candidate of an in-view point is refined (the reference skips the
// candidates that follow a point's first success -- their result is never used, so evaluating them changes nothing);
// the first success in candidate order is kept with an integer atomicMin on the candidate index.
__global__ __launch_bounds__(64) void k_lmap_match(LmapArgs A)
{
    __shared__ uint8_t pwb_all[100 * 64];
    __shared__ uint32_t stg_all[STG_DWORDS * 64];
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= A.C) return;
    const int p = A.cand_point[c], kf = A.cand_kf[c];
    A.cand_ok[c] = 0; A.cand_sl[c] = 0;
    if (p < 0 || p >= A.P || kf < 0 || kf >= A.K || !A.in_view[p]) return;
    const double *Tr7 = A.kf_T + 7 * (size_t)kf;
    Se3 T;
    for (int k = 0; k < 4; ++k) T.q[k] = Tr7[k];
    for (int k = 0; k < 3; ++k) T.t[k] = Tr7[4 + k];
    const double pw[3] = { A.pos_world[3 * (size_t)p], A.pos_world[3 * (size_t)p + 1], A.pos_world[3 * (size_t)p + 2] };
    double pr[3];
    se3_act_d(&T, pw, pr);                                   // depth = World2Camera(mp->_pos_world, ref->_TCW)[2] (Matcher.cpp:362), sign not tested
    const double px_ref[2] = { A.cand_px_ref[2 * (size_t)c], A.cand_px_ref[2 * (size_t)c + 1] };
    double px_cur[2] = { A.px_proj[2 * p], A.px_proj[2 * p + 1] };
    int sl;
    const bool ok = fdp_core(A.F, A.kf_slot[kf], A.cur_slot, Tr7, A.T_cur, px_ref, pr[2], A.cand_level[c], pwb_all + threadIdx.x, stg_all + threadIdx.x, px_cur, &sl);
    A.cand_px[2 * (size_t)c] = px_cur[0]; A.cand_px[2 * (size_t)c + 1] = px_cur[1];
    A.cand_sl[c] = sl; A.cand_ok[c] = (uint8_t)ok;
    if (ok) atomicMin(&A.match_cand[p], c);
}

__global__ __launch_bounds__(256) void k_lmap_gather(LmapArgs A)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= A.P) return;
    const int c = A.match_cand[p];
    if (c == 0x7fffffff) { A.match_cand[p] = -1; return; }
    A.px_match[2 * p] = A.cand_px[2 * (size_t)c]; A.px_match[2 * p + 1] = A.cand_px[2 * (size_t)c + 1];
    A.match_level[p] = A.cand_sl[c];
}

__global__ __launch_bounds__(64) void k_align2d(const uint8_t *__restrict__ cur, int w, int h,
                                                const uint8_t *__restrict__ pwb_in, double *__restrict__ uv,
                                                uint8_t *__restrict__ ok, float *__restrict__ chi2, int n, int n_iter)
{
    __shared__ uint8_t pwb_all[100 * 64];
    __shared__ uint32_t stg_all[STG_DWORDS * 64];
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    uint8_t *pwb = pwb_all + threadIdx.x;
    for (int k = 0; k < 100; ++k) pwb[k * 64] = pwb_in[(size_t)i * 100 + k];
    double u = uv[2 * i], v = uv[2 * i + 1];
    float c2 = 0.f;
    const bool good = align2d_core(cur, w, h, pwb, stg_all + threadIdx.x, n_iter, &u, &v, &c2);
    uv[2 * i] = u; uv[2 * i + 1] = v; ok[i] = (uint8_t)good; chi2[i] = c2;
}

int ygz_launch_fdp(ygz_hip_ctx *ctx, int n_pairs)
{
    FdpArgs A;
    for (int L = 0; L < YGZ_MAX_LEVELS; ++L) { A.lvl[L] = ctx->lvl[L]; A.w[L] = ctx->lw[L]; A.h[L] = ctx->lh[L]; A.sstride[L] = (size_t)ctx->lw[L] * ctx->lh[L]; }
    A.n_levels = ctx->prm.pyramid_levels; A.cells = ctx->cells; A.n_pairs = n_pairs;
    A.cam = Cam{ ctx->prm.fx, ctx->prm.fy, ctx->prm.cx, ctx->prm.cy };
    A.pair_q = ctx->pair_q; A.pair_t = ctx->pair_t; A.trk_n = ctx->trk_n; A.pair_T = ctx->pair_T;
    A.trk_px = ctx->trk_px; A.trk_depth = ctx->trk_depth; A.trk_level = ctx->trk_level;
    A.px_cur = ctx->fdp_px; A.search_level = ctx->fdp_level; A.ok = ctx->fdp_ok; A.cand = ctx->fdp_cand;
    A.prio = (ctx->wave_prio_mask >> 1) & 1;
    YGZ_LAUNCH(ctx, KID_FDP, k_find_direct_projection, dim3(ygz_div_up(ctx->cells, 64), ygz_round_up8(n_pairs)), dim3(64), A);
    YGZ_HIPCHK(ctx, hipGetLastError());
    return YGZ_OK;
}

// ---- the observations of a BA window by direct projection (the offline run's stand-in for what LocalMapping::ProjectMapPoints leaves in
// every keyframe: a Feature at the patch-aligned pixel of each local map point, src/Module/LocalMapping.cpp:82-120) ----
struct WinProjArgs { FdpArgs F; YgzWinProject P; };
__global__ __launch_bounds__(64) void k_win_project(WinProjArgs A)
{
    __shared__ uint8_t pwb_all[100 * 64];
    __shared__ uint32_t stg_all[STG_DWORDS * 64];
    int bx, p;
    if (!ygz_xcd_remap(A.P.n_pairs, bx, p)) return;
    const int w = A.P.pair_w[p], j = A.P.pair_j[p], s = bx * 64 + (int)threadIdx.x;
    const int S = A.P.set_base + w;
    if (s >= A.P.counts[S] || s >= A.P.Pcap) return;
    const uint8_t *srow = A.P.rows + (size_t)S * A.P.row_bytes;
    const double *spx = reinterpret_cast<const double *>(srow + A.P.off_px), *sdp = reinterpret_cast<const double *>(srow + A.P.off_depth);
    const int32_t *slv = reinterpret_cast<const int32_t *>(srow + A.P.off_level);
    const size_t o = (size_t)p * A.P.stride + s;
    const double px_ref[2] = { spx[2 * s], spx[2 * s + 1] };
    double pw[3];
    pixel2camera_d(A.F.cam, px_ref, sdp[s], pw);                // the map point in the anchor's camera = the window's gauge (Camera.h:53-59)
    const double *Tc7 = A.P.Tj + 7 * ((size_t)w * A.P.Kcap + j);
    Se3 T;
    for (int k = 0; k < 4; ++k) T.q[k] = Tc7[k];
    for (int k = 0; k < 3; ++k) T.t[k] = Tc7[4 + k];
    double pc[3], px_cur[2];
    se3_act_d(&T, pw, pc);
    camera2pixel_d(A.F.cam, pc, px_cur);
    // FindCandidates (LocalMapping.cpp:60-64): behind the camera or outside InFrame(px, 20) -> not a candidate
    const bool vis = !(pc[2] < 0) && px_cur[0] >= 20 && px_cur[0] < A.F.w[0] - 20 && px_cur[1] >= 20 && px_cur[1] < A.F.h[0] - 20;
    bool ok = false;
    if (vis) {
        const double I7[7] = { 0, 0, 0, 1, 0, 0, 0 };           // the anchor's pose in its own gauge
        Se3 Ti; Ti.q[0] = Ti.q[1] = Ti.q[2] = 0; Ti.q[3] = 1; Ti.t[0] = Ti.t[1] = Ti.t[2] = 0;
        double pr[3];
        se3_act_d(&Ti, pw, pr);                                  // depth = World2Camera(mp->_pos_world, ref->_TCW)[2] (Matcher.cpp:362)
        int sl;
        ok = fdp_core(A.F, A.P.kf_index[(size_t)w * A.P.Kcap], A.P.kf_index[(size_t)w * A.P.Kcap + j], I7, Tc7, px_ref, pr[2], slv[s],
                      pwb_all + threadIdx.x, stg_all + threadIdx.x, px_cur, &sl);
    }
    A.P.obs_px[2 * o] = px_cur[0]; A.P.obs_px[2 * o + 1] = px_cur[1];
    A.P.obs_ok[o] = (uint8_t)ok;
}

int ygz_launch_win_project(ygz_hip_ctx *ctx, const YgzWinProject &P)
{
    WinProjArgs A;
    A.P = P;
    for (int L = 0; L < YGZ_MAX_LEVELS; ++L) {
        A.F.lvl[L] = L < P.n_levels ? P.rows + P.off_img[L] : nullptr; A.F.sstride[L] = P.row_bytes; A.F.w[L] = P.w[L]; A.F.h[L] = P.h[L];
    }
    A.F.n_levels = P.n_levels; A.F.cells = ctx->cells; A.F.n_pairs = P.n_pairs;
    A.F.cam = Cam{ ctx->prm.fx, ctx->prm.fy, ctx->prm.cx, ctx->prm.cy }; A.F.prio = 0;
    A.F.pair_q = A.F.pair_t = A.F.trk_n = nullptr; A.F.pair_T = A.F.trk_px = A.F.trk_depth = nullptr; A.F.trk_level = nullptr; A.F.cand = nullptr;
    A.F.px_cur = nullptr; A.F.search_level = nullptr; A.F.ok = nullptr;
    YGZ_LAUNCH(ctx, KID_FDP, k_win_project, dim3(ygz_div_up(P.Pcap, 64), ygz_round_up8(P.n_pairs)), dim3(64), A);
    YGZ_HIPCHK(ctx, hipGetLastError());
    return YGZ_OK;
}

extern "C" {

// single pair, host arrays: fills track set 0 and runs the batched kernel with one pair -- in pieces of `cells` candidates (the track set's size;
// candidates are independent, so any n is served: Matcher::FindDirectProjection has no limit either)
int ygz_hip_find_direct_projection(ygz_hip_ctx *ctx, const ygz_align_pair *pair, const double *px_ref, const double *depth_ref,
                                   const int32_t *level_ref, double *px_cur, int32_t *search_level, uint8_t *ok, int n)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || !pair || n < 0) return YGZ_E_INVALID;
    if (n == 0) return YGZ_OK;
    if (!px_ref || !depth_ref || !level_ref || !px_cur || !search_level || !ok) return YGZ_E_INVALID;
    if (pair->ref_slot < 0 || pair->ref_slot >= ctx->prm.max_frames || pair->cur_slot < 0 || pair->cur_slot >= ctx->prm.max_frames) return YGZ_E_INVALID;
    if (!ctx->pyr_valid[pair->ref_slot] || !ctx->pyr_valid[pair->cur_slot]) return YGZ_E_STATE;
    for (int i = 0; i < n; ++i) if (level_ref[i] < 0 || level_ref[i] >= ctx->prm.pyramid_levels) return YGZ_E_INVALID;
    int rc = ygz_track_set_pairs(ctx, &pair->cur_slot, &pair->ref_slot, pair->T_cur, pair->T_ref, 1);
    if (rc != YGZ_OK) return rc;
    for (int base = 0; base < n; base += ctx->cells) {
        const int m = n - base < ctx->cells ? n - base : ctx->cells;
        const size_t N = (size_t)m;
        YGZ_HIPCHK(ctx, hipMemcpyAsync(ctx->trk_px, px_ref + 2 * (size_t)base, N * 16, hipMemcpyHostToDevice, ctx->stream));
        YGZ_HIPCHK(ctx, hipMemcpyAsync(ctx->trk_depth, depth_ref + base, N * 8, hipMemcpyHostToDevice, ctx->stream));
        YGZ_HIPCHK(ctx, hipMemcpyAsync(ctx->trk_level, level_ref + base, N * 4, hipMemcpyHostToDevice, ctx->stream));
        YGZ_HIPCHK(ctx, hipMemcpyAsync(ctx->fdp_px, px_cur + 2 * (size_t)base, N * 16, hipMemcpyHostToDevice, ctx->stream));
        YGZ_HIPCHK(ctx, hipMemsetAsync(ctx->fdp_cand, 1, N, ctx->stream));
        YGZ_HIPCHK(ctx, hipMemcpyAsync(ctx->trk_n, &m, 4, hipMemcpyHostToDevice, ctx->stream));
        YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if ((rc = ygz_launch_fdp(ctx, 1)) != YGZ_OK) return rc;
        YGZ_HIPCHK(ctx, hipMemcpyAsync(px_cur + 2 * (size_t)base, ctx->fdp_px, N * 16, hipMemcpyDeviceToHost, ctx->stream));
        YGZ_HIPCHK(ctx, hipMemcpyAsync(search_level + base, ctx->fdp_level, N * 4, hipMemcpyDeviceToHost, ctx->stream));
        YGZ_HIPCHK(ctx, hipMemcpyAsync(ok + base, ctx->fdp_ok, N, hipMemcpyDeviceToHost, ctx->stream));
        YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return YGZ_OK;
}

int ygz_hip_align2d(ygz_hip_ctx *ctx, int cur_slot, int level, const uint8_t *pwb, const uint8_t *patch, double *uv,
                    uint8_t *ok, float *chi2, int n, int n_iter)
{
    YgzDeviceGuard dg_(ctx);
    (void)patch;     // the 8x8 patch is the interior of the 10x10 one (Matcher.cpp:369-375); kept for signature parity
    if (!ctx || n < 0 || cur_slot < 0 || cur_slot >= ctx->prm.max_frames || level < 0 || level >= ctx->n_levels_alloc) return YGZ_E_INVALID;
    if (n == 0) return YGZ_OK;
    if (!pwb || !uv || !ok) return YGZ_E_INVALID;
    if (!ctx->pyr_valid[cur_slot]) return YGZ_E_STATE;
    uint8_t *buf = nullptr;
    const size_t N = (size_t)n, bytes = N * (16 + 4 + 100 + 1) + 64;
    int rc = ygz_scratch(ctx, SCR_ALIGN_IN, bytes, (void **)&buf);
    if (rc != YGZ_OK) return rc;
    double *d_uv = (double *)buf; float *d_chi = (float *)(d_uv + 2 * N);
    uint8_t *d_pwb = (uint8_t *)(d_chi + N), *d_ok = d_pwb + 100 * N;
    YGZ_HIPCHK(ctx, hipMemcpyAsync(d_uv, uv, N * 16, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(d_pwb, pwb, N * 100, hipMemcpyHostToDevice, ctx->stream));
    const int w = ctx->lw[level], h = ctx->lh[level];
    YGZ_LAUNCH(ctx, KID_ALIGN2D, k_align2d, dim3(ygz_div_up(n, 64)), dim3(64),
                       ctx->lvl[level] + (size_t)cur_slot * w * h, w, h, d_pwb, d_uv, d_ok, d_chi, n, n_iter);
    YGZ_HIPCHK(ctx, hipGetLastError());
    YGZ_HIPCHK(ctx, hipMemcpyAsync(uv, d_uv, N * 16, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(ok, d_ok, N, hipMemcpyDeviceToHost, ctx->stream));
    if (chi2) YGZ_HIPCHK(ctx, hipMemcpyAsync(chi2, d_chi, N * 4, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return YGZ_OK;
}

}  // extern "C"

// per-candidate outputs of the local-map run (what k_lmap_match leaves for EVERY candidate, not only a point's first success)
struct LmapCandOut { uint8_t *ok; double *px; int32_t *sl; const double *px_given; };

// one run = LocalMapping::FindCandidates + ProjectMapPoints for the current frame against K resident keyframes
// the device / page-locked layout of one run (P points, K keyframes, Cn candidates) in a scratch block and its mirror
struct LmapLayout {
    size_t total;
    double *d_T, *d_pw, *d_kfT, *d_cpx, *d_proj, *d_pm, *d_candpx;
    int32_t *d_kfs, *d_cp, *d_ck, *d_cl, *d_mc, *d_ml, *d_csl;
    uint8_t *d_bad, *d_vis, *d_cok;
};
static size_t lmap_bytes(size_t Ps, size_t Ks, size_t Cs)
{ return (8 + 3 * Ps + 7 * Ks + 2 * Cs + 2 * Ps + 2 * Ps + 2 * Cs) * 8 + (Ks + 3 * Cs + 2 * Ps + Cs) * 4 + 2 * Ps + Cs; }
static LmapLayout lmap_layout(uint8_t *buf, size_t Ps, size_t Ks, size_t Cs)
{
    LmapLayout L;
    L.total = lmap_bytes(Ps, Ks, Cs);
    L.d_T = (double *)buf; L.d_pw = L.d_T + 8; L.d_kfT = L.d_pw + 3 * Ps; L.d_cpx = L.d_kfT + 7 * Ks; L.d_proj = L.d_cpx + 2 * Cs;
    L.d_pm = L.d_proj + 2 * Ps; L.d_candpx = L.d_pm + 2 * Ps;
    L.d_kfs = (int32_t *)(L.d_candpx + 2 * Cs); L.d_cp = L.d_kfs + Ks; L.d_ck = L.d_cp + Cs; L.d_cl = L.d_ck + Cs; L.d_mc = L.d_cl + Cs;
    L.d_ml = L.d_mc + Ps; L.d_csl = L.d_ml + Ps;
    L.d_bad = (uint8_t *)(L.d_csl + Cs); L.d_vis = L.d_bad + Ps; L.d_cok = L.d_vis + Ps;
    return L;
}
// defer: the run is queued (inputs up, kernels, results into the page-locked mirror of scratch block `scr`) and NOT waited for: lmap_collect
// hands the per-candidate results out later (ygz_hip_find_direct_projection_mp_begin / _end)
static int lmap_run(ygz_hip_ctx *ctx, int cur_slot, const double T_cur[7], const ygz_local_map *m,
                    uint8_t *in_view, double *px_proj, int32_t *match_cand, double *px_match, int32_t *match_level, int32_t *n_matched,
                    const LmapCandOut *co, int scr = SCR_LMAP, bool defer = false)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || !T_cur || !m || m->n_points < 0 || m->n_keyframes < 0 || m->n_candidates < 0) return YGZ_E_INVALID;
    if (cur_slot < 0 || cur_slot >= ctx->prm.max_frames) return YGZ_E_INVALID;
    if (!ctx->pyr_valid[cur_slot]) return YGZ_E_STATE;
    const int P = m->n_points, K = m->n_keyframes, Cn = m->n_candidates;
    if (n_matched) *n_matched = 0;
    if (P == 0) return YGZ_OK;
    if (!m->pos_world || (!defer && (!in_view || !px_proj))) return YGZ_E_INVALID;
    if (!co && (!match_cand || !px_match || !match_level)) return YGZ_E_INVALID;
    if (K > 0 && (!m->kf_slot || !m->kf_T)) return YGZ_E_INVALID;
    if (Cn > 0 && (!m->cand_point || !m->cand_kf || !m->cand_level || !m->cand_px_ref || K == 0)) return YGZ_E_INVALID;
    for (int i = 0; i < K; ++i) {
        if (m->kf_slot[i] < 0 || m->kf_slot[i] >= ctx->prm.max_frames) return YGZ_E_INVALID;
        if (!ctx->pyr_valid[m->kf_slot[i]]) return YGZ_E_STATE;
    }
    for (int i = 0; i < Cn; ++i) if (m->cand_level[i] < 0 || m->cand_level[i] >= ctx->prm.pyramid_levels) return YGZ_E_INVALID;
    const size_t Ps = (size_t)P, Ks = (size_t)K, Cs = (size_t)Cn;
    uint8_t *buf = nullptr;
    int rc = ygz_scratch(ctx, scr, lmap_bytes(Ps, Ks, Cs) + 64, (void **)&buf);
    if (rc != YGZ_OK) return rc;
    const LmapLayout Y = lmap_layout(buf, Ps, Ks, Cs);
    double *const d_T = Y.d_T, *const d_pw = Y.d_pw, *const d_kfT = Y.d_kfT, *const d_cpx = Y.d_cpx, *const d_proj = Y.d_proj, *const d_pm = Y.d_pm, *const d_candpx = Y.d_candpx;
    int32_t *const d_kfs = Y.d_kfs, *const d_cp = Y.d_cp, *const d_ck = Y.d_ck, *const d_cl = Y.d_cl, *const d_mc = Y.d_mc, *const d_ml = Y.d_ml, *const d_csl = Y.d_csl;
    uint8_t *const d_bad = Y.d_bad, *const d_vis = Y.d_vis, *const d_cok = Y.d_cok;
    // inputs packed into the page-locked mirror of the scratch block at the device offsets: ONE copy up (and one down below) instead of nine + five
    uint8_t *hb = nullptr;
    if ((rc = ygz_scratch_mirror(ctx, scr, (void **)&hb)) != YGZ_OK) return rc;
    const size_t total = Y.total;
#define H_(dptr) (hb + ((const uint8_t *)(dptr) - buf))
    memcpy(H_(d_T), T_cur, 56);
    memcpy(H_(d_pw), m->pos_world, Ps * 24);
    if (m->point_bad) memcpy(H_(d_bad), m->point_bad, Ps);
    if (K) { memcpy(H_(d_kfT), m->kf_T, Ks * 56); memcpy(H_(d_kfs), m->kf_slot, Ks * 4); }
    if (Cn) { memcpy(H_(d_cpx), m->cand_px_ref, Cs * 16); memcpy(H_(d_cp), m->cand_point, Cs * 4); memcpy(H_(d_ck), m->cand_kf, Cs * 4); memcpy(H_(d_cl), m->cand_level, Cs * 4); }
    if (co && co->px_given) memcpy(H_(d_proj), co->px_given, Ps * 16);
    if ((rc = ygz_kcopy(ctx, buf, hb, total, hipMemcpyHostToDevice)) != YGZ_OK) return rc;
    LmapArgs A;
    for (int L = 0; L < YGZ_MAX_LEVELS; ++L) { A.F.lvl[L] = ctx->lvl[L]; A.F.w[L] = ctx->lw[L]; A.F.h[L] = ctx->lh[L]; A.F.sstride[L] = (size_t)ctx->lw[L] * ctx->lh[L]; }
    A.F.n_levels = ctx->prm.pyramid_levels; A.F.cells = ctx->cells; A.F.n_pairs = 0;
    A.F.cam = Cam{ ctx->prm.fx, ctx->prm.fy, ctx->prm.cx, ctx->prm.cy }; A.F.prio = 0;
    A.F.pair_q = A.F.pair_t = A.F.trk_n = nullptr; A.F.pair_T = A.F.trk_px = A.F.trk_depth = nullptr; A.F.trk_level = nullptr; A.F.cand = nullptr;
    A.F.px_cur = nullptr; A.F.search_level = nullptr; A.F.ok = nullptr;
    A.cur_slot = cur_slot; A.P = P; A.C = Cn; A.K = K; A.T_cur = d_T;
    A.pos_world = d_pw; A.point_bad = m->point_bad ? d_bad : nullptr; A.kf_slot = d_kfs; A.kf_T = d_kfT;
    A.cand_point = d_cp; A.cand_kf = d_ck; A.cand_level = d_cl; A.cand_px_ref = d_cpx;
    A.in_view = d_vis; A.px_proj = d_proj; A.match_cand = d_mc; A.px_match = d_pm; A.match_level = d_ml;
    A.cand_px = d_candpx; A.cand_sl = d_csl; A.cand_ok = d_cok;
    A.given_px = co && co->px_given ? 1 : 0;
    YGZ_LAUNCH(ctx, KID_LMAP_AUX, k_lmap_project, dim3(ygz_div_up(P, 256)), dim3(256), A);
    if (Cn) YGZ_LAUNCH(ctx, KID_LMAP_MATCH, k_lmap_match, dim3(ygz_div_up(Cn, 64)), dim3(64), A);
    if (!co) YGZ_LAUNCH(ctx, KID_LMAP_AUX, k_lmap_gather, dim3(ygz_div_up(P, 256)), dim3(256), A);
    YGZ_HIPCHK(ctx, hipGetLastError());
    if ((rc = ygz_kcopy(ctx, hb, buf, total, hipMemcpyDeviceToHost)) != YGZ_OK) return rc;
    if (defer) return YGZ_OK;
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(in_view, H_(d_vis), Ps); memcpy(px_proj, H_(d_proj), Ps * 16);
    if (co) {
        if (Cn) { memcpy(co->ok, H_(d_cok), Cs); memcpy(co->px, H_(d_candpx), Cs * 16); memcpy(co->sl, H_(d_csl), Cs * 4); }
        return YGZ_OK;
    }
    memcpy(match_cand, H_(d_mc), Ps * 4); memcpy(px_match, H_(d_pm), Ps * 16); memcpy(match_level, H_(d_ml), Ps * 4);
#undef H_
    if (n_matched) { int n = 0; for (int p = 0; p < P; ++p) n += match_cand[p] >= 0; *n_matched = n; }
    return YGZ_OK;
}

extern "C" {

// one call = LocalMapping::FindCandidates + ProjectMapPoints for the current frame against K resident keyframes
int ygz_hip_track_local_map(ygz_hip_ctx *ctx, int cur_slot, const double T_cur[7], const ygz_local_map *m,
                            uint8_t *in_view, double *px_proj, int32_t *match_cand, double *px_match, int32_t *match_level, int32_t *n_matched)
{
    return lmap_run(ctx, cur_slot, T_cur, m, in_view, px_proj, match_cand, px_match, match_level, n_matched, nullptr);
}

// Matcher::FindDirectProjection, MapPoint overload (Matcher.cpp:356-383), for C independent candidates over K resident reference keyframes in one
// launch -- the per-candidate call LocalMapping::ProjectMapPoints makes (LocalMapping.cpp:98), every candidate evaluated, none skipped.
int ygz_hip_find_direct_projection_mp(ygz_hip_ctx *ctx, int cur_slot, const double T_cur[7], int n_keyframes, const int32_t *kf_slot, const double *kf_T,
                                      int n, const int32_t *cand_kf, const double *pos_world, const double *px_ref, const int32_t *level_ref,
                                      const double *px_in, uint8_t *in_view, double *px_proj, uint8_t *ok, double *px_cur, int32_t *search_level)
{
    if (!ctx || n < 0) return YGZ_E_INVALID;
    if (n == 0) return YGZ_OK;
    if (!cand_kf || !pos_world || !px_ref || !level_ref || !ok || !px_cur || !search_level) return YGZ_E_INVALID;
    if (!px_in && (!in_view || !px_proj)) return YGZ_E_INVALID;
    std::vector<int32_t> iota((size_t)n);
    for (int i = 0; i < n; ++i) iota[i] = i;
    std::vector<uint8_t> vis; std::vector<double> proj;
    if (!in_view) { vis.resize((size_t)n); in_view = vis.data(); }
    if (!px_proj) { proj.resize(2 * (size_t)n); px_proj = proj.data(); }
    ygz_local_map m;
    m.n_points = n; m.pos_world = pos_world; m.point_bad = nullptr;
    m.n_keyframes = n_keyframes; m.kf_slot = kf_slot; m.kf_T = kf_T;
    m.n_candidates = n; m.cand_point = iota.data(); m.cand_kf = cand_kf; m.cand_level = level_ref; m.cand_px_ref = px_ref;
    LmapCandOut co = { ok, px_cur, search_level, px_in };
    return lmap_run(ctx, cur_slot, T_cur, &m, in_view, px_proj, nullptr, nullptr, nullptr, nullptr, &co);
}


// The same launch in two halves, for a caller that has host work of its own between asking and needing the answers (the class surface queues the
// speculative launch of FdpMemo at the end of Matcher::SparseImageAlignment and collects it at the first Matcher::FindDirectProjection call: the
// unchanged caller spends 0.2 ms in LocalMapping::FindCandidates in between).  _begin: px_in = NULL form only (the launch makes FindCandidates'
// prediction); queues uploads, kernels and the copy back into a page-locked block of its own, does not wait.  _end (same n): waits and hands out what
// ygz_hip_find_direct_projection_mp would have returned.  One run can be pending per context; a second _begin replaces it (after waiting for it),
// YGZ_E_STATE from _end when nothing is pending (or n differs).  Other calls on the context in between are fine (they queue behind it).
int ygz_hip_find_direct_projection_mp_begin(ygz_hip_ctx *ctx, int cur_slot, const double T_cur[7], int n_keyframes, const int32_t *kf_slot, const double *kf_T,
                                            int n, const int32_t *cand_kf, const double *pos_world, const double *px_ref, const int32_t *level_ref)
{
    if (!ctx || n < 1 || !cand_kf || !pos_world || !px_ref || !level_ref) return YGZ_E_INVALID;
    YgzDeviceGuard dg_(ctx);
    if (ctx->lmap_async_n) { YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); ctx->lmap_async_n = 0; }   // the page-locked block is about to be rewritten
    std::vector<int32_t> iota((size_t)n);
    for (int i = 0; i < n; ++i) iota[i] = i;
    ygz_local_map m;
    m.n_points = n; m.pos_world = pos_world; m.point_bad = nullptr;
    m.n_keyframes = n_keyframes; m.kf_slot = kf_slot; m.kf_T = kf_T;
    m.n_candidates = n; m.cand_point = iota.data(); m.cand_kf = cand_kf; m.cand_level = level_ref; m.cand_px_ref = px_ref;
    LmapCandOut co = { nullptr, nullptr, nullptr, nullptr };
    const int rc = lmap_run(ctx, cur_slot, T_cur, &m, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &co, SCR_GEN_0 + 10, true);
    if (rc == YGZ_OK) { ctx->lmap_async_n = n; ctx->lmap_async_k = n_keyframes; }
    return rc;
}

int ygz_hip_find_direct_projection_mp_end(ygz_hip_ctx *ctx, int n, uint8_t *in_view, double *px_proj, uint8_t *ok, double *px_cur, int32_t *search_level)
{
    if (!ctx || !in_view || !px_proj || !ok || !px_cur || !search_level) return YGZ_E_INVALID;
    if (ctx->lmap_async_n == 0 || ctx->lmap_async_n != n) return YGZ_E_STATE;
    YgzDeviceGuard dg_(ctx);
    const size_t Ns = (size_t)n;
    ctx->lmap_async_n = 0;
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    uint8_t *buf = (uint8_t *)ctx->scratch[SCR_GEN_0 + 10], *hb = (uint8_t *)ctx->scratch_host[SCR_GEN_0 + 10];
    if (!buf || !hb) return YGZ_E_STATE;
    const LmapLayout Y = lmap_layout(buf, Ns, (size_t)ctx->lmap_async_k, Ns);
#define H_(dptr) (hb + ((const uint8_t *)(dptr) - buf))
    memcpy(in_view, H_(Y.d_vis), Ns); memcpy(px_proj, H_(Y.d_proj), Ns * 16);
    memcpy(ok, H_(Y.d_cok), Ns); memcpy(px_cur, H_(Y.d_candpx), Ns * 16); memcpy(search_level, H_(Y.d_csl), Ns * 4);
#undef H_
    return YGZ_OK;
}

}  // extern "C"

// ---- cvutils::DepthFromTriangulation (include/ygz/Algorithm/CVUtils.h:18-38), Eigen's evaluation order (2x2 inverse =
// adjugate / det, (-inv A^T) formed before it multiplies t).  T7 = (qx,qy,qz,qw,tx,ty,tz) of T_search_ref.
__device__ __forceinline__ bool tri_depth_d(const double *T7, const double fr[3], const double fc[3], double det_th, double *depth1, double *depth2)
{
    double R[9];
    { const double q[4] = { T7[0], T7[1], T7[2], T7[3] }; quat_to_R_d(q, R); }
    double a0[3];
    for (int r = 0; r < 3; ++r) a0[r] = R[3 * r] * fr[0] + R[3 * r + 1] * fr[1] + R[3 * r + 2] * fr[2];
    const double a1[3] = { -fc[0], -fc[1], -fc[2] };
    const double m00 = a0[0] * a0[0] + a0[1] * a0[1] + a0[2] * a0[2], m01 = a0[0] * a1[0] + a0[1] * a1[1] + a0[2] * a1[2];
    const double m10 = a1[0] * a0[0] + a1[1] * a0[1] + a1[2] * a0[2], m11 = a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2];
    const double det = m00 * m11 - m10 * m01;
    if (det < det_th) return false;
    const double invdet = 1.0 / det;
    const double i00 = -(m11 * invdet), i01 = -(-m01 * invdet), i10 = -(-m10 * invdet), i11 = -(m00 * invdet);
    double M[6];
    for (int c = 0; c < 3; ++c) { M[c] = i00 * a0[c] + i01 * a1[c]; M[3 + c] = i10 * a0[c] + i11 * a1[c]; }
    *depth1 = fabs(M[0] * T7[4] + M[1] * T7[5] + M[2] * T7[6]);
    *depth2 = fabs(M[3] * T7[4] + M[4] * T7[5] + M[5] * T7[6]);
    return true;
}

// batch of ray pairs, lane = pair
__global__ __launch_bounds__(256) void k_depth_from_triangulation(const double *__restrict__ T /*q(4) t(3)*/, const double *__restrict__ f_ref,
                                                                  const double *__restrict__ f_cur, int n, double det_th,
                                                                  double *__restrict__ depth1, double *__restrict__ depth2, uint8_t *__restrict__ ok)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double fr[3] = { f_ref[3 * (size_t)i], f_ref[3 * (size_t)i + 1], f_ref[3 * (size_t)i + 2] };
    const double fc[3] = { f_cur[3 * (size_t)i], f_cur[3 * (size_t)i + 1], f_cur[3 * (size_t)i + 2] };
    double d1, d2;
    if (!tri_depth_d(T, fr, fc, det_th, &d1, &d2)) { ok[i] = 0; return; }
    depth1[i] = d1; depth2[i] = d2; ok[i] = 1;
}

// ---- the triangulation loop of LocalMapping::CreateNewMapPoints (src/Module/LocalMapping.cpp:416-495, first branch: neither feature has
// a map point), lane = matched feature pair of (frame 1 = current keyframe, frame 2 = neighbour): parallax test, DepthFromTriangulation,
// FindDirectProjection of feature 1 into frame 2 with that depth (the Feature overload: fdp_core), second triangulation with the
// refined pixel, reprojection test, map point.  The reference's loop is sequential only because it appends to lists; the pairs are
// independent.
struct CmpArgs {
    FdpArgs F;
    int slot1, slot2, n;
    const double *T;                 // T1 (7), T2 (7), T12.inverse() (7), T1.inverse() (7)
    const double *px1; const int32_t *level1; double *px2;
    int32_t *code; double *depth1, *depth2, *pos_world; int32_t *search_level;
};
__global__ __launch_bounds__(64) void k_create_map_points(CmpArgs A)
{
    __shared__ uint8_t pwb_all[100 * 64];
    __shared__ uint32_t stg_all[STG_DWORDS * 64];
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= A.n) return;
    const double *T1 = A.T, *T2 = A.T + 7, *T21 = A.T + 14, *T1i = A.T + 21;
    const double p1[2] = { A.px1[2 * (size_t)i], A.px1[2 * (size_t)i + 1] };
    double p2[2] = { A.px2[2 * (size_t)i], A.px2[2 * (size_t)i + 1] };
    double pt1[3], pt2[3], d1 = 0, d2 = 0;
    pixel2camera_d(A.F.cam, p1, 1.0, pt1); pixel2camera_d(A.F.cam, p2, 1.0, pt2);
    int code = 0, sl = 0;
    do {
        const double dot = pt1[0] * pt2[0] + pt1[1] * pt2[1] + pt1[2] * pt2[2];
        const double n1 = sqrt(pt1[0] * pt1[0] + pt1[1] * pt1[1] + pt1[2] * pt1[2]), n2 = sqrt(pt2[0] * pt2[0] + pt2[1] * pt2[1] + pt2[2] * pt2[2]);
        if (dot / (n1 * n2) >= 0.9998) { code = 1; break; }                       // :433-435
        if (!tri_depth_d(T21, pt1, pt2, 1e-5, &d1, &d2) || d1 < 0 || d2 < 0) { code = 2; break; }
        const bool ok = fdp_core(A.F, A.slot1, A.slot2, T1, T2, p1, d1, A.level1[i], pwb_all + threadIdx.x, stg_all + threadIdx.x, p2, &sl);
        if (!ok) { code = 3; break; }
        A.px2[2 * (size_t)i] = p2[0]; A.px2[2 * (size_t)i + 1] = p2[1];           // fea2->_pixel = px_curr (:453)
        pixel2camera_d(A.F.cam, p2, 1.0, pt2);
        if (!tri_depth_d(T21, pt1, pt2, 1e-5, &d1, &d2) || d1 < 0 || d2 < 0) { code = 4; break; }
        const double ptt[3] = { pt1[0] * d1, pt1[1] * d1, pt1[2] * d1 };
        Se3 S;
        for (int k = 0; k < 4; ++k) S.q[k] = T21[k];
        for (int k = 0; k < 3; ++k) S.t[k] = T21[4 + k];
        double pc[3], pr[2];
        se3_act_d(&S, ptt, pc);
        camera2pixel_d(A.F.cam, pc, pr);
        const double rx = pr[0] - p2[0], ry = pr[1] - p2[1];
        if (sqrt(rx * rx + ry * ry) > 5.991) { code = 5; break; }                  // :460-466
        for (int k = 0; k < 4; ++k) S.q[k] = T1i[k];
        for (int k = 0; k < 3; ++k) S.t[k] = T1i[4 + k];
        double pw[3];
        se3_act_d(&S, ptt, pw);                                                    // Camera2World(pt1 * depth1, _current_kf->_TCW) (:478)
        A.depth1[i] = d1; A.depth2[i] = d2;
        A.pos_world[3 * (size_t)i] = pw[0]; A.pos_world[3 * (size_t)i + 1] = pw[1]; A.pos_world[3 * (size_t)i + 2] = pw[2];
    } while (0);
    A.code[i] = code; A.search_level[i] = sl;
}

extern "C" int ygz_hip_create_map_points(ygz_hip_ctx *ctx, int slot1, const double T1[7], int slot2, const double T2[7], int n, const double *px1,
                                         const int32_t *level1, double *px2, int32_t *code, double *depth1, double *depth2, double *pos_world,
                                         int32_t *search_level, int *n_created)
{
    YgzDeviceGuard dg_(ctx);
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || !T1 || !T2 || n < 0 || slot1 < 0 || slot1 >= ctx->prm.max_frames || slot2 < 0 || slot2 >= ctx->prm.max_frames) return YGZ_E_INVALID;
    if (n_created) *n_created = 0;
    if (n == 0) return YGZ_OK;
    if (!px1 || !level1 || !px2 || !code || !depth1 || !depth2 || !pos_world || !search_level) return YGZ_E_INVALID;
    if (!ctx->pyr_valid[slot1] || !ctx->pyr_valid[slot2]) return YGZ_E_STATE;
    for (int i = 0; i < n; ++i) if (level1[i] < 0 || level1[i] >= ctx->prm.pyramid_levels) return YGZ_E_INVALID;
    const size_t N = (size_t)n;
    double hT[28];
    {   // T12 = T1 * T2^-1 (LocalMapping.cpp:402); DepthFromTriangulation and the reprojection take T12.inverse(); the map point T1^-1
        Se3 a, b, bi, t12, t21, ai;
        for (int k = 0; k < 4; ++k) { a.q[k] = T1[k]; b.q[k] = T2[k]; }
        for (int k = 0; k < 3; ++k) { a.t[k] = T1[4 + k]; b.t[k] = T2[4 + k]; }
        se3_inv_d(&b, &bi); se3_mul_d(&a, &bi, &t12); se3_inv_d(&t12, &t21); se3_inv_d(&a, &ai);
        for (int k = 0; k < 7; ++k) { hT[k] = T1[k]; hT[7 + k] = T2[k]; }
        for (int k = 0; k < 4; ++k) { hT[14 + k] = t21.q[k]; hT[21 + k] = ai.q[k]; }
        for (int k = 0; k < 3; ++k) { hT[18 + k] = t21.t[k]; hT[25 + k] = ai.t[k]; }
    }
    uint8_t *buf = nullptr;
    const size_t o_px1 = 256, o_px2 = o_px1 + N * 16, o_d1 = o_px2 + N * 16, o_d2 = o_d1 + N * 8, o_pw = o_d2 + N * 8, o_lv = o_pw + N * 24,
                 o_code = o_lv + N * 4, o_sl = o_code + N * 4, total = o_sl + N * 4 + 64;
    int rc = ygz_scratch(ctx, SCR_GEN_0, total, (void **)&buf);
    if (rc != YGZ_OK) return rc;
    YGZ_HIPCHK(ctx, hipMemcpyAsync(buf, hT, sizeof(hT), hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(buf + o_px1, px1, N * 16, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(buf + o_px2, px2, N * 16, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(buf + o_lv, level1, N * 4, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemsetAsync(buf + o_d1, 0, N * 40, ctx->stream));
    CmpArgs A;
    for (int L = 0; L < YGZ_MAX_LEVELS; ++L) { A.F.lvl[L] = ctx->lvl[L]; A.F.w[L] = ctx->lw[L]; A.F.h[L] = ctx->lh[L]; A.F.sstride[L] = (size_t)ctx->lw[L] * ctx->lh[L]; }
    A.F.n_levels = ctx->prm.pyramid_levels; A.F.cells = ctx->cells; A.F.n_pairs = 1;
    A.F.cam = Cam{ ctx->prm.fx, ctx->prm.fy, ctx->prm.cx, ctx->prm.cy }; A.F.prio = 0;
    A.F.pair_q = A.F.pair_t = A.F.trk_n = nullptr; A.F.pair_T = A.F.trk_px = A.F.trk_depth = nullptr; A.F.trk_level = nullptr; A.F.cand = nullptr;
    A.F.px_cur = nullptr; A.F.search_level = nullptr; A.F.ok = nullptr;
    A.slot1 = slot1; A.slot2 = slot2; A.n = n; A.T = (const double *)buf;
    A.px1 = (const double *)(buf + o_px1); A.level1 = (const int32_t *)(buf + o_lv); A.px2 = (double *)(buf + o_px2);
    A.code = (int32_t *)(buf + o_code); A.depth1 = (double *)(buf + o_d1); A.depth2 = (double *)(buf + o_d2); A.pos_world = (double *)(buf + o_pw);
    A.search_level = (int32_t *)(buf + o_sl);
    YGZ_LAUNCH(ctx, KID_DEPTH_TRI, k_create_map_points, dim3(ygz_div_up(n, 64)), dim3(64), A);
    YGZ_HIPCHK(ctx, hipGetLastError());
    YGZ_HIPCHK(ctx, hipMemcpyAsync(px2, buf + o_px2, N * 16, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(code, buf + o_code, N * 4, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(depth1, buf + o_d1, N * 8, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(depth2, buf + o_d2, N * 8, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(pos_world, buf + o_pw, N * 24, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(search_level, buf + o_sl, N * 4, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (n_created) { int c = 0; for (int i = 0; i < n; ++i) c += code[i] == 0; *n_created = c; }
    return YGZ_OK;
}

extern "C" int ygz_hip_depth_from_triangulation(ygz_hip_ctx *ctx, const double T_search_ref[7], const double *f_ref, const double *f_cur, int n,
                                                double determinant_th, double *depth1, double *depth2, uint8_t *ok)
{
    YgzDeviceGuard dg_(ctx);
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || !T_search_ref || n < 0 || (n > 0 && (!f_ref || !f_cur || !depth1 || !depth2 || !ok))) return YGZ_E_INVALID;
    if (n == 0) return YGZ_OK;
    const size_t N = (size_t)n;
    uint8_t *buf = nullptr;
    int rc = ygz_scratch(ctx, SCR_GEN_0, 64 + N * (24 + 24 + 8 + 8 + 1) + 64, (void **)&buf);
    if (rc != YGZ_OK) return rc;
    double *d_T = (double *)buf, *d_fr = d_T + 8, *d_fc = d_fr + 3 * N, *d_d1 = d_fc + 3 * N, *d_d2 = d_d1 + N;
    uint8_t *d_ok = (uint8_t *)(d_d2 + N);
    YGZ_HIPCHK(ctx, hipMemcpyAsync(d_T, T_search_ref, 56, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(d_fr, f_ref, N * 24, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(d_fc, f_cur, N * 24, hipMemcpyHostToDevice, ctx->stream));
    YGZ_LAUNCH(ctx, KID_DEPTH_TRI, k_depth_from_triangulation, dim3(ygz_div_up(n, 256)), dim3(256), d_T, d_fr, d_fc, n, determinant_th, d_d1, d_d2, d_ok);
    YGZ_HIPCHK(ctx, hipGetLastError());
    YGZ_HIPCHK(ctx, hipMemcpyAsync(depth1, d_d1, N * 8, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(depth2, d_d2, N * 8, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(ok, d_ok, N, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return YGZ_OK;
}

// =====================================================================================================================================
// the SVO depth filter of the legacy tree -- DepthFilter::UpdateSeeds / UpdateSeed / ComputeTau (src/optimizer.cpp:537-735)
// with utils::FindEpipolarMatchDirect, utils::GetWarpAffineMatrix, ZMSSD<4> and the legacy utils::Align2D (src/utils.cpp:37-98,102-281,
// 330-661; include/ygz/utils.h:185-196,288-465).  lane = seed: the seeds of one new frame are independent (the reference walks a
// std::list only to erase from it).  What the legacy tree takes from headers that no longer exist (Frame::InFrame, PinholeCamera::focal)
// comes from their live successors, and the two undefined spots (ZMSSD patches outside a level image, matched_px on paths that never
// assign it) are defined as in oracle/mapping.c -- the kernel follows that restatement decision by decision.

// legacy utils::Align2D (src/utils.cpp:102-281, convergence_condition = false): same float chains as cvutils::Align2D, other stop rules
// (update^2 < 0.001, stop when chi2 grows, accept below 15000)
static __device__ bool align2d_legacy_core(const uint8_t *__restrict__ cur, int w, int h, const uint8_t *pwb, int n_iter, double *pu, double *pv)
{
    float H[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    for (int y = 0; y < 8; ++y)
        for (int x = 0; x < 8; ++x) {
            const int k = (y + 1) * 10 + (x + 1);
            const float jx = 0.5f * (float)((int)pwb[(k + 1) * 64] - (int)pwb[(k - 1) * 64]);
            const float jy = 0.5f * (float)((int)pwb[(k + 10) * 64] - (int)pwb[(k - 10) * 64]);
            H[0] = __fadd_rn(H[0], __fmul_rn(jx, jx)); H[1] = __fadd_rn(H[1], __fmul_rn(jx, jy)); H[2] = __fadd_rn(H[2], jx);
            H[4] = __fadd_rn(H[4], __fmul_rn(jy, jy)); H[5] = __fadd_rn(H[5], jy); H[8] = __fadd_rn(H[8], 1.0f);
        }
    H[3] = H[1]; H[6] = H[2]; H[7] = H[5];
    float Hinv[9];
    {
        const float c0 = cof3(H, 0, 0), c1 = cof3(H, 1, 0), c2 = cof3(H, 2, 0);
        const float det = __fadd_rn(__fadd_rn(__fmul_rn(c0, H[0]), __fmul_rn(c1, H[3])), __fmul_rn(c2, H[6]));
        const float invdet = __fdiv_rn(1.0f, det);
        Hinv[0] = __fmul_rn(c0, invdet); Hinv[1] = __fmul_rn(c1, invdet); Hinv[2] = __fmul_rn(c2, invdet);
        Hinv[3] = __fmul_rn(cof3(H, 0, 1), invdet); Hinv[4] = __fmul_rn(cof3(H, 1, 1), invdet); Hinv[5] = __fmul_rn(cof3(H, 2, 1), invdet);
        Hinv[6] = __fmul_rn(cof3(H, 0, 2), invdet); Hinv[7] = __fmul_rn(cof3(H, 1, 2), invdet); Hinv[8] = __fmul_rn(cof3(H, 2, 2), invdet);
    }
    double first_u = *pu, first_v = *pv;
    float mean_diff = 0.f, u = (float)*pu, v = (float)*pv, last_chi2 = 0.f;
    const float min_update_squared = 0.001f;
    int n_chi2 = 0;
    bool converged = false, error_increased = false;
    for (int iter = 0; iter < n_iter; ++iter) {
        float chi2 = 0.f;
        if (u != u || v != v) return false;
        const int u_r = (int)floorf(u), v_r = (int)floorf(v);
        if (u_r < 4 || v_r < 4 || u_r >= w - 4 || v_r >= h - 4) break;
        const float sx = __fsub_rn(u, (float)u_r), sy = __fsub_rn(v, (float)v_r);
        const float wTL = (float)((1.0 - (double)sx) * (1.0 - (double)sy)), wTR = (float)((double)sx * (1.0 - (double)sy));
        const float wBL = (float)((1.0 - (double)sx) * (double)sy), wBR = __fmul_rn(sx, sy);
        float J0 = 0.f, J1 = 0.f, J2 = 0.f;
        uint32_t wl[9], wh[9], w8[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const uintptr_t a = reinterpret_cast<uintptr_t>(cur + (size_t)(v_r + r - 4) * w + (u_r - 4));
            ygz_gptr32 q = (ygz_gptr32)(a & ~(uintptr_t)3);
            const uint32_t sh = (uint32_t)(a & 3), d0 = q[0], d1 = q[1], d2 = q[2];
            wl[r] = __builtin_amdgcn_alignbyte(d1, d0, sh); wh[r] = __builtin_amdgcn_alignbyte(d2, d1, sh); w8[r] = (d2 >> (8 * sh)) & 255u;
        }
#define WINL(r, c) ((c) < 8 ? YGZ_BYTE(wl[r], wh[r], (c) & 7) : (int)w8[r])
#pragma unroll
        for (int y = 0; y < 8; ++y) {
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const float tl = (float)WINL(y, x), tr = (float)WINL(y, x + 1), bl = (float)WINL(y + 1, x), br = (float)WINL(y + 1, x + 1);
                const float sp = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(wTL, tl), __fmul_rn(wTR, tr)), __fmul_rn(wBL, bl)), __fmul_rn(wBR, br));
                const int k = (y + 1) * 10 + (x + 1);
                const float res = __fadd_rn(__fsub_rn(sp, (float)pwb[k * 64]), mean_diff);
                const float jx = 0.5f * (float)((int)pwb[(k + 1) * 64] - (int)pwb[(k - 1) * 64]);
                const float jy = 0.5f * (float)((int)pwb[(k + 10) * 64] - (int)pwb[(k - 10) * 64]);
                J0 = __fsub_rn(J0, __fmul_rn(res, jx)); J1 = __fsub_rn(J1, __fmul_rn(res, jy)); J2 = __fsub_rn(J2, res);
                chi2 = __fadd_rn(chi2, __fmul_rn(res, res));
            }
        }
#undef WINL
        const float up0 = __fadd_rn(__fadd_rn(__fmul_rn(Hinv[0], J0), __fmul_rn(Hinv[1], J1)), __fmul_rn(Hinv[2], J2));
        const float up1 = __fadd_rn(__fadd_rn(__fmul_rn(Hinv[3], J0), __fmul_rn(Hinv[4], J1)), __fmul_rn(Hinv[5], J2));
        const float up2 = __fadd_rn(__fadd_rn(__fmul_rn(Hinv[6], J0), __fmul_rn(Hinv[7], J1)), __fmul_rn(Hinv[8], J2));
        u = __fadd_rn(u, up0); v = __fadd_rn(v, up1); mean_diff = __fadd_rn(mean_diff, up2);
        if (iter > 0 && chi2 > last_chi2) { error_increased = true; break; }
        last_chi2 = chi2; ++n_chi2;
        if (__fadd_rn(__fmul_rn(up0, up0), __fmul_rn(up1, up1)) < min_update_squared) { first_u = (double)u; first_v = (double)v; converged = true; break; }
    }
    *pu = (double)u; *pv = (double)v;
    if (converged) return true;
    if (n_chi2 == 0) return false;
    if (error_increased) {
        if (last_chi2 < 15000.f) { *pu = first_u; *pv = first_v; return true; }
        return false;
    }
    return last_chi2 < 15000.f;
}

struct DfArgs {
    const uint8_t *lvl[YGZ_MAX_LEVELS];
    int w[YGZ_MAX_LEVELS], h[YGZ_MAX_LEVELS];
    Cam cam;
    int cur_slot, n, batch_counter, max_n_kfs;
    double conv_thresh, px_error_angle;
    const double *T_cur;                          // T_cur (7), T_cur.inverse() (7)
    const int32_t *ref_slot; const double *T_refs;        // [n_refs], [n_refs][7]
    const float *kp; const int32_t *octave, *seed_ref; const unsigned long long *frame_id;
    float *a, *b, *mu, *sigma2; const float *z_range;
    int32_t *state; double *z_out, *matched_px, *pos_world;
};

__device__ __forceinline__ void df_cam2px_unit(const Cam &c, const double uv[2], double px[2])
{ px[0] = (double)c.fx * uv[0] / 1.0 + (double)c.cx; px[1] = (double)c.fy * uv[1] / 1.0 + (double)c.cy; }

// utils::FindEpipolarMatchDirect (src/utils.cpp:330-661)
static __device__ bool df_epipolar_match(const DfArgs &A, int ref_slot, const Se3 &T_ref, const Se3 &T_cur, const double px_ref[2], int octave,
                                         double d_estimate, double d_min, double d_max, uint8_t *pwb, double *depth, double matched_px[2])
{
    Se3 Tri, T_cur_ref;
    se3_inv_d(&T_ref, &Tri); se3_mul_d(&T_cur, &Tri, &T_cur_ref);
    double pt_ref[3], t[3], q[3], Ae[2], Be[2];
    pixel2camera_d(A.cam, px_ref, 1.0, pt_ref);
    for (int k = 0; k < 3; ++k) t[k] = pt_ref[k] * d_min;
    se3_act_d(&T_cur_ref, t, q); Ae[0] = q[0] / q[2]; Ae[1] = q[1] / q[2];
    for (int k = 0; k < 3; ++k) t[k] = pt_ref[k] * d_max;
    se3_act_d(&T_cur_ref, t, q); Be[0] = q[0] / q[2]; Be[1] = q[1] / q[2];
    const double ep0 = Ae[0] - Be[0], ep1 = Ae[1] - Be[1];
    double Am[4];
    {   // utils::GetWarpAffineMatrix (src/utils.cpp:37-64)
        double p3[3], pw[3], pdu[3], pdv[3], pc[2], pu[2], pv[2], c3[3];
        for (int k = 0; k < 3; ++k) p3[k] = pt_ref[k] * d_estimate;
        se3_act_d(&Tri, p3, pw);
        const double s = (double)(1 << octave);
        const double pxu[2] = { px_ref[0] + 4.0 * s, px_ref[1] + 0.0 * s }, pxv[2] = { px_ref[0] + 0.0 * s, px_ref[1] + 4.0 * s };
        double cu[3], cv[3];
        pixel2camera_d(A.cam, pxu, p3[2], cu); pixel2camera_d(A.cam, pxv, p3[2], cv);
        se3_act_d(&Tri, cu, pdu); se3_act_d(&Tri, cv, pdv);
        se3_act_d(&T_cur, pw, c3); camera2pixel_d(A.cam, c3, pc);
        se3_act_d(&T_cur, pdu, c3); camera2pixel_d(A.cam, c3, pu);
        se3_act_d(&T_cur, pdv, c3); camera2pixel_d(A.cam, c3, pv);
        Am[0] = (pu[0] - pc[0]) / 4; Am[2] = (pu[1] - pc[1]) / 4; Am[1] = (pv[0] - pc[0]) / 4; Am[3] = (pv[1] - pc[1]) / 4;
    }
    int sl = 0;
    { double D = Am[0] * Am[3] - Am[2] * Am[1]; while (D > 3.0 && sl < 2) { sl += 1; D *= 0.25; } }            // GetBestSearchLevel(A, 2)
    double px_A[2], px_B[2];
    df_cam2px_unit(A.cam, Ae, px_A); df_cam2px_unit(A.cam, Be, px_B);
    const double dxl = px_A[0] - px_B[0], dyl = px_A[1] - px_B[1];
    const double epi_length = sqrt(dxl * dxl + dyl * dyl) / (double)(1 << sl);
    warp_affine_lds(Am, A.lvl[octave] + (size_t)ref_slot * A.w[octave] * A.h[octave], A.w[octave], A.h[octave], px_ref, octave, sl, pwb);
    const int cw = A.w[sl], ch = A.h[sl];
    const uint8_t *cimg = A.lvl[sl] + (size_t)A.cur_slot * cw * ch;
    const double T7[7] = { T_cur_ref.q[0], T_cur_ref.q[1], T_cur_ref.q[2], T_cur_ref.q[3], T_cur_ref.t[0], T_cur_ref.t[1], T_cur_ref.t[2] };
    double px_cur[2];
    if (epi_length < 2.0) {
        px_cur[0] = (px_A[0] + px_B[0]) / 2.0; px_cur[1] = (px_A[1] + px_B[1]) / 2.0;
        double su = px_cur[0] / (double)(1 << sl), sv = px_cur[1] / (double)(1 << sl);
        const bool res = align2d_legacy_core(cimg, cw, ch, pwb, 10, &su, &sv);
        if (res) {
            px_cur[0] = su * (double)(1 << sl); px_cur[1] = sv * (double)(1 << sl);
            double fc[3], d2;
            pixel2camera_d(A.cam, px_cur, 1.0, fc);
            matched_px[0] = px_cur[0]; matched_px[1] = px_cur[1];
            return tri_depth_d(T7, pt_ref, fc, 1e-5, depth, &d2);
        }
        matched_px[0] = px_cur[0]; matched_px[1] = px_cur[1];
        return false;
    }
    unsigned long long n_steps = (unsigned long long)(epi_length / 0.7);
    const double st0 = ep0 / (double)n_steps, st1 = ep1 / (double)n_steps;
    if (n_steps > 1000) return false;
    int sumA = 0, sumAA = 0;                                   // ZMSSD<4> of the warped reference patch (rows 1..8, columns 1..8 of pwb)
    for (int y = 0; y < 8; ++y) for (int x = 0; x < 8; ++x) { const int p = pwb[((y + 1) * 10 + x + 1) * 64]; sumA += p; sumAA += p * p; }
    int zmssd_best = 2000 * 64;
    double uv[2] = { Be[0] - st0, Be[1] - st1 }, uv_best[2] = { 0, 0 };
    int last_x = 0, last_y = 0;
    ++n_steps;
    for (unsigned long long i = 0; i < n_steps; ++i, uv[0] += st0, uv[1] += st1) {
        double px[2];
        df_cam2px_unit(A.cam, uv, px);
        const int pxi_x = (int)(px[0] / (double)(1 << sl) + 0.5), pxi_y = (int)(px[1] / (double)(1 << sl) + 0.5);
        if (pxi_x == last_x && pxi_y == last_y) continue;
        last_x = pxi_x; last_y = pxi_y;
        const double xx = (double)pxi_x / (double)(1 << sl), yy = (double)pxi_y / (double)(1 << sl);
        if (!(xx >= 8 && xx < A.w[0] - 8 && yy >= 8 && yy < A.h[0] - 8)) continue;
        if (pxi_x - 4 < 0 || pxi_y - 4 < 0 || pxi_x + 4 > cw || pxi_y + 4 > ch) continue;
        const uint8_t *cp = cimg + (size_t)(pxi_y - 4) * cw + (pxi_x - 4);
        int sumB = 0, sumBB = 0, sumAB = 0;
        for (int y = 0; y < 8; ++y) {
            uint32_t lo, hi;
            ygz_load8(cp + (size_t)y * cw, lo, hi);
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const int c = YGZ_BYTE(lo, hi, x), p = pwb[((y + 1) * 10 + x + 1) * 64];
                sumB += c; sumBB += c * c; sumAB += c * p;
            }
        }
        const int zmssd = sumAA - 2 * sumAB + sumBB - (sumA * sumA - 2 * sumA * sumB + sumB * sumB) / 64;
        if (zmssd < zmssd_best) { zmssd_best = zmssd; uv_best[0] = uv[0]; uv_best[1] = uv[1]; }
    }
    if (zmssd_best < 2000 * 64) {
        df_cam2px_unit(A.cam, uv_best, px_cur);
        double su = px_cur[0] / (double)(1 << sl), sv = px_cur[1] / (double)(1 << sl);
        const bool res = align2d_legacy_core(cimg, cw, ch, pwb, 10, &su, &sv);
        if (res) {
            px_cur[0] = su * (double)(1 << sl); px_cur[1] = sv * (double)(1 << sl);
            double fc[3], d2;
            pixel2camera_d(A.cam, px_cur, 1.0, fc);
            matched_px[0] = px_cur[0]; matched_px[1] = px_cur[1];
            return tri_depth_d(T7, pt_ref, fc, 1e-5, depth, &d2);
        }
        matched_px[0] = px_cur[0]; matched_px[1] = px_cur[1];
    }
    return false;
}

__global__ __launch_bounds__(64) void k_depth_filter(DfArgs A)
{
    __shared__ uint8_t pwb_all[100 * 64];
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= A.n) return;
    A.z_out[i] = 0.0; A.matched_px[2 * (size_t)i] = 0.0; A.matched_px[2 * (size_t)i + 1] = 0.0;
    if (((unsigned long long)(long long)A.batch_counter - A.frame_id[i]) > (unsigned long long)(long long)A.max_n_kfs) { A.state[i] = 4; return; }
    const int r = A.seed_ref[i];
    Se3 T_ref, T_cur, T_cur_inv, T_ref_cur, T_ref_cur_inv;
    for (int k = 0; k < 4; ++k) { T_ref.q[k] = A.T_refs[7 * (size_t)r + k]; T_cur.q[k] = A.T_cur[k]; T_cur_inv.q[k] = A.T_cur[7 + k]; }
    for (int k = 0; k < 3; ++k) { T_ref.t[k] = A.T_refs[7 * (size_t)r + 4 + k]; T_cur.t[k] = A.T_cur[4 + k]; T_cur_inv.t[k] = A.T_cur[11 + k]; }
    se3_mul_d(&T_ref, &T_cur_inv, &T_ref_cur); se3_inv_d(&T_ref_cur, &T_ref_cur_inv);
    const double px_ref[2] = { (double)A.kp[2 * (size_t)i], (double)A.kp[2 * (size_t)i + 1] };
    float a = A.a[i], b = A.b[i], mu = A.mu[i], sigma2 = A.sigma2[i];
    const float z_range = A.z_range[i];
    double pt_ref[3], sc[3], xyz_f[3];
    pixel2camera_d(A.cam, px_ref, 1.0, pt_ref);
    for (int k = 0; k < 3; ++k) sc[k] = 1.0 / (double)mu * pt_ref[k];
    se3_act_d(&T_ref_cur_inv, sc, xyz_f);
    if (xyz_f[2] < 0.0) { A.state[i] = 1; return; }
    {
        double uvp[2];
        camera2pixel_d(A.cam, xyz_f, uvp);
        if (!(uvp[0] >= 10 && uvp[0] < A.w[0] - 10 && uvp[1] >= 10 && uvp[1] < A.h[0] - 10)) { A.state[i] = 2; return; }
    }
    const float ssig = ygz_sqrtf_cr(sigma2);
    const float z_inv_min = __fadd_rn(mu, ssig);
    const float z_inv_max = fmaxf(__fsub_rn(mu, ssig), 0.00000001f);
    double z = 0.0, mpx[2] = { 0.0, 0.0 };
    if (!df_epipolar_match(A, A.ref_slot[r], T_ref, T_cur, px_ref, A.octave[i], 0.9 / (double)mu, 1.1 / (double)z_inv_min, 1.0 / (double)z_inv_max,
                           pwb_all + threadIdx.x, &z, mpx)) {
        A.matched_px[2 * (size_t)i] = mpx[0]; A.matched_px[2 * (size_t)i + 1] = mpx[1];
        A.state[i] = 3; return;
    }
    A.matched_px[2 * (size_t)i] = mpx[0]; A.matched_px[2 * (size_t)i + 1] = mpx[1];
    A.z_out[i] = z;
    double tau;
    {   // DepthFilter::ComputeTau (src/optimizer.cpp:711-726)
        const double *t = T_ref_cur.t;
        const double av[3] = { pt_ref[0] * z - t[0], pt_ref[1] * z - t[1], pt_ref[2] * z - t[2] };
        const double t_norm = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]), a_norm = sqrt(av[0] * av[0] + av[1] * av[1] + av[2] * av[2]);
        const double alpha = acos((pt_ref[0] * t[0] + pt_ref[1] * t[1] + pt_ref[2] * t[2]) / t_norm);
        const double beta = acos((av[0] * -t[0] + av[1] * -t[1] + av[2] * -t[2]) / (t_norm * a_norm));
        const double beta_plus = beta + A.px_error_angle;
        const double gamma_plus = 3.14159265358979323846 - alpha - beta_plus;
        tau = t_norm * sin(beta_plus) / sin(gamma_plus) - z;
    }
    const double tau_inverse = 0.5 * (1.0 / fmax(0.0000001, z - tau) - 1.0 / (z + tau));
    {   // DepthFilter::UpdateSeed (src/optimizer.cpp:683-708), float throughout
        const float x = (float)(1. / z), tau2 = (float)(tau_inverse * tau_inverse);
        const float norm_scale = ygz_sqrtf_cr(__fadd_rn(sigma2, tau2));
        if (!(norm_scale != norm_scale)) {
            float e_ = __fsub_rn(x, mu); e_ = __fmul_rn(e_, -e_); e_ = __fdiv_rn(e_, __fmul_rn(__fmul_rn(2.f, norm_scale), norm_scale));
            const float pdf = __fdiv_rn((float)ygz_exp_nonpos((double)e_) /* = oracle/mapping.c::yo_expf_cr */, __fmul_rn(norm_scale, ygz_sqrtf_cr(__fmul_rn(2.f, 3.14159265358979323846f))));
            const float s2 = (float)(1. / (1. / (double)sigma2 + 1. / (double)tau2));
            const float m = __fmul_rn(s2, __fadd_rn(__fdiv_rn(mu, sigma2), __fdiv_rn(x, tau2)));
            float C1 = __fmul_rn(__fdiv_rn(a, __fadd_rn(a, b)), pdf);
            float C2 = (float)((double)__fdiv_rn(b, __fadd_rn(a, b)) * 1. / (double)z_range);
            const float nc = __fadd_rn(C1, C2);
            C1 = __fdiv_rn(C1, nc); C2 = __fdiv_rn(C2, nc);
            const float ab = __fadd_rn(a, b);
            const float f = (float)((double)C1 * ((double)a + 1.) / ((double)ab + 1.) + (double)__fmul_rn(C2, a) / ((double)ab + 1.));
            const float e = (float)((double)C1 * ((double)a + 1.) * ((double)a + 2.) / (((double)ab + 1.) * ((double)ab + 2.))
                                    + (double)__fdiv_rn(__fmul_rn(__fmul_rn(C2, a), __fadd_rn(a, 1.0f)), __fmul_rn(__fadd_rn(ab, 1.0f), __fadd_rn(ab, 2.0f))));
            const float mu_new = __fadd_rn(__fmul_rn(C1, m), __fmul_rn(C2, mu));
            sigma2 = __fsub_rn(__fadd_rn(__fmul_rn(C1, __fadd_rn(s2, __fmul_rn(m, m))), __fmul_rn(C2, __fadd_rn(sigma2, __fmul_rn(mu, mu)))), __fmul_rn(mu_new, mu_new));
            mu = mu_new;
            a = __fdiv_rn(__fsub_rn(e, f), __fsub_rn(f, __fdiv_rn(e, f)));
            b = __fdiv_rn(__fmul_rn(a, __fsub_rn(1.0f, f)), f);
        }
    }
    A.a[i] = a; A.b[i] = b; A.mu[i] = mu; A.sigma2[i] = sigma2;
    if ((double)ygz_sqrtf_cr(sigma2) < (double)z_range / A.conv_thresh) {
        double p[3], pw[3];
        for (int k = 0; k < 3; ++k) p[k] = pt_ref[k] * (1.0 / (double)mu);
        se3_act_d(&T_cur_inv, p, pw);                       // frame->_T_c_w.inverse() * (pt_ref / mu), src/optimizer.cpp:628 as written
        A.pos_world[3 * (size_t)i] = pw[0]; A.pos_world[3 * (size_t)i + 1] = pw[1]; A.pos_world[3 * (size_t)i + 2] = pw[2];
        A.state[i] = 5;
    } else if (z_inv_min != z_inv_min) A.state[i] = 6;
    else A.state[i] = 0;
}

extern "C" int ygz_hip_depth_filter_update(ygz_hip_ctx *ctx, int cur_slot, const double T_cur[7], int n_refs, const int32_t *ref_slot,
                                           const double *T_refs, int batch_counter, int max_n_kfs, double convergence_sigma2_thresh, int n,
                                           const float *kp, const int32_t *octave, const int32_t *seed_ref, const uint64_t *seed_frame_id,
                                           float *a, float *b, float *mu, const float *z_range, float *sigma2, int32_t *state, double *z,
                                           double *matched_px, double *pos_world, int *n_updated)
{
    YgzDeviceGuard dg_(ctx);
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || !T_cur || n < 0 || n_refs < 0 || cur_slot < 0 || cur_slot >= ctx->prm.max_frames) return YGZ_E_INVALID;
    if (n_updated) *n_updated = 0;
    if (n == 0) return YGZ_OK;
    if (n_refs < 1 || !ref_slot || !T_refs || !kp || !octave || !seed_ref || !seed_frame_id || !a || !b || !mu || !z_range || !sigma2 || !state || !z ||
        !matched_px || !pos_world) return YGZ_E_INVALID;
    if (!ctx->pyr_valid[cur_slot]) return YGZ_E_STATE;
    for (int r = 0; r < n_refs; ++r) {
        if (ref_slot[r] < 0 || ref_slot[r] >= ctx->prm.max_frames) return YGZ_E_INVALID;
        if (!ctx->pyr_valid[ref_slot[r]]) return YGZ_E_STATE;
    }
    for (int i = 0; i < n; ++i)
        if (seed_ref[i] < 0 || seed_ref[i] >= n_refs || octave[i] < 0 || octave[i] >= ctx->prm.pyramid_levels) return YGZ_E_INVALID;
    if (ctx->prm.pyramid_levels < 3) return YGZ_E_STATE;              // GetBestSearchLevel(A, 2) may pick level 2
    const size_t N = (size_t)n, R = (size_t)n_refs;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 15) & ~(size_t)15; return o; };
    const size_t o_T = take(14 * 8), o_rs = take(R * 4), o_Tr = take(R * 56), o_kp = take(N * 8), o_oc = take(N * 4), o_sr = take(N * 4), o_fid = take(N * 8),
                 o_a = take(N * 4), o_b = take(N * 4), o_mu = take(N * 4), o_zr = take(N * 4), o_s2 = take(N * 4), o_st = take(N * 4), o_z = take(N * 8),
                 o_mp = take(N * 16), o_pw = take(N * 24);
    uint8_t *buf = nullptr;
    int rc = ygz_scratch(ctx, SCR_GEN_0, off + 64, (void **)&buf);
    if (rc != YGZ_OK) return rc;
    double hT[14];
    {
        Se3 c, ci;
        for (int k = 0; k < 4; ++k) c.q[k] = T_cur[k];
        for (int k = 0; k < 3; ++k) c.t[k] = T_cur[4 + k];
        se3_inv_d(&c, &ci);
        for (int k = 0; k < 7; ++k) hT[k] = T_cur[k];
        for (int k = 0; k < 4; ++k) hT[7 + k] = ci.q[k];
        for (int k = 0; k < 3; ++k) hT[11 + k] = ci.t[k];
    }
#define UPD_(o, src, bytes) YGZ_HIPCHK(ctx, hipMemcpyAsync(buf + (o), (src), (bytes), hipMemcpyHostToDevice, ctx->stream))
    UPD_(o_T, hT, sizeof(hT)); UPD_(o_rs, ref_slot, R * 4); UPD_(o_Tr, T_refs, R * 56); UPD_(o_kp, kp, N * 8); UPD_(o_oc, octave, N * 4);
    UPD_(o_sr, seed_ref, N * 4); UPD_(o_fid, seed_frame_id, N * 8); UPD_(o_a, a, N * 4); UPD_(o_b, b, N * 4); UPD_(o_mu, mu, N * 4);
    UPD_(o_zr, z_range, N * 4); UPD_(o_s2, sigma2, N * 4);
#undef UPD_
    YGZ_HIPCHK(ctx, hipMemsetAsync(buf + o_pw, 0, N * 24, ctx->stream));
    DfArgs A;
    for (int L = 0; L < YGZ_MAX_LEVELS; ++L) { A.lvl[L] = ctx->lvl[L]; A.w[L] = ctx->lw[L]; A.h[L] = ctx->lh[L]; }
    A.cam = Cam{ ctx->prm.fx, ctx->prm.fy, ctx->prm.cx, ctx->prm.cy };
    A.cur_slot = cur_slot; A.n = n; A.batch_counter = batch_counter; A.max_n_kfs = max_n_kfs; A.conv_thresh = convergence_sigma2_thresh;
    {   // px_error_angle = atan(px_noise / (2 focal)) * 2, focal = float (fx + fy) / 2 (Camera.h:24)
        const double focal_length = (double)((ctx->prm.fx + ctx->prm.fy) / 2);
        A.px_error_angle = atan(1.0 / (2.0 * focal_length)) * 2.0;
    }
    A.T_cur = (const double *)(buf + o_T); A.ref_slot = (const int32_t *)(buf + o_rs); A.T_refs = (const double *)(buf + o_Tr);
    A.kp = (const float *)(buf + o_kp); A.octave = (const int32_t *)(buf + o_oc); A.seed_ref = (const int32_t *)(buf + o_sr);
    A.frame_id = (const unsigned long long *)(buf + o_fid);
    A.a = (float *)(buf + o_a); A.b = (float *)(buf + o_b); A.mu = (float *)(buf + o_mu); A.z_range = (const float *)(buf + o_zr); A.sigma2 = (float *)(buf + o_s2);
    A.state = (int32_t *)(buf + o_st); A.z_out = (double *)(buf + o_z); A.matched_px = (double *)(buf + o_mp); A.pos_world = (double *)(buf + o_pw);
    YGZ_LAUNCH(ctx, KID_DEPTH_FILTER, k_depth_filter, dim3(ygz_div_up(n, 64)), dim3(64), A);
    YGZ_HIPCHK(ctx, hipGetLastError());
#define DNL_(dst, o, bytes) YGZ_HIPCHK(ctx, hipMemcpyAsync((dst), buf + (o), (bytes), hipMemcpyDeviceToHost, ctx->stream))
    DNL_(a, o_a, N * 4); DNL_(b, o_b, N * 4); DNL_(mu, o_mu, N * 4); DNL_(sigma2, o_s2, N * 4); DNL_(state, o_st, N * 4); DNL_(z, o_z, N * 8);
    DNL_(matched_px, o_mp, N * 16); DNL_(pos_world, o_pw, N * 24);
#undef DNL_
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (n_updated) { int c = 0; for (int i = 0; i < n; ++i) c += state[i] == 0 || state[i] == 5 || state[i] == 6; *n_updated = c; }
    return YGZ_OK;
}
