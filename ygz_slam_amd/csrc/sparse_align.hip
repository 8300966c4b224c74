// L3 -- SVO-style sparse image alignment, whole Gauss-Newton loop resident on the GPU.
// Replaces SparseImgAlign::run / precomputeReferencePatches / computeResiduals / solve / update
// (src/Algorithm/SparseImageAlign.cpp:21-238) and NLLSSolver::optimizeGaussNewton
// (include/ygz/Algorithm/NLSSolver_impl.hpp:15-89).
//
// One workgroup (1024 lanes) per alignment problem; lane = feature (4x4 patch).  Per iteration:
//   lanes warp their feature, gather the 5x5 current-image window, form 16 residuals, accumulate
//   their own H (21 unique) / Jres (6) in FP64 registers, and publish res^2;
//   H/Jres are reduced in a fixed tree order (wave shuffles, then 16 wave partials in order);
//   chi2 is the reference's FLOAT sum in feature/pixel order: it decides "error increased ->
//   rollback" (NLSSolver_impl.hpp:53-63), so it is reproduced exactly -- res^2 staged through LDS
//   and summed by one lane in raster order (adding the 0.0f of a skipped feature is exact);
//   lane 0 solves the 6x6 LDLT, applies T <- T * exp(-x), and broadcasts the loop decision.
// No host round trip per iteration (the reference solves ~30 6x6 systems per frame pair).
#include "ygz_internal.h"
#include "se3_dev.h"
#include "ldlt6.h"

#define SA_THREADS 1024
#define SA_CHUNK   4096          // floats of res^2 staged in LDS per pass

struct SaArgs {
    const uint8_t *lvl[YGZ_MAX_LEVELS];
    int w[YGZ_MAX_LEVELS], h[YGZ_MAX_LEVELS];
    int ref_slot, cur_slot;
    Se3 T_ref, T_cur;
    float fx, fy, cx, cy;
    const double *px, *depth; const uint8_t *has_mp;
    int n, max_level, min_level, n_iter;
    float *patch_cache;      // [n][16]
    double *jac_cache;       // [n][16][6]
    uint8_t *visible;        // [n]
    float *r2;               // [n][16]
    double *out;             // [7 pose][1 n_meas][YGZ_MAX_LEVELS iters]
};

__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__global__ __launch_bounds__(SA_THREADS) void k_sparse_align(SaArgs A)
{
    __shared__ double red[16][28];
    __shared__ float stage[SA_CHUNK];
    __shared__ Se3 sT;
    __shared__ int s_ctl;                 // 0 continue, 1 leave level
    __shared__ int s_nmeas_w[16];
    __shared__ float s_chi2;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = A.n;
    // thread-0 solver state (NLLSSolver::reset, NLSSolver_impl.hpp:283-293)
    double chi2_ = 1e10; bool stop_ = false;
    Se3 old_model;
    int n_meas_last = 0;

    if (tid == 0) {
        Se3 Tri; se3_inv_d(&A.T_ref, &Tri);
        se3_mul_d(&A.T_cur, &Tri, &sT);                 // T_cur_from_ref (SparseImageAlign.cpp:37)
        for (int l = 0; l < YGZ_MAX_LEVELS; ++l) A.out[8 + l] = 0;
    }
    for (int f = tid; f < n; f += SA_THREADS) A.visible[f] = 0;
    for (int i = tid; i < n * 16; i += SA_THREADS) A.patch_cache[i] = 0.f;
    __syncthreads();

    for (int level = A.max_level; level >= A.min_level; --level) {
        const int cols = A.w[level], rows = A.h[level];
        const uint8_t *ref_img = A.lvl[level] + (size_t)A.ref_slot * cols * rows;
        const uint8_t *cur_img = A.lvl[level] + (size_t)A.cur_slot * cols * rows;
        const float scale = 1.0f / (float)(1 << level);
        const int border = 3;
        // jacobian_cache_.setZero(); have_ref_patch_cache_ = false (:42-43): each lane clears the
        // cache rows of the features it owns (the rows it does not refill below stay zero)
        // precomputeReferencePatches (:59-122)
        const double focal = (double)((A.fx + A.fy) / 2);
        for (int f = tid; f < n; f += SA_THREADS) {
            const double pxx = A.px[2 * f], pxy = A.px[2 * f + 1];
            const float u_ref = (float)(pxx * scale), v_ref = (float)(pxy * scale);
            const int ui = (int)floorf(u_ref), vi = (int)floorf(v_ref);
            if (!A.has_mp[f] || ui - border < 0 || vi - border < 0 || ui + border >= cols || vi + border >= rows) {
                double *jz = A.jac_cache + 96 * (size_t)f;
                for (int k = 0; k < 96; ++k) jz[k] = 0.0;
                continue;
            }
            A.visible[f] = 1;
            const double dep = A.depth[f];
            const double x = (pxx - A.cx) * dep / A.fx, y = (pxy - A.cy) * dep / A.fy;
            const double z_inv = 1. / dep, z_inv_2 = z_inv * z_inv;
            double fj[12];      // cvutils::JacobXYZ2Cam (CVUtils.h:77-99)
            fj[0] = -z_inv; fj[1] = 0.0; fj[2] = x * z_inv_2; fj[3] = y * fj[2]; fj[4] = -(1.0 + x * fj[2]); fj[5] = y * z_inv;
            fj[6] = 0.0; fj[7] = -z_inv; fj[8] = y * z_inv_2; fj[9] = 1.0 + y * fj[8]; fj[10] = -fj[3]; fj[11] = -x * z_inv;
            const float su = __fsub_rn(u_ref, (float)ui), sv = __fsub_rn(v_ref, (float)vi);
            const float w_tl = (float)((1.0 - (double)su) * (1.0 - (double)sv)), w_tr = (float)((double)su * (1.0 - (double)sv));
            const float w_bl = (float)((1.0 - (double)su) * (double)sv), w_br = __fmul_rn(su, sv);
            const double fl = focal / (double)(1 << level);
            int pc = 0;
            for (int yy = 0; yy < 4; ++yy) {
                const uint8_t *p = ref_img + (size_t)(vi + yy - 2) * cols + (ui - 2);
                for (int xx = 0; xx < 4; ++xx, ++p, ++pc) {
#define BIL(a, b, c, d) __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(w_tl, (float)(a)), __fmul_rn(w_tr, (float)(b))), __fmul_rn(w_bl, (float)(c))), __fmul_rn(w_br, (float)(d)))
                    A.patch_cache[16 * (size_t)f + pc] = BIL(p[0], p[1], p[cols], p[cols + 1]);
                    const float dx = __fmul_rn(0.5f, __fsub_rn(BIL(p[1], p[2], p[cols + 1], p[cols + 2]), BIL(p[-1], p[0], p[cols - 1], p[cols])));
                    const float dy = __fmul_rn(0.5f, __fsub_rn(BIL(p[cols], p[1 + cols], p[cols * 2], p[cols * 2 + 1]), BIL(p[-cols], p[1 - cols], p[0], p[1])));
                    double *jc = A.jac_cache + 6 * ((size_t)f * 16 + pc);
#pragma unroll
                    for (int k = 0; k < 6; ++k) jc[k] = ((double)dx * fj[k] + (double)dy * fj[6 + k]) * fl;
                }
            }
        }
        __syncthreads();          // caches visible to the whole workgroup (global writes + barrier, same CU)
        if (tid == 0) old_model = sT;
        int it = 0;
        for (; it < A.n_iter; ++it) {
            // ---- computeResiduals(model, linearize=true) (:124-223)
            const Se3 T = sT;
            double acc[27];
#pragma unroll
            for (int k = 0; k < 27; ++k) acc[k] = 0.0;
            int my_meas = 0;
            for (int f = tid; f < n; f += SA_THREADS) {
                float r2v[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) r2v[k] = 0.f;
                bool use = A.visible[f] != 0;
                float res[16];
                if (use) {
                    const double pxx = A.px[2 * f], pxy = A.px[2 * f + 1], dep = A.depth[f];
                    const double xyz_ref[3] = { (pxx - A.cx) * dep / A.fx, (pxy - A.cy) * dep / A.fy, dep };
                    double xyz_cur[3];
                    se3_act_d(&T, xyz_ref, xyz_cur);
                    const double pu = A.fx * xyz_cur[0] / xyz_cur[2] + A.cx, pv = A.fy * xyz_cur[1] / xyz_cur[2] + A.cy;
                    const float u_cur = __fmul_rn((float)pu, scale), v_cur = __fmul_rn((float)pv, scale);
                    const int ui = (int)floorf(u_cur), vi = (int)floorf(v_cur);
                    if (u_cur != u_cur || v_cur != v_cur || ui < 0 || vi < 0 || ui - border < 0 || vi - border < 0 ||
                        ui + border >= cols || vi + border >= rows) use = false;
                    else {
                        const float su = __fsub_rn(u_cur, (float)ui), sv = __fsub_rn(v_cur, (float)vi);
                        const float w_tl = (float)((1.0 - (double)su) * (1.0 - (double)sv)), w_tr = (float)((double)su * (1.0 - (double)sv));
                        const float w_bl = (float)((1.0 - (double)su) * (double)sv), w_br = __fmul_rn(su, sv);
                        const float *cache = A.patch_cache + 16 * (size_t)f;
                        int pc = 0;
                        for (int yy = 0; yy < 4; ++yy) {
                            const uint8_t *p = cur_img + (size_t)(vi + yy - 2) * cols + (ui - 2);
#pragma unroll
                            for (int xx = 0; xx < 4; ++xx, ++pc) {
                                const float ic = BIL(p[xx], p[xx + 1], p[cols + xx], p[cols + xx + 1]);
                                const float r = __fsub_rn(ic, cache[pc]);
                                res[pc] = r;
                                r2v[pc] = __fmul_rn(__fmul_rn(r, r), 1.0f);
                            }
                        }
                    }
                }
                float4 *dst = reinterpret_cast<float4 *>(A.r2 + 16 * (size_t)f);
                dst[0] = make_float4(r2v[0], r2v[1], r2v[2], r2v[3]);     dst[1] = make_float4(r2v[4], r2v[5], r2v[6], r2v[7]);
                dst[2] = make_float4(r2v[8], r2v[9], r2v[10], r2v[11]);   dst[3] = make_float4(r2v[12], r2v[13], r2v[14], r2v[15]);
                if (use) {
                    my_meas += 16;
                    const double *Jc = A.jac_cache + 96 * (size_t)f;
                    for (int pc = 0; pc < 16; ++pc) {
                        double J[6];
#pragma unroll
                        for (int k = 0; k < 6; ++k) J[k] = Jc[6 * pc + k];
                        const double r = (double)res[pc];
                        int q = 0;
#pragma unroll
                        for (int a = 0; a < 6; ++a) {
#pragma unroll
                            for (int b = a; b < 6; ++b) acc[q++] += J[a] * J[b];
                        }
#pragma unroll
                        for (int a = 0; a < 6; ++a) acc[21 + a] -= J[a] * r;
                    }
                }
            }
#undef BIL
            // ---- fixed-order reduction of H / Jres / n_meas
#pragma unroll
            for (int k = 0; k < 27; ++k) { const double s = wave_sum_d(acc[k]); if (lane == 0) red[wv][k] = s; }
            {
                int m = my_meas;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) m += __shfl_xor(m, off);
                if (lane == 0) s_nmeas_w[wv] = m;
            }
            if (tid == 0) s_chi2 = 0.0f;
            __syncthreads();
            // ---- chi2: float sum in feature/pixel order, staged through LDS
            for (int c0 = 0; c0 < n * 16; c0 += SA_CHUNK) {
                const int cnt = min(SA_CHUNK, n * 16 - c0);
                for (int i = tid; i < cnt; i += SA_THREADS) stage[i] = A.r2[c0 + i];
                __syncthreads();
                if (tid == 0) {
                    float c = s_chi2;
                    for (int i = 0; i < cnt; i += 4) {
                        const float4 q4 = *reinterpret_cast<const float4 *>(&stage[i]);
                        c = __fadd_rn(c, q4.x); c = __fadd_rn(c, q4.y); c = __fadd_rn(c, q4.z); c = __fadd_rn(c, q4.w);
                    }
                    s_chi2 = c;
                }
                __syncthreads();
            }
            // ---- lane 0: solve, decide, update (NLSSolver_impl.hpp:40-87)
            if (tid == 0) {
                double Hm[36], Jr[6], x[6];
                int q = 0;
                for (int a = 0; a < 6; ++a) for (int b = a; b < 6; ++b) {
                    double s = 0; for (int w2 = 0; w2 < 16; ++w2) s += red[w2][q];
                    Hm[6 * a + b] = s; Hm[6 * b + a] = s; ++q;
                }
                for (int a = 0; a < 6; ++a) { double s = 0; for (int w2 = 0; w2 < 16; ++w2) s += red[w2][21 + a]; Jr[a] = s; }
                int nm = 0; for (int w2 = 0; w2 < 16; ++w2) nm += s_nmeas_w[w2];
                n_meas_last = nm;
                const double new_chi2 = (double)__fdiv_rn(s_chi2, (float)nm);
                if (!ldlt6_solve_d(Hm, Jr, x)) stop_ = true;
                int ctl = 0;
                if ((it > 0 && new_chi2 > chi2_) || stop_) { sT = old_model; ctl = 1; }
                else {
                    double mx[6]; for (int k = 0; k < 6; ++k) mx[k] = -x[k];
                    Se3 E, Tn;
                    se3_exp_d(mx, &E);
                    se3_mul_d(&T, &E, &Tn);
                    old_model = T; sT = Tn;
                    chi2_ = new_chi2;
                    double nmx = -1; for (int k = 0; k < 6; ++k) if (fabs(x[k]) > nmx) nmx = fabs(x[k]);
                    if (nmx <= 0.000001) ctl = 2;          // eps_ (SparseImageAlign.cpp:18)
                }
                s_ctl = ctl;
            }
            __syncthreads();
            const int ctl = s_ctl;
            __syncthreads();
            if (ctl == 2) { ++it; break; }
            if (ctl == 1) break;
        }
        if (tid == 0 && level < YGZ_MAX_LEVELS) A.out[8 + level] = (double)it;
    }
    if (tid == 0) {
        Se3 o; se3_mul_d(&sT, &A.T_ref, &o);                 // cur->_TCW = T_cur_from_ref * ref->_TCW (:48)
        for (int k = 0; k < 4; ++k) A.out[k] = o.q[k];
        for (int k = 0; k < 3; ++k) A.out[4 + k] = o.t[k];
        A.out[7] = (double)n_meas_last;
    }
}

static void se3_from7(const double *a, Se3 *T) { for (int k = 0; k < 4; ++k) T->q[k] = a[k]; for (int k = 0; k < 3; ++k) T->t[k] = a[4 + k]; }

extern "C" int ygz_hip_sparse_align(ygz_hip_ctx *ctx, int ref_slot, const double T_ref[7], int cur_slot, double T_cur[7],
                                    const double *px, const double *depth, const uint8_t *has_mappoint, int n,
                                    int max_level, int min_level, int n_iter, int *n_meas_out, int *iters_out)
{
    if (!ctx || !T_ref || !T_cur || n < 0 || ref_slot < 0 || ref_slot >= ctx->prm.max_frames || cur_slot < 0 ||
        cur_slot >= ctx->prm.max_frames || min_level < 0 || max_level < min_level || max_level >= ctx->prm.pyramid_levels || n_iter < 0)
        return YGZ_E_INVALID;
    if (n_meas_out) *n_meas_out = 0;
    if (n == 0) return YGZ_OK;                                // run() returns 0 without touching the pose (:25-29)
    if (!px || !depth || !has_mappoint) return YGZ_E_INVALID;
    if (!ctx->pyr_valid[ref_slot] || !ctx->pyr_valid[cur_slot]) return YGZ_E_STATE;
    const size_t N = (size_t)n;
    uint8_t *in = nullptr, *work = nullptr; double *out = nullptr;
    int rc = ygz_scratch(ctx, SCR_SA_IN, N * (16 + 8 + 1) + 64, (void **)&in);
    if (rc == YGZ_OK) rc = ygz_scratch(ctx, SCR_SA_WORK, N * (64 + 768 + 64 + 1) + 256, (void **)&work);
    if (rc == YGZ_OK) rc = ygz_scratch(ctx, SCR_SA_OUT, (8 + YGZ_MAX_LEVELS) * 8, (void **)&out);
    if (rc != YGZ_OK) return rc;
    double *d_px = (double *)in, *d_dep = d_px + 2 * N; uint8_t *d_mp = (uint8_t *)(d_dep + N);
    YGZ_HIPCHK(ctx, hipMemcpyAsync(d_px, px, N * 16, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(d_dep, depth, N * 8, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(d_mp, has_mappoint, N, hipMemcpyHostToDevice, ctx->stream));
    SaArgs A;
    for (int L = 0; L < YGZ_MAX_LEVELS; ++L) { A.lvl[L] = ctx->lvl[L]; A.w[L] = ctx->lw[L]; A.h[L] = ctx->lh[L]; }
    A.ref_slot = ref_slot; A.cur_slot = cur_slot;
    se3_from7(T_ref, &A.T_ref); se3_from7(T_cur, &A.T_cur);
    A.fx = ctx->prm.fx; A.fy = ctx->prm.fy; A.cx = ctx->prm.cx; A.cy = ctx->prm.cy;
    A.px = d_px; A.depth = d_dep; A.has_mp = d_mp; A.n = n; A.max_level = max_level; A.min_level = min_level; A.n_iter = n_iter;
    A.jac_cache = (double *)work; A.patch_cache = (float *)(A.jac_cache + 96 * N); A.r2 = A.patch_cache + 16 * N;
    A.visible = (uint8_t *)(A.r2 + 16 * N); A.out = out;
    hipLaunchKernelGGL(k_sparse_align, dim3(1), dim3(SA_THREADS), 0, ctx->stream, A);
    YGZ_HIPCHK(ctx, hipGetLastError());
    double h_out[8 + YGZ_MAX_LEVELS];
    YGZ_HIPCHK(ctx, hipMemcpyAsync(h_out, out, sizeof(h_out), hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 7; ++k) T_cur[k] = h_out[k];
    if (n_meas_out) *n_meas_out = (int)(h_out[7] / 16);        // run() returns n_meas_/patch_area_ (:49)
    if (iters_out) for (int l = 0; l < ctx->prm.pyramid_levels; ++l) iters_out[l] = (int)h_out[8 + l];
    return YGZ_OK;
}
