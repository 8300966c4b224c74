// SURVEY 8f-1 -- the whole Levenberg-Marquardt loop of ba::LocalBAG2O (src/Algorithm/BA.cpp:390-395,501-502: g2o
// OptimizationAlgorithmLevenberg + BlockSolver_6_3 with marginalised points + a Cholesky solve of the reduced pose system)
// resident on the GPU: one workgroup per BA window runs linearisation, Schur complement, Cholesky, back-substitution, state
// update, trial evaluation and the lambda policy without a host round trip, so hundreds of windows optimise concurrently
// (one per CU).  The host-loop form (ba_lm.hip::ygz_hip_ba_optimize) keeps the same arithmetic with the reduced system on the
// CPU; oracle/ceres_ba.c::yo_g2o_lm restates it for the tests [frozen spec of g2o, see there].
//
// Per LM trial and window (K <= 16 poses, 14 of them free; P points; E edges):
//   1. Dinv_l = (Hll_l + lambda I)^-1 and Y_e = Hpl_e Dinv_l per point (lane = point).
//   2. S = blockdiag(Hpp + lambda I) - sum_l Y_a(l) Hpl_b(l)^T: one wavefront per pose pair (a <= b) sweeps the points 64 at a
//      time through the (point, pose) -> edge table, accumulates the 6x6 block in registers and reduces it in a fixed order
//      (no floating-point atomics); S lives in LDS (84 x 84 doubles).
//   3. right-looking Cholesky in LDS (the same subtraction order per element as the host's left-looking loop), column-oriented
//      forward / backward substitution inside one wavefront.
//   4. x_l = Dinv_l (b_l - sum_e Hpl_e^T x_p) (lane = point), oplus on the poses (lane = pose), trial chi2, rho, lambda.
#include "ba_dev.h"
#include <vector>
#include <string.h>
#include <float.h>

#define LM_THREADS 1024
#define LM_WAVES   (LM_THREADS / 64)
#define LM_MAXKF   14
#define LM_MAXN    (6 * LM_MAXKF)

#define lm_wave_sum ygz_wave_sum_d
// block-wide sum in a fixed order (xor tree inside the wavefronts, then the wavefronts left to right); result on every lane
__device__ __forceinline__ double lm_block_sum(double v, double *red /*[LM_WAVES]*/)
{
    v = lm_wave_sum(v);
    __syncthreads();                                   // red is free again
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < LM_WAVES; ++w) s += red[w];
    return s;
}
__device__ __forceinline__ double lm_block_max(double v, double *red)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < LM_WAVES; ++w) s = fmax(s, red[w]);
    return s;
}

__device__ __forceinline__ bool lm_inv3(const double *m, double *r)
{
    const double c0 = m[4] * m[8] - m[5] * m[7], c1 = m[5] * m[6] - m[3] * m[8], c2 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c0 + m[1] * c1 + m[2] * c2;
    if (!(fabs(det) > 0) || !isfinite(det)) return false;
    const double id = 1.0 / det;
    r[0] = c0 * id; r[1] = (m[2] * m[7] - m[1] * m[8]) * id; r[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    r[3] = c1 * id; r[4] = (m[0] * m[8] - m[2] * m[6]) * id; r[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    r[6] = c2 * id; r[7] = (m[1] * m[6] - m[0] * m[7]) * id; r[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    return true;
}

// VertexSE3Sophus::oplusImpl (G2oTypes.h:38-45): estimate order [omega; t], Sophus order [t; omega]
__device__ void lm_oplus_pose(double pose[6], const double upd[6])
{
    const double v[6] = { upd[3], upd[4], upd[5], upd[0], upd[1], upd[2] };
    const double est[6] = { pose[3], pose[4], pose[5], pose[0], pose[1], pose[2] };
    Se3 A, Bm, Cm; double r[6];
    se3_exp_d(v, &A); se3_exp_d(est, &Bm);
    se3_mul_d(&A, &Bm, &Cm);
    se3_log_d(&Cm, r);
    pose[0] = r[3]; pose[1] = r[4]; pose[2] = r[5]; pose[3] = r[0]; pose[4] = r[1]; pose[5] = r[2];
}

// computeActiveErrors + buildSystem at the current state; returns the robustified chi2 (block-uniform)
__device__ double lm_linearize(const BaDev &B, double (*red27)[28], double *red)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) *B.n_behind = 0;
    if (tid < B.K) ba_pose_prep_one(B, tid);
    __syncthreads();
    double chi = 0.0;
    for (int il = tid; il < B.P; il += LM_THREADS) chi += ba_point_edges(B, il);
    chi = lm_block_sum(chi, red);                        // (barriers inside: the per-edge records are visible below)
    for (int a = 0; a < B.Kf; ++a) {                     // pose blocks: every point that sees free pose a contributes
        const int k = B.free_pose[a];
        double acc[27];
#pragma unroll
        for (int i = 0; i < 27; ++i) acc[i] = 0.0;
        for (int il = tid; il < B.P; il += LM_THREADS) ba_pose_contrib(B, il, a, acc);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 27; ++i) { const double v = lm_wave_sum(acc[i]); if (lane == 0) red27[wv][i] = v; }
        __syncthreads();
        if (tid < 27) {
            double s = 0.0;
            for (int w = 0; w < LM_WAVES; ++w) s += red27[w][tid];
            if (tid < 21) {
                int u = 0, rem = tid;
                while (rem >= 6 - u) { rem -= 6 - u; ++u; }
                const int v = u + rem;
                B.Hpp[36 * (size_t)k + 6 * u + v] = s; B.Hpp[36 * (size_t)k + 6 * v + u] = s;
            } else B.bp[6 * (size_t)k + (tid - 21)] = s;
        }
    }
    __syncthreads();
    return chi;
}

// computeActiveErrors at the (trial) state
__device__ double lm_errors(const BaDev &B, double *red)
{
    const int tid = threadIdx.x;
    if (tid < B.K) ba_pose_prep_one(B, tid);
    __syncthreads();
    double chi = 0.0;
    for (int il = tid; il < B.P; il += LM_THREADS) chi += ba_point_chi2(B, il);
    return lm_block_sum(chi, red);
}

__global__ __launch_bounds__(LM_THREADS) void k_ba_lm(const BaDev *__restrict__ wins, int max_iterations, ygz_ba_stats *__restrict__ stats)
{
    __shared__ double S[LM_MAXN * LM_MAXN];
    __shared__ double bs[LM_MAXN], xp[LM_MAXN];
    __shared__ double red27[LM_WAVES][28];
    __shared__ double red[LM_WAVES];
    __shared__ int s_fail;
    const BaDev B = wins[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int K = B.K, P = B.P, Kf = B.Kf, n = 6 * Kf;
    double lambda = 0.0, ni = 2.0, currentChi = 0.0, chi_initial = 0.0;
    int iterations = 0, trials = 0;

    for (int it = 0; it < max_iterations; ++it) {
        currentChi = lm_linearize(B, red27, red);
        if (it == 0) {
            chi_initial = currentChi;
            double mx = 0.0;                                        // computeLambdaInit: tau * max |diag| over the active vertices
            for (int i = tid; i < 6 * Kf; i += LM_THREADS) mx = fmax(mx, fabs(B.Hpp[36 * (size_t)B.free_pose[i / 6] + 7 * (i % 6)]));
            for (int i = tid; i < 3 * P; i += LM_THREADS) mx = fmax(mx, fabs(BA_PC(B.Hll_c, i / 3, 9, 4 * (i % 3))));
            lambda = 1e-5 * lm_block_max(mx, red); ni = 2.0;
        }
        double rho = 0.0; int qmax = 0;
        do {
            // ---- _optimizer->push(): backup of the state
            for (int i = tid; i < 6 * K; i += LM_THREADS) B.poses_bk[i] = B.poses_w[i];
            for (int i = tid; i < 3 * P; i += LM_THREADS) B.points_bk[i] = B.points_w[i];
            if (tid == 0) s_fail = 0;
            __syncthreads();
            // ---- 1. Dinv, Y = Hpl Dinv
            for (int il = tid; il < P; il += LM_THREADS) {
                double *Di = B.Dinv + 9 * (size_t)il;
                if (B.point_fixed[il]) { for (int i = 0; i < 9; ++i) Di[i] = 0.0; continue; }
                double D[9];
                for (int i = 0; i < 9; ++i) D[i] = BA_PC(B.Hll_c, il, 9, i);
                D[0] += lambda; D[4] += lambda; D[8] += lambda;
                double Dv[9];
                if (!lm_inv3(D, Dv)) { s_fail = 1; for (int i = 0; i < 9; ++i) Dv[i] = 0.0; }
                for (int i = 0; i < 9; ++i) Di[i] = Dv[i];
                const int ln = il & 63, row0 = B.slot_off[il >> 6], rows = B.slot_off[(il >> 6) + 1] - row0;
                for (int c = 0; c < rows; ++c) {
                    const int row = row0 + c;
                    if (B.pose_c[(size_t)row * 64 + ln] < 0) continue;
                    double W[18];
                    for (int i = 0; i < 18; ++i) W[i] = BA_EC(B.Hpl_c, row, 18, i, ln);
                    for (int r = 0; r < 6; ++r) for (int cc = 0; cc < 3; ++cc)
                        BA_EC(B.Y_c, row, 18, 3 * r + cc, ln) = W[3 * r] * Dv[cc] + W[3 * r + 1] * Dv[3 + cc] + W[3 * r + 2] * Dv[6 + cc];
                }
            }
            // ---- 2. S = blockdiag(Hpp + lambda I), bs = bp
            for (int i = tid; i < n * n; i += LM_THREADS) {
                const int r = i / n, c = i - r * n, a = r / 6, b = c / 6;
                double v = 0.0;
                if (a == b) { v = B.Hpp[36 * (size_t)B.free_pose[a] + 6 * (r - 6 * a) + (c - 6 * b)]; if (r == c) v += lambda; }
                S[i] = v;
            }
            for (int i = tid; i < n; i += LM_THREADS) bs[i] = B.bp[6 * (size_t)B.free_pose[i / 6] + (i % 6)];
            __syncthreads();
            //         S(a,b) -= sum_l Y_a(l) W_b(l)^T, bs_a -= sum_l Y_a(l) b_l: one wavefront per pose pair
            const int npairs = Kf * (Kf + 1) / 2;
            for (int pr = wv; pr < npairs; pr += LM_WAVES) {
                int a = 0, rem = pr;
                while (rem >= Kf - a) { rem -= Kf - a; ++a; }
                const int b = a + rem;
                double acc[36], accb[6];
#pragma unroll
                for (int i = 0; i < 36; ++i) acc[i] = 0.0;
#pragma unroll
                for (int i = 0; i < 6; ++i) accb[i] = 0.0;
                for (int l = lane; l < P; l += 64) {
                    const int ca = B.ppc[(size_t)l * Kf + a], cb = B.ppc[(size_t)l * Kf + b];
                    if (ca < 0 || cb < 0) continue;
                    const int row0 = B.slot_off[l >> 6], ln = l & 63;
                    double Ya[18], Wb[18];
#pragma unroll
                    for (int i = 0; i < 18; ++i) { Ya[i] = BA_EC(B.Y_c, row0 + ca, 18, i, ln); Wb[i] = BA_EC(B.Hpl_c, row0 + cb, 18, i, ln); }
#pragma unroll
                    for (int r = 0; r < 6; ++r) {
#pragma unroll
                        for (int c = 0; c < 6; ++c) acc[6 * r + c] += Ya[3 * r] * Wb[3 * c] + Ya[3 * r + 1] * Wb[3 * c + 1] + Ya[3 * r + 2] * Wb[3 * c + 2];
                    }
                    if (a == b) {
                        const double g0 = BA_PC(B.bl_c, l, 3, 0), g1 = BA_PC(B.bl_c, l, 3, 1), g2 = BA_PC(B.bl_c, l, 3, 2);
#pragma unroll
                        for (int r = 0; r < 6; ++r) accb[r] += Ya[3 * r] * g0 + Ya[3 * r + 1] * g1 + Ya[3 * r + 2] * g2;
                    }
                }
#pragma unroll
                for (int i = 0; i < 36; ++i) acc[i] = lm_wave_sum(acc[i]);
                if (a == b) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) accb[i] = lm_wave_sum(accb[i]);
                }
                if (lane == 0) {
                    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) {
                        S[(6 * a + r) * n + 6 * b + c] -= acc[6 * r + c];
                        if (a != b) S[(6 * b + c) * n + 6 * a + r] -= acc[6 * r + c];
                    }
                    if (a == b) for (int r = 0; r < 6; ++r) bs[6 * a + r] -= accb[r];
                }
            }
            __syncthreads();
            // ---- 3. Cholesky (right-looking) + substitutions
            for (int j = 0; j < n; ++j) {
                if (tid == 0) { const double d = S[j * n + j]; if (!(d > 0) || !isfinite(d)) s_fail = 1; S[j * n + j] = sqrt(d > 0 ? d : 1.0); }
                __syncthreads();
                const double dj = S[j * n + j];
                for (int i = j + 1 + tid; i < n; i += LM_THREADS) S[i * n + j] = S[i * n + j] / dj;
                __syncthreads();
                const int m = n - j - 1;                              // trailing block: rows i > j, columns j < k <= i
                for (int t = tid; t < m * m; t += LM_THREADS) {
                    const int i = j + 1 + t / m, k = j + 1 + t % m;
                    if (k <= i) S[i * n + k] -= S[i * n + j] * S[k * n + j];
                }
                __syncthreads();
            }
            if (wv == 0) {                                            // L y = bs, then L^T x = y (column-oriented, one wavefront)
                for (int k = 0; k < n; ++k) {
                    if (lane == 0) bs[k] = bs[k] / S[k * n + k];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                    const double yk = bs[k];
                    for (int i = k + 1 + lane; i < n; i += 64) bs[i] -= S[i * n + k] * yk;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                }
                for (int k = n - 1; k >= 0; --k) {
                    if (lane == 0) bs[k] = bs[k] / S[k * n + k];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                    const double xk = bs[k];
                    for (int i = lane; i < k; i += 64) bs[i] -= S[k * n + i] * xk;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                }
                for (int i = lane; i < n; i += 64) xp[i] = bs[i];
            }
            __syncthreads();
            const bool ok2 = s_fail == 0;
            double scale = 0.0;
            if (ok2) {
                // ---- 4. x_l, then _optimizer->update(x)
                for (int il = tid; il < P; il += LM_THREADS) {
                    if (B.point_fixed[il]) { B.xl[3 * (size_t)il] = B.xl[3 * (size_t)il + 1] = B.xl[3 * (size_t)il + 2] = 0.0; continue; }
                    double r3[3] = { BA_PC(B.bl_c, il, 3, 0), BA_PC(B.bl_c, il, 3, 1), BA_PC(B.bl_c, il, 3, 2) };
                    const int ln = il & 63, row0 = B.slot_off[il >> 6], rows = B.slot_off[(il >> 6) + 1] - row0;
                    for (int c = 0; c < rows; ++c) {
                        const int row = row0 + c, ip = B.pose_c[(size_t)row * 64 + ln];
                        if (ip < 0) continue;
                        const int a = B.free_idx[ip];
                        if (a < 0) continue;
                        for (int cc = 0; cc < 3; ++cc) for (int r = 0; r < 6; ++r) r3[cc] -= BA_EC(B.Hpl_c, row, 18, 3 * r + cc, ln) * xp[6 * a + r];
                    }
                    const double *Di = B.Dinv + 9 * (size_t)il;
                    double x3[3];
                    for (int cc = 0; cc < 3; ++cc) x3[cc] = Di[3 * cc] * r3[0] + Di[3 * cc + 1] * r3[1] + Di[3 * cc + 2] * r3[2];
                    for (int cc = 0; cc < 3; ++cc) {
                        B.xl[3 * (size_t)il + cc] = x3[cc];
                        scale += x3[cc] * (lambda * x3[cc] + BA_PC(B.bl_c, il, 3, cc));             // computeScale
                        B.points_w[3 * (size_t)il + cc] += x3[cc];
                    }
                }
                if (tid < Kf) {
                    const int k = B.free_pose[tid];
                    double pose[6], upd[6];
                    for (int d = 0; d < 6; ++d) { pose[d] = B.poses_w[6 * (size_t)k + d]; upd[d] = xp[6 * tid + d]; scale += upd[d] * (lambda * upd[d] + B.bp[6 * (size_t)k + d]); }
                    lm_oplus_pose(pose, upd);
                    for (int d = 0; d < 6; ++d) B.poses_w[6 * (size_t)k + d] = pose[d];
                }
            }
            scale = lm_block_sum(scale, red);                          // (barriers inside: the trial state is visible)
            // ---- computeActiveErrors at the trial state
            double tempChi = DBL_MAX;
            if (ok2) tempChi = lm_errors(B, red);
            rho = (currentChi - tempChi) / (scale + 1e-3);
            ++trials;
            if (rho > 0 && isfinite(tempChi)) {                       // good step
                double alpha = 1. - pow((2 * rho - 1), 3);
                alpha = fmin(alpha, 2. / 3.);
                lambda *= fmax(1. / 3., alpha);
                ni = 2; currentChi = tempChi;
            } else {                                                  // bad step: _optimizer->pop()
                lambda *= ni; ni *= 2;
                __syncthreads();
                for (int i = tid; i < 6 * K; i += LM_THREADS) B.poses_w[i] = B.poses_bk[i];
                for (int i = tid; i < 3 * P; i += LM_THREADS) B.points_w[i] = B.points_bk[i];
                __syncthreads();
                if (!isfinite(lambda)) break;
            }
            qmax++;
        } while (rho < 0 && qmax < 10);
        ++iterations;
        if (qmax == 10 || rho == 0 || !isfinite(lambda)) break;        // Terminate
    }
    if (tid == 0) {
        ygz_ba_stats st;
        st.iterations = iterations; st.lm_trials = trials; st.chi2_initial = chi_initial; st.chi2_final = currentChi; st.lambda_final = lambda;
        stats[blockIdx.x] = st;
    }
}

extern "C" {

int ygz_hip_ba_optimize_resident(ygz_hip_ctx *ctx, int window_begin, int n_windows, int max_iterations, ygz_ba_stats *stats)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || window_begin < 0 || n_windows < 1 || window_begin + n_windows > (int)ctx->ba.size() || max_iterations < 0) return YGZ_E_INVALID;
    for (int i = window_begin; i < window_begin + n_windows; ++i) {
        if (!ctx->ba[i]) return YGZ_E_INVALID;
        if (ctx->ba[i]->formulation != 0) return YGZ_E_INVALID;      // the g2o path of the live tree
        if (ctx->ba[i]->Kf > LM_MAXKF || ctx->ba[i]->K > LM_THREADS) return YGZ_E_CAPACITY;   // reduced system must fit LDS
        if (ctx->ba[i]->has_dup) return YGZ_E_INVALID;            // the pair sweep of the Schur step takes ONE edge per (point, pose): use ygz_hip_ba_optimize
    }
    int rc = YGZ_OK;
    const BaDev *table = ygz_ba_table(ctx, &rc);
    if (!table) return rc;
    void *d_stats = nullptr;
    if ((rc = ygz_scratch(ctx, SCR_BA_0, (size_t)n_windows * sizeof(ygz_ba_stats), &d_stats)) != YGZ_OK) return rc;
    YgzAuxScope aux(ctx, 1);
    YGZ_LAUNCH(ctx, KID_BA_LM, k_ba_lm, dim3(n_windows), dim3(LM_THREADS), table + window_begin, max_iterations, (ygz_ba_stats *)d_stats);
    YGZ_HIPCHK(ctx, hipGetLastError());
    if (stats) {
        YGZ_HIPCHK(ctx, hipMemcpyAsync(stats, d_stats, (size_t)n_windows * sizeof(ygz_ba_stats), hipMemcpyDeviceToHost, ctx->stream));
        YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return YGZ_OK;
}

int ygz_hip_ba_get_state(ygz_hip_ctx *ctx, int window, double *poses, double *points)
{
    YgzDeviceGuard dg_(ctx);
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || window < 0 || window >= (int)ctx->ba.size() || !ctx->ba[window]) return YGZ_E_INVALID;
    auto *w = ctx->ba[window];
    if (poses) YGZ_HIPCHK(ctx, hipMemcpyAsync(poses, w->poses, (size_t)w->K * 48, hipMemcpyDeviceToHost, ctx->stream));
    if (points) YGZ_HIPCHK(ctx, hipMemcpyAsync(points, w->points, (size_t)w->P * 24, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return YGZ_OK;
}

}  // extern "C"

// =====================================================================================================================================
// SURVEY 8f-1, second half -- ceres::Solve as ba::LocalBA / OptimizeCurrent / OptimizeCurrentPointOnly / TwoViewBACeres configure it
// (src/Algorithm/BA.cpp:58-62,136-140,308-312,372-375: default options = trust-region Levenberg-Marquardt, Jacobi scaling, Schur
// elimination of the points) resident on the GPU: one workgroup per formulation-2 window runs IterationZero, every
// LevenbergMarquardtStrategy::ComputeStep (scaled blocks, clamped diagonal / radius, Schur complement, Cholesky, back-substitution),
// the step-validity and model-cost tests, the candidate evaluation and the radius policy without a host round trip.  It follows
// ygz_hip_ba_solve_ceres (ba_lm.hip: the same loop with the reduced system on the host) and oracle/ceres_ba.c::yo_ceres_solve
// [frozen spec of ceres-solver 1.13] decision by decision; the parameter update is the additive one of the ceres functors.
__device__ __forceinline__ double ce_point_cost(const BaDev &B, int il, int *behind)
{   // ba_point_chi2 + the PoseOnly functor's failure condition (p_z < 0 on an enabled edge)
    const int lane = il & 63, row0 = B.slot_off[il >> 6], rows = B.slot_off[(il >> 6) + 1] - row0;
    const double pt[3] = { B.points[3 * (size_t)il], B.points[3 * (size_t)il + 1], B.points[3 * (size_t)il + 2] };
    double sum = 0.0;
    for (int c = 0; c < rows; ++c) {
        const int row = row0 + c, ip = B.pose_c[(size_t)row * 64 + lane];
        if (ip < 0 || !B.enable_c[(size_t)row * 64 + lane]) continue;
        double p[3], r[2], rho0, rho1;
        ba_project(B, B.posed + BA_POSED * (size_t)ip, pt, BA_EC(B.obs_c, row, 2, 0, lane), BA_EC(B.obs_c, row, 2, 1, lane), p, r);
        if (p[2] < 0) *behind = 1;
        ba_robust(r[0] * r[0] + r[1] * r[1], B.huber_c[(size_t)row * 64 + lane], &rho0, &rho1);
        sum += rho0;
    }
    return sum;
}

__global__ __launch_bounds__(LM_THREADS) void k_ba_ceres(const BaDev *__restrict__ wins, ygz_ceres_options o, ygz_ceres_summary *__restrict__ sums)
{
    __shared__ double S[LM_MAXN * LM_MAXN];
    __shared__ double bs[LM_MAXN], xp[LM_MAXN];
    __shared__ double red27[LM_WAVES][28];
    __shared__ double red[LM_WAVES];
    __shared__ double dxp[16 * 6];
    __shared__ int s_fail, s_behind;
    const BaDev B = wins[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int K = B.K, P = B.P, Kf = B.Kf, n = 6 * Kf;
    ygz_ceres_summary R;
    R.iterations = 0; R.successful_steps = 0; R.unsuccessful_steps = 0; R.termination = YGZ_CERES_NO_CONVERGENCE;
    R.initial_cost = 0.0; R.final_cost = 0.0; R.final_radius = 0.0;
    double x_cost = 0.0, radius = o.initial_trust_region_radius, decrease_factor = 2.0, x_norm = 0.0, gmax = 0.0;
    int invalid_run = 0, term = YGZ_CERES_NO_CONVERGENCE;

    // x_norm over the free parameters, max |gradient| (block-uniform)
#define CE_NORM_GRADIENT()                                                                                                           \
    {   double s2_ = 0.0, g_ = 0.0;                                                                                                  \
        for (int i = tid; i < 6 * Kf; i += LM_THREADS) { const int k_ = B.free_pose[i / 6], d_ = i % 6;                              \
            const double v_ = B.poses_w[6 * (size_t)k_ + d_]; s2_ += v_ * v_; g_ = fmax(g_, fabs(B.bp[6 * (size_t)k_ + d_])); }     \
        for (int i = tid; i < 3 * P; i += LM_THREADS) { const int l_ = i / 3, d_ = i % 3; if (B.point_fixed[l_]) continue;           \
            const double v_ = B.points_w[i]; s2_ += v_ * v_; g_ = fmax(g_, fabs(BA_PC(B.bl_c, l_, 3, d_))); }                        \
        x_norm = sqrt(lm_block_sum(s2_, red)); gmax = lm_block_max(g_, red); }

    do {
        // ---- IterationZero: residuals, Jacobians (as blocks), cost at the start
        const double chi0 = lm_linearize(B, red27, red);
        const int nb0 = *B.n_behind;                              // ba_point_edges counted the enabled edges with p_z < 0
        x_cost = 0.5 * chi0;
        if ((o.fail_behind_camera && nb0 > 0) || !isfinite(chi0)) { term = YGZ_CERES_FAILURE; break; }
        R.initial_cost = x_cost;
        for (int i = tid; i < 6 * K; i += LM_THREADS) B.sc_p[i] = o.jacobi_scaling ? 1.0 / (1.0 + sqrt(B.Hpp[36 * (size_t)(i / 6) + 7 * (i % 6)])) : 1.0;
        for (int i = tid; i < 3 * P; i += LM_THREADS) B.sc_l[i] = o.jacobi_scaling ? 1.0 / (1.0 + sqrt(BA_PC(B.Hll_c, i / 3, 9, 4 * (i % 3)))) : 1.0;
        __syncthreads();
        CE_NORM_GRADIENT()
        for (;;) {
            if (R.iterations >= o.max_num_iterations) { term = YGZ_CERES_NO_CONVERGENCE; break; }
            if (gmax <= o.gradient_tolerance) { term = YGZ_CERES_GRADIENT_TOLERANCE; break; }
            if (radius <= o.min_trust_region_radius) { term = YGZ_CERES_MIN_RADIUS; break; }
            ++R.iterations;
            __syncthreads();
            if (tid == 0) { s_fail = 0; s_behind = 0; }
            __syncthreads();
            // ---- 1. per point: D = sHll + diag(clamp(diag sHll) / radius), Dinv, Y_e = sHpl_e Dinv (sH = column-scaled blocks)
            for (int il = tid; il < P; il += LM_THREADS) {
                double *Di = B.Dinv + 9 * (size_t)il;
                const int ln = il & 63, row0 = B.slot_off[il >> 6], rows = B.slot_off[(il >> 6) + 1] - row0;
                double Dv[9];
                bool fixedl = B.point_fixed[il] != 0;
                if (!fixedl) {
                    const double sl[3] = { B.sc_l[3 * (size_t)il], B.sc_l[3 * (size_t)il + 1], B.sc_l[3 * (size_t)il + 2] };
                    double D[9];
                    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) D[3 * r + c] = BA_PC(B.Hll_c, il, 9, 3 * r + c) * sl[r] * sl[c];
                    for (int r = 0; r < 3; ++r) { const double dg = fmin(fmax(D[4 * r], o.min_lm_diagonal), o.max_lm_diagonal); D[4 * r] += dg / radius; }
                    if (!lm_inv3(D, Dv)) { s_fail = 1; fixedl = true; }
                }
                if (fixedl) for (int i = 0; i < 9; ++i) Dv[i] = 0.0;
                for (int i = 0; i < 9; ++i) Di[i] = Dv[i];
                for (int c = 0; c < rows; ++c) {
                    const int row = row0 + c, ip = B.pose_c[(size_t)row * 64 + ln];
                    if (ip < 0) continue;
                    double W[18];
                    for (int r = 0; r < 6; ++r) for (int cc = 0; cc < 3; ++cc)
                        W[3 * r + cc] = BA_EC(B.Hpl_c, row, 18, 3 * r + cc, ln) * B.sc_p[6 * (size_t)ip + r] * B.sc_l[3 * (size_t)il + cc];
                    for (int r = 0; r < 6; ++r) for (int cc = 0; cc < 3; ++cc)
                        BA_EC(B.Y_c, row, 18, 3 * r + cc, ln) = W[3 * r] * Dv[cc] + W[3 * r + 1] * Dv[3 + cc] + W[3 * r + 2] * Dv[6 + cc];
                }
            }
            // ---- 2. S = blockdiag(sHpp + dp) - sum_l Y_a sHpl_b^T, bs = sbp - sum_l Y_a sbl
            for (int i = tid; i < n * n; i += LM_THREADS) {
                const int r = i / n, c = i - r * n, a = r / 6, b = c / 6;
                double v = 0.0;
                if (a == b) {
                    const int k = B.free_pose[a], rr = r - 6 * a, cc = c - 6 * b;
                    v = B.Hpp[36 * (size_t)k + 6 * rr + cc] * B.sc_p[6 * (size_t)k + rr] * B.sc_p[6 * (size_t)k + cc];
                    if (r == c) { const double dg = fmin(fmax(v, o.min_lm_diagonal), o.max_lm_diagonal); v += dg / radius; }
                }
                S[i] = v;
            }
            for (int i = tid; i < n; i += LM_THREADS) { const int k = B.free_pose[i / 6]; bs[i] = B.bp[6 * (size_t)k + (i % 6)] * B.sc_p[6 * (size_t)k + (i % 6)]; }
            __syncthreads();
            const int npairs = Kf * (Kf + 1) / 2;
            for (int pr = wv; pr < npairs; pr += LM_WAVES) {
                int a = 0, rem = pr;
                while (rem >= Kf - a) { rem -= Kf - a; ++a; }
                const int b = a + rem, kb = B.free_pose[b];
                double acc[36], accb[6];
#pragma unroll
                for (int i = 0; i < 36; ++i) acc[i] = 0.0;
#pragma unroll
                for (int i = 0; i < 6; ++i) accb[i] = 0.0;
                for (int l = lane; l < P; l += 64) {
                    const int ca = B.ppc[(size_t)l * Kf + a], cb = B.ppc[(size_t)l * Kf + b];
                    if (ca < 0 || cb < 0 || B.point_fixed[l]) continue;
                    const int row0 = B.slot_off[l >> 6], ln = l & 63;
                    const double sl[3] = { B.sc_l[3 * (size_t)l], B.sc_l[3 * (size_t)l + 1], B.sc_l[3 * (size_t)l + 2] };
                    double Ya[18], Wb[18];
#pragma unroll
                    for (int i = 0; i < 18; ++i) { Ya[i] = BA_EC(B.Y_c, row0 + ca, 18, i, ln); Wb[i] = BA_EC(B.Hpl_c, row0 + cb, 18, i, ln) * B.sc_p[6 * (size_t)kb + i / 3] * sl[i % 3]; }
#pragma unroll
                    for (int r = 0; r < 6; ++r) {
#pragma unroll
                        for (int c = 0; c < 6; ++c) acc[6 * r + c] += Ya[3 * r] * Wb[3 * c] + Ya[3 * r + 1] * Wb[3 * c + 1] + Ya[3 * r + 2] * Wb[3 * c + 2];
                    }
                    if (a == b) {
                        const double g0 = BA_PC(B.bl_c, l, 3, 0) * sl[0], g1 = BA_PC(B.bl_c, l, 3, 1) * sl[1], g2 = BA_PC(B.bl_c, l, 3, 2) * sl[2];
#pragma unroll
                        for (int r = 0; r < 6; ++r) accb[r] += Ya[3 * r] * g0 + Ya[3 * r + 1] * g1 + Ya[3 * r + 2] * g2;
                    }
                }
#pragma unroll
                for (int i = 0; i < 36; ++i) acc[i] = lm_wave_sum(acc[i]);
                if (a == b) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) accb[i] = lm_wave_sum(accb[i]);
                }
                if (lane == 0) {
                    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) {
                        S[(6 * a + r) * n + 6 * b + c] -= acc[6 * r + c];
                        if (a != b) S[(6 * b + c) * n + 6 * a + r] -= acc[6 * r + c];
                    }
                    if (a == b) for (int r = 0; r < 6; ++r) bs[6 * a + r] -= accb[r];
                }
            }
            __syncthreads();
            // ---- 3. Cholesky + substitutions (as in k_ba_lm)
            for (int j = 0; j < n; ++j) {
                if (tid == 0) { const double d = S[j * n + j]; if (!(d > 0) || !isfinite(d)) s_fail = 1; S[j * n + j] = sqrt(d > 0 ? d : 1.0); }
                __syncthreads();
                const double dj = S[j * n + j];
                for (int i = j + 1 + tid; i < n; i += LM_THREADS) S[i * n + j] = S[i * n + j] / dj;
                __syncthreads();
                const int m = n - j - 1;
                for (int t = tid; t < m * m; t += LM_THREADS) {
                    const int i = j + 1 + t / m, k = j + 1 + t % m;
                    if (k <= i) S[i * n + k] -= S[i * n + j] * S[k * n + j];
                }
                __syncthreads();
            }
            if (wv == 0) {
                for (int k = 0; k < n; ++k) {
                    if (lane == 0) bs[k] = bs[k] / S[k * n + k];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                    const double yk = bs[k];
                    for (int i = k + 1 + lane; i < n; i += 64) bs[i] -= S[i * n + k] * yk;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                }
                for (int k = n - 1; k >= 0; --k) {
                    if (lane == 0) bs[k] = bs[k] / S[k * n + k];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                    const double xk = bs[k];
                    for (int i = lane; i < k; i += 64) bs[i] -= S[k * n + i] * xk;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                }
                for (int i = lane; i < n; i += 64) xp[i] = bs[i];
            }
            for (int i = tid; i < 6 * K; i += LM_THREADS) dxp[i] = 0.0;
            __syncthreads();
            bool valid = s_fail == 0;
            // ---- 4. x_l = Dinv (sbl - sum sHpl^T xp); steps in the unscaled parameters; finiteness
            int bad = 0;
            if (valid) {
                for (int i = tid; i < n; i += LM_THREADS) { const int k = B.free_pose[i / 6]; const double v = xp[i] * B.sc_p[6 * (size_t)k + (i % 6)]; if (!isfinite(v)) bad = 1; dxp[6 * k + (i % 6)] = v; }
                for (int il = tid; il < P; il += LM_THREADS) {
                    double x3[3] = { 0.0, 0.0, 0.0 };
                    if (!B.point_fixed[il]) {
                        const double sl[3] = { B.sc_l[3 * (size_t)il], B.sc_l[3 * (size_t)il + 1], B.sc_l[3 * (size_t)il + 2] };
                        double r3[3] = { BA_PC(B.bl_c, il, 3, 0) * sl[0], BA_PC(B.bl_c, il, 3, 1) * sl[1], BA_PC(B.bl_c, il, 3, 2) * sl[2] };
                        const int ln = il & 63, row0 = B.slot_off[il >> 6], rows = B.slot_off[(il >> 6) + 1] - row0;
                        for (int c = 0; c < rows; ++c) {
                            const int row = row0 + c, ip = B.pose_c[(size_t)row * 64 + ln];
                            if (ip < 0) continue;
                            const int a = B.free_idx[ip];
                            if (a < 0) continue;
                            for (int cc = 0; cc < 3; ++cc) for (int r = 0; r < 6; ++r)
                                r3[cc] -= BA_EC(B.Hpl_c, row, 18, 3 * r + cc, ln) * B.sc_p[6 * (size_t)ip + r] * sl[cc] * xp[6 * a + r];
                        }
                        const double *Di = B.Dinv + 9 * (size_t)il;
                        for (int cc = 0; cc < 3; ++cc) { x3[cc] = (Di[3 * cc] * r3[0] + Di[3 * cc + 1] * r3[1] + Di[3 * cc + 2] * r3[2]) * sl[cc]; if (!isfinite(x3[cc])) bad = 1; }
                    }
                    for (int cc = 0; cc < 3; ++cc) B.xl[3 * (size_t)il + cc] = x3[cc];
                }
            }
            if (__syncthreads_or(bad)) valid = false;
            // ---- model_cost_change = d.b - d.H d / 2 from the (unscaled) blocks
            double model_cost_change = 0.0;
            if (valid) {
                double mcc = 0.0;
                for (int i = tid; i < n; i += LM_THREADS) {
                    const int k = B.free_pose[i / 6], r = i % 6;
                    double hd = 0.0;
                    for (int c = 0; c < 6; ++c) hd += B.Hpp[36 * (size_t)k + 6 * r + c] * dxp[6 * k + c];
                    mcc += dxp[6 * k + r] * (B.bp[6 * (size_t)k + r] - 0.5 * hd);
                }
                for (int il = tid; il < P; il += LM_THREADS) {
                    if (B.point_fixed[il]) continue;
                    const double d3[3] = { B.xl[3 * (size_t)il], B.xl[3 * (size_t)il + 1], B.xl[3 * (size_t)il + 2] };
                    for (int r = 0; r < 3; ++r) {
                        const double hd = BA_PC(B.Hll_c, il, 9, 3 * r) * d3[0] + BA_PC(B.Hll_c, il, 9, 3 * r + 1) * d3[1] + BA_PC(B.Hll_c, il, 9, 3 * r + 2) * d3[2];
                        mcc += d3[r] * (BA_PC(B.bl_c, il, 3, r) - 0.5 * hd);
                    }
                    const int ln = il & 63, row0 = B.slot_off[il >> 6], rows = B.slot_off[(il >> 6) + 1] - row0;
                    for (int c = 0; c < rows; ++c) {
                        const int row = row0 + c, ip = B.pose_c[(size_t)row * 64 + ln];
                        if (ip < 0) continue;
                        for (int r = 0; r < 6; ++r)
                            mcc -= dxp[6 * ip + r] * (BA_EC(B.Hpl_c, row, 18, 3 * r, ln) * d3[0] + BA_EC(B.Hpl_c, row, 18, 3 * r + 1, ln) * d3[1] + BA_EC(B.Hpl_c, row, 18, 3 * r + 2, ln) * d3[2]);
                    }
                }
                model_cost_change = lm_block_sum(mcc, red);
                if (!(model_cost_change > 0)) valid = false;
            }
            if (!valid) {                                                  // HandleInvalidStep
                if (++invalid_run >= o.max_num_consecutive_invalid_steps) { term = YGZ_CERES_FAILURE; break; }
                radius *= 0.5;
                ++R.unsuccessful_steps;
                continue;
            }
            invalid_run = 0;
            // ---- candidate = x + d (backup first), its cost
            double step2 = 0.0;
            for (int i = tid; i < 6 * K; i += LM_THREADS) { B.poses_bk[i] = B.poses_w[i]; B.poses_w[i] += dxp[i]; step2 += dxp[i] * dxp[i]; }
            for (int i = tid; i < 3 * P; i += LM_THREADS) { const double d = B.xl[i]; B.points_bk[i] = B.points_w[i]; B.points_w[i] += d; step2 += d * d; }
            step2 = lm_block_sum(step2, red);                              // (barriers inside: the candidate is visible)
            if (tid < K) ba_pose_prep_one(B, tid);
            __syncthreads();
            double cc = 0.0; int behind = 0;
            for (int il = tid; il < P; il += LM_THREADS) cc += ce_point_cost(B, il, &behind);
            cc = lm_block_sum(cc, red);
            const int any_behind = __syncthreads_or(behind);
            double cand_cost = 0.5 * cc;
            if ((o.fail_behind_camera && any_behind) || !isfinite(cc)) cand_cost = DBL_MAX;      // a failed evaluation = a step of very high cost
            bool accept = false, leave = false;
            if (sqrt(step2) <= o.parameter_tolerance * (x_norm + o.parameter_tolerance)) { term = YGZ_CERES_PARAMETER_TOLERANCE; leave = true; }
            else {
                const double cost_change = x_cost - cand_cost;
                if (fabs(cost_change) <= o.function_tolerance * x_cost) { term = YGZ_CERES_FUNCTION_TOLERANCE; leave = true; }
                else {
                    const double relative_decrease = cost_change / model_cost_change;
                    if (relative_decrease > o.min_relative_decrease) {     // HandleSuccessfulStep
                        accept = true;
                        double t = 2.0 * relative_decrease - 1.0;
                        t = 1.0 - t * t * t;
                        radius = fmin(radius / fmax(1.0 / 3.0, t), o.max_trust_region_radius);
                        decrease_factor = 2.0;
                        ++R.successful_steps;
                    } else {                                               // StepRejected
                        radius = radius / decrease_factor;
                        decrease_factor *= 2.0;
                        ++R.unsuccessful_steps;
                    }
                }
            }
            if (!accept) {                                                 // the iterate stays (also on the two tolerance exits)
                __syncthreads();
                for (int i = tid; i < 6 * K; i += LM_THREADS) B.poses_w[i] = B.poses_bk[i];
                for (int i = tid; i < 3 * P; i += LM_THREADS) B.points_w[i] = B.points_bk[i];
                __syncthreads();
                if (leave) break;
                continue;
            }
            const double chi = lm_linearize(B, red27, red);                // blocks of the new iterate
            x_cost = 0.5 * chi;
            CE_NORM_GRADIENT()
        }
    } while (0);
#undef CE_NORM_GRADIENT
    if (tid == 0) { R.termination = term; R.final_cost = x_cost; R.final_radius = radius; sums[blockIdx.x] = R; }
}

extern "C" int ygz_hip_ba_solve_ceres_resident(ygz_hip_ctx *ctx, int window_begin, int n_windows, const ygz_ceres_options *opt_in,
                                               ygz_ceres_summary *summaries)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || window_begin < 0 || n_windows < 1 || window_begin + n_windows > (int)ctx->ba.size()) return YGZ_E_INVALID;
    for (int i = window_begin; i < window_begin + n_windows; ++i) {
        if (!ctx->ba[i]) return YGZ_E_INVALID;
        if (ctx->ba[i]->formulation != 2) return YGZ_E_INVALID;      // the ceres functors
        if (ctx->ba[i]->Kf > LM_MAXKF || ctx->ba[i]->K > 16) return YGZ_E_CAPACITY;
        if (ctx->ba[i]->has_dup) return YGZ_E_INVALID;
    }
    ygz_ceres_options opt;
    if (opt_in) opt = *opt_in; else ygz_hip_ceres_default_options(&opt);
    int rc = YGZ_OK;
    const BaDev *table = ygz_ba_table(ctx, &rc);
    if (!table) return rc;
    void *d_sum = nullptr;
    if ((rc = ygz_scratch(ctx, SCR_BA_0, (size_t)n_windows * sizeof(ygz_ceres_summary), &d_sum)) != YGZ_OK) return rc;
    YgzAuxScope aux(ctx, 1);
    YGZ_LAUNCH(ctx, KID_BA_LM, k_ba_ceres, dim3(n_windows), dim3(LM_THREADS), table + window_begin, opt, (ygz_ceres_summary *)d_sum);
    YGZ_HIPCHK(ctx, hipGetLastError());
    if (summaries) {
        YGZ_HIPCHK(ctx, hipMemcpyAsync(summaries, d_sum, (size_t)n_windows * sizeof(ygz_ceres_summary), hipMemcpyDeviceToHost, ctx->stream));
        YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return YGZ_OK;
}
