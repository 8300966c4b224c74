// B7 -- ba::OptimizeCurrentPoseOnly (src/Algorithm/BA.cpp:188-264), the per-frame pose refinement of
// LocalMapping::OptimizeCurrent (src/Module/LocalMapping.cpp:126), for a batch of frames.
//
// The reference builds one ceres problem per frame (one CeresReprojectionErrorPoseOnly residual per feature, 6 parameters),
// solves it four times from the same entry pose while re-classifying inliers against chi2Mono = 5.991 px^2 in between.
// Here a workgroup owns a frame and runs all four rounds on the device: lanes stride over the frame's features, each pass
// (residual + closed-form 2x6 Jacobian -> 21 + 6 + 1 sums, or cost only) ends in a fixed-order block reduction, and lane 0
// runs ceres' trust-region Levenberg-Marquardt bookkeeping on the 6x6 system [frozen spec of ceres-solver 1.13, restated in
// oracle/ceres_ba.c::yo_ceres_solve -- this kernel follows it decision by decision].  No host round trip, no atomics on
// floating point; hundreds of frames per launch.
//
// Reproduced as written: every round restarts from the ENTRY pose (:229); the inlier test of a round uses the _TCW committed
// by the previous round (:233 vs :254); fewer than 10 inliers ends the loop before the commit (:252-253); the functor
// fails behind the camera (CeresReprojectionErrorPoseOnly.h:48-51), which makes the solver reject that step (or give up at
// iteration zero, leaving the pose as it was).
#include "ygz_internal.h"
#include <cstring>
#include "se3_dev.h"

#define PO_THREADS 256
#define PO_NV 28

struct PoArgs {
    YgzPoDev d;               // off (or cnt + stride), use, px, pw, poses, bad, depth, inliers, rounds
    double fx, fy, cx, cy;
    ygz_ceres_options opt;
};

// R(aa) and the left Jacobian J_l(aa) as ceres::AngleAxisRotatePoint defines the rotation (first-order branch at theta^2 <= eps)
__device__ void po_rot_prep(const double *pose, double R[9], double Jl[9])
{
    const double ax = pose[3], ay = pose[4], az = pose[5], theta2 = ax * ax + ay * ay + az * az;
    if (theta2 > 2.220446049250313e-16) {
        const double theta = sqrt(theta2), c = cos(theta), s = sin(theta), ti = 1.0 / theta;
        const double w[3] = { ax * ti, ay * ti, az * ti }, c1 = 1.0 - c, sa = s * ti, cb = c1 * ti;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            R[3 * i + j] = c1 * w[i] * w[j] + (i == j ? c : 0.0);
            Jl[3 * i + j] = (1.0 - sa) * w[i] * w[j] + (i == j ? sa : 0.0);
        }
        R[1] -= s * w[2]; R[2] += s * w[1]; R[3] += s * w[2]; R[5] -= s * w[0]; R[6] -= s * w[1]; R[7] += s * w[0];
        Jl[1] -= cb * w[2]; Jl[2] += cb * w[1]; Jl[3] += cb * w[2]; Jl[5] -= cb * w[0]; Jl[6] -= cb * w[1]; Jl[7] += cb * w[0];
    } else {
        R[0] = 1; R[1] = -az; R[2] = ay; R[3] = az; R[4] = 1; R[5] = -ax; R[6] = -ay; R[7] = ax; R[8] = 1;
        for (int i = 0; i < 9; ++i) Jl[i] = (i % 4 == 0) ? 1.0 : 0.0;
    }
}

// The features of a frame a lane visits are the same in every pass (i = tid, tid + 256, ...), and a single frame's solve makes ~45 passes whose
// chain is load latency, not arithmetic: the first PO_CACHE of them live in registers for the whole kernel (pixel, map point, enabled bit) --
// 4 x 256 = 1024 features (a fifth would cost the second wavefront per SIMD: 255 + 4 registers) -- and only the ones beyond are fetched per pass.
#define PO_CACHE 4
struct PoLane {
    double px[PO_CACHE][2], pw[PO_CACHE][3];
    uint32_t off;                               // bit k: cached feature k is switched off (SetEnable(false)) or absent
};

// one feature's share of a pass (residual; with `full` its Jacobian blocks); returns false when it lies behind the camera
__device__ __forceinline__ bool po_feature(const PoArgs &A, const double R[9], const double Jl[9], double t0, double t1, double t2, double ifx, double ify,
                                           double pxu, double pxv, double X, double Y, double Z, bool full, double acc[PO_NV])
{
    const double a = R[0] * X + R[1] * Y + R[2] * Z, b = R[3] * X + R[4] * Y + R[5] * Z, c = R[6] * X + R[7] * Y + R[8] * Z;
    const double x = a + t0, y = b + t1, z = c + t2;
    if (z < 0) return false;
    // one division per feature and evaluation (1 / z); the quotients by fx, fy and z are products with the reciprocals -- one more rounding than the
    // reference's divisions (1e-16 relative, the bar of the path is 1e-5), a fifth of the FP64 instructions of the loop
    const double obx = (pxu - A.cx) * ifx, oby = (pxv - A.cy) * ify;      // Pixel2Camera2D, Camera.h:64-69
    const double zi = 1. / z;
    const double r0 = obx - x * zi, r1 = oby - y * zi;
    acc[27] += 0.5 * (r0 * r0 + r1 * r1);
    if (!full) return true;
    const double xz = x * zi * zi, yz = y * zi * zi;
    double M[9], Jx[12];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        M[j] = -c * Jl[3 + j] + b * Jl[6 + j];
        M[3 + j] = c * Jl[j] - a * Jl[6 + j];
        M[6 + j] = -b * Jl[j] + a * Jl[3 + j];
    }
    Jx[0] = -zi; Jx[1] = 0.0; Jx[2] = xz; Jx[6] = 0.0; Jx[7] = -zi; Jx[8] = yz;
#pragma unroll
    for (int j = 0; j < 3; ++j) { Jx[3 + j] = zi * M[j] - xz * M[6 + j]; Jx[9 + j] = zi * M[3 + j] - yz * M[6 + j]; }
    int q = 0;
#pragma unroll
    for (int u = 0; u < 6; ++u) {
#pragma unroll
        for (int v = u; v < 6; ++v) acc[q++] += Jx[u] * Jx[v] + Jx[6 + u] * Jx[6 + v];
    }
#pragma unroll
    for (int u = 0; u < 6; ++u) acc[21 + u] += -(Jx[u] * r0 + Jx[6 + u] * r1);
    return true;
}

// One pass over the frame's enabled features at `pose`: sum[0..20] = upper triangle of J^T J, sum[21..26] = -J^T r,
// sum[27] = cost = 1/2 sum |r|^2 (full == false: cost only).  Returns (block-uniform) whether any enabled point was behind
// the camera.  Fixed summation order: lane-strided partials, xor-shuffle tree, the four wave partials in wave order.
__device__ __forceinline__ bool po_eval(const PoArgs &A, const PoLane &C, int beg, int n, const double *pose /*LDS*/, bool full, double (*red)[PO_NV], double *sum)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double R[9], Jl[9];
    po_rot_prep(pose, R, Jl);
    const double t0 = pose[0], t1 = pose[1], t2 = pose[2];
    const double ifx = 1.0 / A.fx, ify = 1.0 / A.fy;
    double acc[PO_NV];
#pragma unroll
    for (int i = 0; i < PO_NV; ++i) acc[i] = 0.0;
    int behind = 0;
#pragma unroll
    for (int k = 0; k < PO_CACHE; ++k) {
        if ((C.off >> k) & 1u) continue;
        if (!po_feature(A, R, Jl, t0, t1, t2, ifx, ify, C.px[k][0], C.px[k][1], C.pw[k][0], C.pw[k][1], C.pw[k][2], full, acc)) behind = 1;
    }
    for (int i = tid + PO_CACHE * PO_THREADS; i < n; i += PO_THREADS) {
        const size_t g = (size_t)(beg + i);
        if (A.d.bad[g]) continue;                                   // SetEnable(false)
        if (!po_feature(A, R, Jl, t0, t1, t2, ifx, ify, A.d.px[2 * g], A.d.px[2 * g + 1], A.d.pw[3 * g], A.d.pw[3 * g + 1], A.d.pw[3 * g + 2], full, acc)) behind = 1;
    }
    const int first = full ? 0 : 27;
    if (full) {                                                 // all 28 sums stage by stage (ygz_wave_sums_d: same additions, 28 independent chains per stage)
        ygz_wave_sums_d(acc);
        if (lane == 63) {
#pragma unroll
            for (int i = 0; i < PO_NV; ++i) red[wv][i] = acc[i];
        }
    } else {
        const double v = ygz_wave_sum_d(acc[27]);
        if (lane == 0) red[wv][27] = v;
    }
    const int any_behind = __syncthreads_or(behind);
    if (tid >= first && tid < PO_NV) sum[tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
    __syncthreads();
    return any_behind != 0;
}

// shared solver state
struct PoState {
    double pose[6], cand[6], backup[6], tcw[6], scale[6];
    double x_cost, radius, decrease_factor, x_norm, gmax, model_cost_change, step_norm;
    int iterations, invalid_run, term, cont, need_eval, accept, cnt;
};
enum { PO_RUNNING = -1 };

// LevenbergMarquardtStrategy::ComputeStep + the validity tests of TrustRegionMinimizer::ComputeTrustRegionStep (lane 0)
__device__ bool po_compute_step(const ygz_ceres_options &o, PoState &S, const double *sum)
{
    double H[36], b[6], Am[36], y[6];
    int q = 0;
    for (int u = 0; u < 6; ++u) for (int v = u; v < 6; ++v) { H[6 * u + v] = sum[q]; H[6 * v + u] = sum[q]; ++q; }
    for (int u = 0; u < 6; ++u) b[u] = sum[21 + u];
    for (int r = 0; r < 6; ++r) {
        for (int c = 0; c < 6; ++c) Am[6 * r + c] = H[6 * r + c] * S.scale[r] * S.scale[c];
        double dg = Am[7 * r];
        dg = dg < o.min_lm_diagonal ? o.min_lm_diagonal : (dg > o.max_lm_diagonal ? o.max_lm_diagonal : dg);
        Am[7 * r] += dg / S.radius;
        y[r] = b[r] * S.scale[r];
    }
    for (int j = 0; j < 6; ++j) {                              // dense Cholesky, as the oracle's chol_solve
        double d = Am[7 * j];
        for (int k = 0; k < j; ++k) d -= Am[6 * j + k] * Am[6 * j + k];
        if (!(d > 0) || !isfinite(d)) return false;
        d = sqrt(d);
        Am[7 * j] = d;
        for (int i = j + 1; i < 6; ++i) {
            double s = Am[6 * i + j];
            for (int k = 0; k < j; ++k) s -= Am[6 * i + k] * Am[6 * j + k];
            Am[6 * i + j] = s / d;
        }
    }
    for (int i = 0; i < 6; ++i) { double s = y[i]; for (int k = 0; k < i; ++k) s -= Am[6 * i + k] * y[k]; y[i] = s / Am[7 * i]; }
    for (int i = 5; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < 6; ++k) s -= Am[6 * k + i] * y[k]; y[i] = s / Am[7 * i]; }
    double mcc = 0, step2 = 0, d[6];
    for (int i = 0; i < 6; ++i) { d[i] = y[i] * S.scale[i]; if (!isfinite(d[i])) return false; }
    for (int r = 0; r < 6; ++r) {
        double hd = 0;
        for (int c = 0; c < 6; ++c) hd += H[6 * r + c] * d[c];
        mcc += d[r] * (b[r] - 0.5 * hd);
    }
    if (!(mcc > 0)) return false;
    for (int i = 0; i < 6; ++i) { S.cand[i] = S.pose[i] + d[i]; step2 += d[i] * d[i]; }
    S.model_cost_change = mcc; S.step_norm = sqrt(step2);
    return true;
}

__device__ void po_norm_gradient(PoState &S, const double *sum)
{
    double s2 = 0, g = 0;
    for (int i = 0; i < 6; ++i) { s2 += S.pose[i] * S.pose[i]; g = fmax(g, fabs(sum[21 + i])); }
    S.x_norm = sqrt(s2); S.gmax = g;
}

__global__ __launch_bounds__(PO_THREADS) void k_pose_only_ba(PoArgs A)
{
    __shared__ double red[4][PO_NV];
    __shared__ double lin[PO_NV], tmp[PO_NV];       // the iterate's linearisation; the candidate's cost
    __shared__ PoState S;
    const int f = blockIdx.x, tid = threadIdx.x;
    const int beg = A.d.cnt ? f * A.d.stride : A.d.off[f], n = A.d.cnt ? A.d.cnt[f] : A.d.off[f + 1] - beg;
    const ygz_ceres_options &o = A.opt;
    if (tid < 6) { const double v = A.d.poses[6 * (size_t)f + tid]; S.backup[tid] = v; S.tcw[tid] = v; }
    for (int i = tid; i < n; i += PO_THREADS) A.d.bad[beg + i] = (uint8_t)(A.d.use ? !A.d.use[beg + i] : 0);   // features that do not exist in this frame
    PoLane C;
    C.off = 0u;
    uint32_t c_absent = 0u;                                      // cached slots past the end of the frame or not features of it (`use`): never enabled
#pragma unroll
    for (int k = 0; k < PO_CACHE; ++k) {
        const int i = tid + k * PO_THREADS;
        const bool have = i < n && !(A.d.use && !A.d.use[beg + (i < n ? i : 0)]);
        const size_t g = (size_t)(beg + (i < n ? i : 0));
        C.px[k][0] = A.d.px[2 * g]; C.px[k][1] = A.d.px[2 * g + 1];
        C.pw[k][0] = A.d.pw[3 * g]; C.pw[k][1] = A.d.pw[3 * g + 1]; C.pw[k][2] = A.d.pw[3 * g + 2];
        if (!have) c_absent |= 1u << k;
    }
    C.off = c_absent;
    __syncthreads();

    int it = 0, cntInlier = 0;
    for (it = 0; it < 4; ++it) {
        // ---------------- ceres::Solve from the entry pose
        if (tid < 6) S.pose[tid] = S.backup[tid];
        __syncthreads();
        bool behind = po_eval(A, C, beg, n, S.pose, true, red, lin);
        if (tid == 0) {                                          // IterationZero
            S.term = PO_RUNNING; S.iterations = 0; S.invalid_run = 0;
            S.radius = o.initial_trust_region_radius; S.decrease_factor = 2.0;
            if (behind) S.term = YGZ_CERES_FAILURE;
            else {
                S.x_cost = lin[27];
                int q = 0;
                for (int u = 0; u < 6; ++u) { S.scale[u] = o.jacobi_scaling ? 1.0 / (1.0 + sqrt(lin[q])) : 1.0; q += 6 - u; }
                po_norm_gradient(S, lin);
            }
        }
        for (;;) {
            __syncthreads();                                     // every lane is done reading S from the previous pass
            if (tid == 0 && S.term == PO_RUNNING) {
                S.need_eval = 0;
                if (S.iterations >= o.max_num_iterations) S.term = YGZ_CERES_NO_CONVERGENCE;
                else if (S.gmax <= o.gradient_tolerance) S.term = YGZ_CERES_GRADIENT_TOLERANCE;
                else if (S.radius <= o.min_trust_region_radius) S.term = YGZ_CERES_MIN_RADIUS;
                else {
                    ++S.iterations;
                    if (po_compute_step(o, S, lin)) { S.invalid_run = 0; S.need_eval = 1; }
                    else if (++S.invalid_run >= o.max_num_consecutive_invalid_steps) S.term = YGZ_CERES_FAILURE;
                    else S.radius *= 0.5;                       // StepIsInvalid
                }
            }
            __syncthreads();
            if (S.term != PO_RUNNING) break;                     // block-uniform: read between two barriers
            if (!S.need_eval) continue;
            behind = po_eval(A, C, beg, n, S.cand, false, red, tmp);
            if (tid == 0) {
                const double cand_cost = behind ? 1.7976931348623157e308 : tmp[27];   // failed evaluation = a step of very high cost
                S.accept = 0;
                if (S.step_norm <= o.parameter_tolerance * (S.x_norm + o.parameter_tolerance)) S.term = YGZ_CERES_PARAMETER_TOLERANCE;
                else {
                    const double cost_change = S.x_cost - cand_cost;
                    if (fabs(cost_change) <= o.function_tolerance * S.x_cost) S.term = YGZ_CERES_FUNCTION_TOLERANCE;
                    else {
                        const double rd = cost_change / S.model_cost_change;
                        if (rd > o.min_relative_decrease) {      // StepAccepted
                            S.accept = 1;
                            for (int i = 0; i < 6; ++i) S.pose[i] = S.cand[i];
                            double t = 2.0 * rd - 1.0;
                            t = 1.0 - t * t * t;
                            S.radius = fmin(S.radius / fmax(1.0 / 3.0, t), o.max_trust_region_radius);
                            S.decrease_factor = 2.0;
                        } else {                                 // StepRejected
                            S.radius = S.radius / S.decrease_factor;
                            S.decrease_factor *= 2.0;
                        }
                    }
                }
            }
            __syncthreads();
            if (S.term != PO_RUNNING) break;
            if (S.accept) {                                      // EvaluateGradientAndJacobian at the new iterate
                po_eval(A, C, beg, n, S.pose, true, red, lin);
                if (tid == 0) { S.x_cost = lin[27]; po_norm_gradient(S, lin); }
            }
        }
        // ---------------- re-classify with current->_TCW (the pose committed by the previous round)
        if (tid == 0) S.cnt = 0;
        __syncthreads();
        {
            double q[4], th;
            so3_exp_d(S.tcw + 3, q, &th);                        // SE3(SO3::exp(aa), t), BA.cpp:254
            int mine = 0;
            auto classify = [&](size_t g, double X, double Y, double Z, double pxu, double pxv) {
                const double pw[3] = { X, Y, Z };
                double pc[3];
                quat_rotate_d(q, pw, pc);
                pc[0] += S.tcw[0]; pc[1] += S.tcw[1]; pc[2] += S.tcw[2];
                // Camera2Pixel as written (Camera.h:48-53: fx * x / z + cx), with the IEEE quotients: this value decides inlier / outlier against a
                // threshold, and a product with 1 / z could flip a feature that sits within an ulp of it.  (The solver passes above
                // keep the reciprocal: their sums are compared at 1e-7, not thresholded.)
                const double u = A.fx * pc[0] / pc[2] + A.cx, v = A.fy * pc[1] / pc[2] + A.cy;
                const double dx = u - pxu, dy = v - pxv, error2 = dx * dx + dy * dy;
                if (error2 > (double)5.991f) { A.d.bad[g] = 1; return false; }                // const float chi2Mono = 5.991, BA.cpp:195
                A.d.depth[g] = pc[2]; A.d.bad[g] = 0;
                return true;
            };
            uint32_t off = c_absent;
#pragma unroll
            for (int k = 0; k < PO_CACHE; ++k) {
                if ((c_absent >> k) & 1u) continue;
                if (classify((size_t)(beg + tid + k * PO_THREADS), C.pw[k][0], C.pw[k][1], C.pw[k][2], C.px[k][0], C.px[k][1])) ++mine; else off |= 1u << k;
            }
            C.off = off;
            for (int i = tid + PO_CACHE * PO_THREADS; i < n; i += PO_THREADS) {
                const size_t g = (size_t)(beg + i);
                if (A.d.use && !A.d.use[g]) continue;
                if (classify(g, A.d.pw[3 * g], A.d.pw[3 * g + 1], A.d.pw[3 * g + 2], A.d.px[2 * g], A.d.px[2 * g + 1])) ++mine;
            }
            if (mine) atomicAdd(&S.cnt, mine);
        }
        __syncthreads();
        cntInlier = S.cnt;
        if (cntInlier < 10) { ++it; break; }
        if (tid < 6) S.tcw[tid] = S.pose[tid];
        __syncthreads();
    }
    if (tid < 6) A.d.poses[6 * (size_t)f + tid] = S.tcw[tid];
    if (tid == 0 && A.d.T_out) {
        double q[4], th;
        so3_exp_d(S.tcw + 3, q, &th);
        for (int k = 0; k < 4; ++k) A.d.T_out[7 * (size_t)f + k] = q[k];
        for (int k = 0; k < 3; ++k) A.d.T_out[7 * (size_t)f + 4 + k] = S.tcw[k];
    }
    if (tid == 0) { if (A.d.inliers) A.d.inliers[f] = cntInlier; if (A.d.rounds) A.d.rounds[f] = it; }
}

// device-array form (resident tracking, track.hip): frame f owns rows [f * stride, f * stride + cnt[f]), `use` masks the rows
// that are features of the frame; everything stays in HBM, asynchronous on the context's stream
int ygz_launch_pose_only(ygz_hip_ctx *ctx, int n_frames, const YgzPoDev &d)
{
    PoArgs A;
    A.d = d;
    A.fx = (double)ctx->prm.fx; A.fy = (double)ctx->prm.fy; A.cx = (double)ctx->prm.cx; A.cy = (double)ctx->prm.cy;
    ygz_hip_ceres_default_options(&A.opt);
    A.opt.fail_behind_camera = 1;
    YGZ_LAUNCH(ctx, KID_POSE_ONLY, k_pose_only_ba, dim3(n_frames), dim3(PO_THREADS), A);
    YGZ_HIPCHK(ctx, hipGetLastError());
    return YGZ_OK;
}

extern "C" int ygz_hip_optimize_pose_only(ygz_hip_ctx *ctx, int n_frames, const int32_t *frame_off, const double *px,
                                          const double *pw, double *poses_io, uint8_t *bad, double *depth, int32_t *inliers,
                                          int32_t *rounds)
{
    YgzDeviceGuard dg_(ctx);
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || n_frames < 0 || !frame_off || !poses_io) return YGZ_E_INVALID;
    if (n_frames == 0) return YGZ_OK;
    if (frame_off[0] != 0) return YGZ_E_INVALID;
    for (int f = 0; f < n_frames; ++f) if (frame_off[f + 1] < frame_off[f]) return YGZ_E_INVALID;
    const size_t n = (size_t)frame_off[n_frames], nz = n > 0 ? n : 1;
    if (n > 0 && (!px || !pw || !bad || !depth)) return YGZ_E_INVALID;
    void *blob = nullptr;
    const size_t b_off = ((size_t)n_frames + 1) * 4, b_px = nz * 16, b_pw = nz * 24, b_pose = (size_t)n_frames * 48, b_depth = nz * 8,
                 b_cnt = (size_t)n_frames * 8, b_bad = nz;
    int rc = ygz_scratch(ctx, SCR_GEN_0, ((b_off + 7) & ~(size_t)7) + b_px + b_pw + b_pose + b_depth + b_cnt + b_bad + 64, &blob);
    if (rc != YGZ_OK) return rc;
    uint8_t *base = (uint8_t *)blob;
    PoArgs A;
    A.d.cnt = nullptr; A.d.stride = 0; A.d.use = nullptr; A.d.T_out = nullptr;
    A.d.off = (const int32_t *)base; base += (b_off + 7) & ~(size_t)7;
    double *d_px = (double *)base; base += b_px;
    double *d_pw = (double *)base; base += b_pw;
    A.d.poses = (double *)base; base += b_pose;
    A.d.depth = (double *)base; base += b_depth;
    A.d.inliers = (int32_t *)base; A.d.rounds = A.d.inliers + n_frames; base += b_cnt;
    A.d.bad = base;
    A.d.px = d_px; A.d.pw = d_pw;
    A.fx = (double)ctx->prm.fx; A.fy = (double)ctx->prm.fy; A.cx = (double)ctx->prm.cx; A.cy = (double)ctx->prm.cy;
    ygz_hip_ceres_default_options(&A.opt);
    A.opt.fail_behind_camera = 1;
    // one copy up, one down: the arrays are packed into the page-locked mirror of the blob at the device offsets
    uint8_t *hb = nullptr;
    if ((rc = ygz_scratch_mirror(ctx, SCR_GEN_0, (void **)&hb)) != YGZ_OK) return rc;
    const uint8_t *b0 = (const uint8_t *)blob;
    const size_t total = (size_t)((const uint8_t *)A.d.bad + b_bad - b0);
#define H_(dptr) (hb + ((const uint8_t *)(dptr) - b0))
    memcpy(H_(A.d.off), frame_off, b_off);
    if (n > 0) { memcpy(H_(d_px), px, n * 16); memcpy(H_(d_pw), pw, n * 24); memcpy(H_(A.d.depth), depth, n * 8); }     // outliers keep their depth
    memcpy(H_(A.d.poses), poses_io, b_pose);
    if ((rc = ygz_kcopy(ctx, blob, hb, total, hipMemcpyHostToDevice)) != YGZ_OK) return rc;
    YGZ_LAUNCH(ctx, KID_POSE_ONLY, k_pose_only_ba, dim3(n_frames), dim3(PO_THREADS), A);
    YGZ_HIPCHK(ctx, hipGetLastError());
    const size_t out0 = (size_t)((const uint8_t *)A.d.poses - b0);                 // poses | depth | counts | bad are the tail of the blob
    if ((rc = ygz_kcopy(ctx, hb + out0, (uint8_t *)blob + out0, total - out0, hipMemcpyDeviceToHost)) != YGZ_OK) return rc;
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(poses_io, H_(A.d.poses), b_pose);
    if (n > 0) { memcpy(bad, H_(A.d.bad), n); memcpy(depth, H_(A.d.depth), n * 8); }
    if (inliers) memcpy(inliers, H_(A.d.inliers), (size_t)n_frames * 4);
    if (rounds) memcpy(rounds, H_(A.d.rounds), (size_t)n_frames * 4);
#undef H_
    return YGZ_OK;
}
