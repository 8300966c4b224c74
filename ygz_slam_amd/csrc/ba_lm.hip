// B4 (+ SURVEY 8f-1) -- the Levenberg-Marquardt loop around the GPU linearisation: what g2o does when
// ba::LocalBAG2O calls optimizer.optimize(20) (src/Algorithm/BA.cpp:390-395,501-502) with
// OptimizationAlgorithmLevenberg + BlockSolver_6_3 (map points marginalised, Schur complement) + a Cholesky solve of
// the reduced pose system [frozen spec of g2o: lambda_init = 1e-5 * max diag(H), rho = (chi - chi_new) /
// (sum x_i (lambda x_i + b_i) + 1e-3), good step: lambda *= max(1/3, min(1 - (2 rho - 1)^3, 2/3)), bad step:
// lambda *= nu, nu *= 2, at most 10 trials per iteration].
// Every residual / Jacobian / block evaluation (one per outer iteration + one per LM trial) runs on the GPU
// through the resident BA window (ba.hip); only the 6K x 6K reduced system (K <= a few dozen keyframes) and the
// per-point 3x3 back-substitution are solved on the host, in FP64.
#include "ygz_internal.h"
#include <chrono>
#include "se3_dev.h"
#include <vector>
#include <cmath>
#include <cfloat>
#include <string.h>
#include <stdlib.h>

static void oplus_pose(double pose[6], const double upd[6])
{   // VertexSE3Sophus::oplusImpl (G2oTypes.h:38-45): estimate order [omega; t], Sophus order [t; omega]
    const double v[6] = { upd[3], upd[4], upd[5], upd[0], upd[1], upd[2] };
    const double est[6] = { pose[3], pose[4], pose[5], pose[0], pose[1], pose[2] };
    Se3 A, B, C; double r[6];
    se3_exp_d(v, &A); se3_exp_d(est, &B);
    se3_mul_d(&A, &B, &C);
    se3_log_d(&C, r);
    pose[0] = r[3]; pose[1] = r[4]; pose[2] = r[5]; pose[3] = r[0]; pose[4] = r[1]; pose[5] = r[2];
}

static bool inv3(const double *m, double *r)
{
    const double c0 = m[4] * m[8] - m[5] * m[7], c1 = m[5] * m[6] - m[3] * m[8], c2 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c0 + m[1] * c1 + m[2] * c2;
    if (!(fabs(det) > 0) || !std::isfinite(det)) return false;
    const double id = 1.0 / det;
    r[0] = c0 * id; r[1] = (m[2] * m[7] - m[1] * m[8]) * id; r[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    r[3] = c1 * id; r[4] = (m[0] * m[8] - m[2] * m[6]) * id; r[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    r[6] = c2 * id; r[7] = (m[1] * m[6] - m[0] * m[7]) * id; r[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    return true;
}

// dense Cholesky solve of an n x n SPD system (row-major, overwritten); false if not positive definite
static bool chol_solve(std::vector<double> &A, std::vector<double> &b, int n)
{
    for (int j = 0; j < n; ++j) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
        if (!(d > 0) || !std::isfinite(d)) return false;
        d = sqrt(d);
        A[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
            A[(size_t)i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= A[(size_t)i * n + k] * b[k]; b[i] = s / A[(size_t)i * n + i]; }
    for (int i = n - 1; i >= 0; --i) { double s = b[i]; for (int k = i + 1; k < n; ++k) s -= A[(size_t)k * n + i] * b[k]; b[i] = s / A[(size_t)i * n + i]; }
    return true;
}

// Solve [Hpp + diag(dp), Hpl; Hpl^T, Hll + diag(dl)] [xp; xl] = [bp; bl], points eliminated first (Schur complement on the
// pose block, dense Cholesky of the 6Kf x 6Kf reduced system, 3x3 back-substitution per point).  free_idx[k] < 0: constant
// pose; lfree[l] == 0: constant point.  xp is [Kf][6] (free poses only), xl [P][3].
struct BlockSystem {
    int K, P, E, Kf;
    const int32_t *edge_pose, *edge_point;
    std::vector<int> free_idx;
    std::vector<uint8_t> lfree;
    std::vector<std::vector<int>> pt_edges;
    std::vector<double> S, bs, Dinv;
};

static void block_system_init(BlockSystem &B, const ygz_ba_problem *pb)
{
    B.K = pb->n_poses; B.P = pb->n_points; B.E = pb->n_edges; B.edge_pose = pb->edge_pose; B.edge_point = pb->edge_point;
    B.free_idx.assign(B.K, -1); B.Kf = 0;
    for (int k = 0; k < B.K; ++k) if (!(pb->pose_fixed && pb->pose_fixed[k])) B.free_idx[k] = B.Kf++;
    B.lfree.assign(B.P, 1);
    if (pb->point_fixed) for (int l = 0; l < B.P; ++l) B.lfree[l] = pb->point_fixed[l] ? 0 : 1;
    B.pt_edges.assign(B.P, std::vector<int>());
    for (int e = 0; e < B.E; ++e) B.pt_edges[pb->edge_point[e]].push_back(e);
    B.Dinv.assign((size_t)B.P * 9, 0.0);
}

static bool block_solve(BlockSystem &B, const double *Hpp, const double *Hll, const double *Hpl, const double *bp, const double *bl,
                        const double *dp /*[K][6]*/, const double *dl /*[P][3]*/, std::vector<double> &xp, std::vector<double> &xl)
{
    const int n = 6 * B.Kf;
    B.S.assign((size_t)n * n, 0.0); B.bs.assign((size_t)n, 0.0);
    xl.assign((size_t)B.P * 3, 0.0);
    for (int k = 0; k < B.K; ++k) if (B.free_idx[k] >= 0) {
        const int a = B.free_idx[k];
        for (int r = 0; r < 6; ++r) {
            for (int c = 0; c < 6; ++c) B.S[(size_t)(6 * a + r) * n + 6 * a + c] = Hpp[(size_t)k * 36 + 6 * r + c];
            B.S[(size_t)(6 * a + r) * n + 6 * a + r] += dp[(size_t)k * 6 + r];
            B.bs[6 * a + r] = bp[(size_t)k * 6 + r];
        }
    }
    for (int l = 0; l < B.P; ++l) {
        if (!B.lfree[l]) continue;
        double D[9]; memcpy(D, &Hll[(size_t)l * 9], sizeof(D));
        D[0] += dl[(size_t)l * 3]; D[4] += dl[(size_t)l * 3 + 1]; D[8] += dl[(size_t)l * 3 + 2];
        double *Di = &B.Dinv[(size_t)l * 9];
        if (!inv3(D, Di)) return false;
        for (int ei : B.pt_edges[l]) {
            const int a = B.free_idx[B.edge_pose[ei]];
            if (a < 0) continue;
            const double *Bi = &Hpl[(size_t)ei * 18];           // 6x3
            double BD[18];
            for (int r = 0; r < 6; ++r) for (int c = 0; c < 3; ++c) BD[3 * r + c] = Bi[3 * r] * Di[c] + Bi[3 * r + 1] * Di[3 + c] + Bi[3 * r + 2] * Di[6 + c];
            for (int r = 0; r < 6; ++r) B.bs[6 * a + r] -= BD[3 * r] * bl[(size_t)l * 3] + BD[3 * r + 1] * bl[(size_t)l * 3 + 1] + BD[3 * r + 2] * bl[(size_t)l * 3 + 2];
            for (int ej : B.pt_edges[l]) {
                const int b2 = B.free_idx[B.edge_pose[ej]];
                if (b2 < 0) continue;
                const double *Bj = &Hpl[(size_t)ej * 18];
                for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c)
                    B.S[(size_t)(6 * a + r) * n + 6 * b2 + c] -= BD[3 * r] * Bj[3 * c] + BD[3 * r + 1] * Bj[3 * c + 1] + BD[3 * r + 2] * Bj[3 * c + 2];
            }
        }
    }
    xp = B.bs;
    if (n > 0 && !chol_solve(B.S, xp, n)) return false;
    for (int l = 0; l < B.P; ++l) {
        if (!B.lfree[l]) continue;
        double r3[3] = { bl[(size_t)l * 3], bl[(size_t)l * 3 + 1], bl[(size_t)l * 3 + 2] };
        for (int ei : B.pt_edges[l]) {
            const int a = B.free_idx[B.edge_pose[ei]];
            if (a < 0) continue;
            const double *Bi = &Hpl[(size_t)ei * 18];
            for (int c = 0; c < 3; ++c) for (int r = 0; r < 6; ++r) r3[c] -= Bi[3 * r + c] * xp[6 * a + r];
        }
        const double *Di = &B.Dinv[(size_t)l * 9];
        for (int c = 0; c < 3; ++c) xl[(size_t)l * 3 + c] = Di[3 * c] * r3[0] + Di[3 * c + 1] * r3[1] + Di[3 * c + 2] * r3[2];
    }
    return true;
}

namespace {
struct AbiTrace {                                       // YGZ_HOST_TRACE=1: host clock per step of ygz_hip_ba_optimize_chi2
    bool on = getenv("YGZ_HOST_TRACE") && atoi(getenv("YGZ_HOST_TRACE")) != 0;
    double ms[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }; long calls = 0;
    std::chrono::steady_clock::time_point t;
    void start() { if (on) t = std::chrono::steady_clock::now(); }
    void lap(int k) { if (on) { const auto n = std::chrono::steady_clock::now(); ms[k] += std::chrono::duration<double, std::milli>(n - t).count(); t = n; } }
    ~AbiTrace() { if (on && calls) fprintf(stderr, "ygz_hip_ba_optimize_chi2 x %ld: upload %.3f  optimize_resident %.3f  get_state %.3f  linearize %.3f  download %.3f ms per call\n",
                                            calls, ms[0] / calls, ms[1] / calls, ms[2] / calls, ms[3] / calls, ms[4] / calls); }
};
AbiTrace g_abi_trace;
}
extern "C" int ygz_hip_ba_optimize(ygz_hip_ctx *ctx, const ygz_ba_problem *pb, double *poses_io, double *points_io,
                                   int max_iterations, ygz_ba_stats *stats)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || !pb || !poses_io || !points_io || max_iterations < 0) return YGZ_E_INVALID;
    if (pb->formulation != 0) return YGZ_E_INVALID;       // the g2o path of the live tree
    const int K = pb->n_poses, P = pb->n_points, E = pb->n_edges, W = 1022;
    ygz_ba_problem prob = *pb;
    prob.poses = poses_io; prob.points = points_io;
    g_abi_trace.start();
    int rc = ygz_hip_ba_upload(ctx, W, &prob);
    g_abi_trace.lap(0);
    if (rc != YGZ_OK) return rc;
    {   // the loop runs entirely on the GPU when the reduced system fits LDS (ba_resident_lm.hip); YGZ_BA_HOST_LOOP=1 forces
        // the host-side Schur / Cholesky below (kept as the large-window path and as a cross-check)
        int Kfree = 0;
        for (int k = 0; k < K; ++k) if (!(pb->pose_fixed && pb->pose_fixed[k])) ++Kfree;
        const char *force = getenv("YGZ_BA_HOST_LOOP");
        const bool dup = ygz_ba_window_has_dup(ctx, W), forced = force && force[0] == '1';
        // which loop ran and why, for ygz_hip_ba_last_path: the host loop is ~10x slower and the caller should be able to tell
        ctx->ba_last_path = (Kfree <= 20 && !forced && !dup) ? YGZ_BA_PATH_RESIDENT
                            : (YGZ_BA_PATH_HOST_LOOP | (Kfree > 20 ? YGZ_BA_WHY_FREE_POSES : 0) | (dup ? YGZ_BA_WHY_REPEATED_EDGES : 0) | (forced ? YGZ_BA_WHY_FORCED : 0));
        if (Kfree <= 20 && !forced && !dup) {      // (20 free poses: LM_MAXKF of ba_resident_lm.hip)  repeated (point, pose) pairs: host-side Schur sums them per edge
            ygz_ba_stats st;
            if ((rc = ygz_hip_ba_optimize_resident(ctx, W, 1, max_iterations, &st)) != YGZ_OK) return rc;
            g_abi_trace.lap(1);
            if ((rc = ygz_hip_ba_get_state(ctx, W, poses_io, points_io)) != YGZ_OK) return rc;
            g_abi_trace.lap(2);
            if (stats) *stats = st;
            return YGZ_OK;
        }
    }
    BlockSystem BS; block_system_init(BS, pb);
    const std::vector<int> &free_idx = BS.free_idx;
    std::vector<double> Hpp((size_t)K * 36), bp((size_t)K * 6), Hll((size_t)P * 9), bl((size_t)P * 3), Hpl((size_t)std::max(E, 1) * 18);
    std::vector<double> poses(poses_io, poses_io + (size_t)K * 6), points(points_io, points_io + (size_t)P * 3);
    std::vector<double> poses_bk, points_bk, xp, xl, dp((size_t)K * 6), dl((size_t)P * 3);
    double lambda = 0, ni = 2, currentChi = 0;
    ygz_ba_stats st; memset(&st, 0, sizeof(st));
    for (int it = 0; it < max_iterations; ++it) {
        // computeActiveErrors + buildSystem at the current state (GPU)
        if ((rc = ygz_hip_ba_set_state(ctx, W, poses.data(), points.data())) != YGZ_OK) return rc;
        if ((rc = ygz_hip_ba_linearize_resident(ctx, W, 1)) != YGZ_OK) return rc;
        if ((rc = ygz_hip_ba_download(ctx, W, Hpp.data(), bp.data(), Hll.data(), bl.data(), Hpl.data(), nullptr, nullptr, &currentChi)) != YGZ_OK) return rc;
        if (it == 0) {
            st.chi2_initial = currentChi;
            double mx = 0;                                  // computeLambdaInit: tau * max |diag| over the active vertices
            for (int k = 0; k < K; ++k) if (free_idx[k] >= 0) for (int d = 0; d < 6; ++d) mx = std::max(mx, fabs(Hpp[(size_t)k * 36 + 7 * d]));
            for (int l = 0; l < P; ++l) for (int d = 0; d < 3; ++d) mx = std::max(mx, fabs(Hll[(size_t)l * 9 + 4 * d]));
            lambda = 1e-5 * mx; ni = 2;
        }
        double rho = 0; int qmax = 0;
        do {
            poses_bk = poses; points_bk = points;           // _optimizer->push()
            std::fill(dp.begin(), dp.end(), lambda); std::fill(dl.begin(), dl.end(), lambda);
            const bool ok2 = block_solve(BS, Hpp.data(), Hll.data(), Hpl.data(), bp.data(), bl.data(), dp.data(), dl.data(), xp, xl);
            if (ok2) {                                      // _optimizer->update(x)
                for (int k = 0; k < K; ++k) if (free_idx[k] >= 0) oplus_pose(&poses[(size_t)k * 6], &xp[(size_t)free_idx[k] * 6]);
                for (size_t i = 0; i < points.size(); ++i) points[i] += xl[i];
            }
            // ---- computeActiveErrors at the trial state (GPU)
            double tempChi = DBL_MAX;
            if (ok2) {
                if ((rc = ygz_hip_ba_set_state(ctx, W, poses.data(), points.data())) != YGZ_OK) return rc;
                if ((rc = ygz_hip_ba_linearize_resident(ctx, W, 1)) != YGZ_OK) return rc;
                if ((rc = ygz_hip_ba_download(ctx, W, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &tempChi)) != YGZ_OK) return rc;
            }
            rho = currentChi - tempChi;
            double scale = 0;                               // computeScale
            if (ok2) {
                for (int k = 0; k < K; ++k) if (free_idx[k] >= 0) for (int d = 0; d < 6; ++d) { const double x = xp[(size_t)free_idx[k] * 6 + d]; scale += x * (lambda * x + bp[(size_t)k * 6 + d]); }
                for (size_t i = 0; i < xl.size(); ++i) scale += xl[i] * (lambda * xl[i] + bl[i]);
            }
            scale += 1e-3;
            rho /= scale;
            ++st.lm_trials;
            if (rho > 0 && std::isfinite(tempChi)) {        // good step
                double alpha = 1. - pow((2 * rho - 1), 3);
                alpha = std::min(alpha, 2. / 3.);
                lambda *= std::max(1. / 3., alpha);
                ni = 2; currentChi = tempChi;
            } else {                                        // bad step: restore
                lambda *= ni; ni *= 2;
                poses = poses_bk; points = points_bk;
                if (!std::isfinite(lambda)) break;
            }
            qmax++;
        } while (rho < 0 && qmax < 10);
        ++st.iterations;
        if (qmax == 10 || rho == 0 || !std::isfinite(lambda)) break;       // Terminate
    }
    st.chi2_final = currentChi; st.lambda_final = lambda;
    memcpy(poses_io, poses.data(), poses.size() * 8); memcpy(points_io, points.data(), points.size() * 8);
    if (stats) *stats = st;
    return YGZ_OK;
}

// ba::LocalBAG2O's optimize(20) AND its inlier pass (BA.cpp:501-515) in one call: the window the loop ran on is still resident, so the
// per-edge chi2 at the final state is one more linearisation and ONE array back -- the class surface used to upload the whole graph a
// second time through ygz_hip_ba_linearize for it
extern "C" int ygz_hip_ba_optimize_chi2(ygz_hip_ctx *ctx, const ygz_ba_problem *pb, double *poses_io, double *points_io, int max_iterations,
                                        ygz_ba_stats *stats, double *chi2_edge)
{
    const int W = 1022;                                       // the window ygz_hip_ba_optimize uploads
    int rc = YGZ_OK;
    if (ctx && pb && poses_io && points_io && chi2_edge && max_iterations >= 0 && pb->formulation == 0) {
        // the resident loop (the window of LocalBAG2O): upload, LM kernel, one more linearisation and the unpacking of its chi2 are queued without a
        // wait in between; statistics, state and chi2 come back in one transfer (ygz_ba_fetch_result)
        YgzDeviceGuard dg_(ctx);
        int Kfree = 0;
        for (int k = 0; k < pb->n_poses; ++k) if (!(pb->pose_fixed && pb->pose_fixed[k])) ++Kfree;
        const char *force = getenv("YGZ_BA_HOST_LOOP");
        if (Kfree <= 20 && !(force && force[0] == '1')) {
            ygz_ba_problem prob = *pb;
            prob.poses = poses_io; prob.points = points_io;
            g_abi_trace.start();
            if ((rc = ygz_hip_ba_upload(ctx, W, &prob)) != YGZ_OK) return rc;
            g_abi_trace.lap(0);
            if (!ygz_ba_window_has_dup(ctx, W)) {              // (repeated (point, pose) pairs: the host loop below, which uploads again)
                ctx->ba_last_path = YGZ_BA_PATH_RESIDENT;
                if ((rc = ygz_hip_ba_optimize_resident(ctx, W, 1, max_iterations, nullptr)) != YGZ_OK) return rc;
                const void *d_stats = ctx->scratch[SCR_BA_0];  // where the launch keeps its statistics record
                g_abi_trace.lap(1);
                if ((rc = ygz_hip_ba_linearize_resident(ctx, W, 1)) != YGZ_OK) return rc;
                g_abi_trace.lap(3);
                ygz_ba_stats st;
                if ((rc = ygz_ba_fetch_result(ctx, W, d_stats, &st, poses_io, points_io, chi2_edge)) != YGZ_OK) return rc;
                g_abi_trace.lap(4); ++g_abi_trace.calls;
                if (st.iterations < 0) { ctx->last_hip_error = (int)hipErrorLaunchTimeOut; return YGZ_E_HIP; }     // a team member never reached a barrier
                if (stats) *stats = st;
                return YGZ_OK;
            }
        }
    }
    rc = ygz_hip_ba_optimize(ctx, pb, poses_io, points_io, max_iterations, stats);
    if (rc != YGZ_OK || !chi2_edge) return rc;
    g_abi_trace.start();
    if (ygz_hip_ba_last_path(ctx) != YGZ_BA_PATH_RESIDENT && (rc = ygz_hip_ba_set_state(ctx, W, poses_io, points_io)) != YGZ_OK) return rc;   // (the host loop leaves its last TRIAL state there)
    if ((rc = ygz_hip_ba_linearize_resident(ctx, W, 1)) != YGZ_OK) return rc;
    g_abi_trace.lap(3);
    rc = ygz_hip_ba_download(ctx, W, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, chi2_edge, nullptr);
    g_abi_trace.lap(4); ++g_abi_trace.calls;
    return rc;
}

// ---------------------------------------------------------------------------------------------------------------------
// B6/B7 -- ceres::Solve with the reference's (default) options: trust-region Levenberg-Marquardt
// [frozen spec of ceres-solver 1.13 trust_region_minimizer.cc / levenberg_marquardt_strategy.cc, restated in
// oracle/ceres_ba.c::yo_ceres_solve, which this mirrors step by step].  Every evaluation (cost, residuals, Jacobians,
// blocks) is a GPU linearisation of the resident window; the candidate's linearisation becomes the next iterate's when
// the step is accepted, so an accepted iteration costs exactly one launch group.
extern "C" void ygz_hip_ceres_default_options(ygz_ceres_options *o)
{
    if (!o) return;
    o->max_num_iterations = 50;
    o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-8;
    o->initial_trust_region_radius = 1e4; o->max_trust_region_radius = 1e16; o->min_trust_region_radius = 1e-32;
    o->min_relative_decrease = 1e-3; o->min_lm_diagonal = 1e-6; o->max_lm_diagonal = 1e32;
    o->jacobi_scaling = 1; o->max_num_consecutive_invalid_steps = 5; o->fail_behind_camera = 0;
    o->trust_region_strategy = YGZ_CERES_LEVENBERG_MARQUARDT;
}

extern "C" int ygz_hip_ba_solve_ceres(ygz_hip_ctx *ctx, const ygz_ba_problem *pb, double *poses_io, double *points_io,
                                      const ygz_ceres_options *opt_in, ygz_ceres_summary *summary)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || !pb || !poses_io || !points_io) return YGZ_E_INVALID;
    if (pb->formulation != 2) return YGZ_E_INVALID;
    ygz_ceres_options opt;
    if (opt_in) opt = *opt_in; else ygz_hip_ceres_default_options(&opt);
    const int K = pb->n_poses, P = pb->n_points, E = pb->n_edges, W = 1021;
    ygz_ba_problem prob = *pb;
    prob.poses = poses_io; prob.points = points_io;
    int rc = ygz_hip_ba_upload(ctx, W, &prob);
    if (rc != YGZ_OK) return rc;
    {   // the whole trust-region loop on the GPU when the reduced system fits LDS (ba_resident_lm.hip::k_ba_ceres); YGZ_BA_HOST_LOOP=1
        // forces the loop below (reduced system on the host; also the path for repeated (point, pose) edges and large windows)
        int Kfree = 0;
        for (int k = 0; k < K; ++k) if (!(pb->pose_fixed && pb->pose_fixed[k])) ++Kfree;
        const char *force = getenv("YGZ_BA_HOST_LOOP");
        const bool dup = ygz_ba_window_has_dup(ctx, W), big = Kfree > 14 || K > 16;
        const bool forced = (force && force[0] == '1') || opt.trust_region_strategy == YGZ_CERES_DOGLEG;      // (the resident kernel has the Levenberg-Marquardt strategy only)
        ctx->ba_last_path = (!big && !forced && !dup) ? YGZ_BA_PATH_RESIDENT
                            : (YGZ_BA_PATH_HOST_LOOP | (big ? YGZ_BA_WHY_FREE_POSES : 0) | (dup ? YGZ_BA_WHY_REPEATED_EDGES : 0) | (forced ? YGZ_BA_WHY_FORCED : 0));
        if (!big && !forced && !dup) {
            ygz_ceres_summary sm;
            if ((rc = ygz_hip_ba_solve_ceres_resident(ctx, W, 1, &opt, &sm)) != YGZ_OK) return rc;
            if ((rc = ygz_hip_ba_get_state(ctx, W, poses_io, points_io)) != YGZ_OK) return rc;
            if (summary) *summary = sm;
            return YGZ_OK;
        }
    }
    BlockSystem BS; block_system_init(BS, pb);
    const std::vector<int> &fidx = BS.free_idx;
    std::vector<double> Hpp((size_t)K * 36), bp((size_t)K * 6), Hll((size_t)P * 9), bl((size_t)P * 3), Hpl((size_t)std::max(E, 1) * 18);
    std::vector<double> sHpp(Hpp.size()), sbp(bp.size()), sHll(Hll.size()), sbl(bl.size()), sHpl(Hpl.size());
    std::vector<double> scp((size_t)K * 6, 1.0), scl((size_t)P * 3, 1.0), dp((size_t)K * 6), dl((size_t)P * 3), xp, xl;
    std::vector<double> poses(poses_io, poses_io + (size_t)K * 6), points(points_io, points_io + (size_t)P * 3), cposes, cpoints;
    std::vector<double> dxp((size_t)K * 6), dxl((size_t)P * 3);
    // DoglegStrategy (TRADITIONAL_DOGLEG; oracle/ceres_ba.c: yo_ceres_solve is the same loop on the CPU): diagonal_, gradient_, gauss_newton_step_ in
    // the coordinates scaled by diagonal_, the regulariser mu_, the Cauchy step length alpha_, reuse_ after a rejected step
    const bool dogleg = opt.trust_region_strategy == YGZ_CERES_DOGLEG;
    std::vector<double> dgp((size_t)K * 6, 1.0), dgl((size_t)P * 3, 1.0), grp((size_t)K * 6, 0.0), grl((size_t)P * 3, 0.0), gnp((size_t)K * 6, 0.0), gnl((size_t)P * 3, 0.0);
    double dl_mu = 1e-8, dl_alpha = 0, dl_step_norm = 0;
    bool dl_reuse = false, dl_gn_ok = false;
    ygz_ceres_summary S; memset(&S, 0, sizeof(S));
    double x_cost = 0, radius = opt.initial_trust_region_radius, decrease_factor = 2.0, x_norm = 0, gmax = 0;
    int invalid_run = 0, term = YGZ_CERES_NO_CONVERGENCE;
    bool hard_error = false;

    // evaluate at (ps, ls); cost = chi2 / 2; 0 ok, 1 = the functor reported failure, < 0 = ABI error
    auto evaluate = [&](const std::vector<double> &ps, const std::vector<double> &ls, double *cost) -> int {
        int r;
        if ((r = ygz_hip_ba_set_state(ctx, W, ps.data(), ls.data())) != YGZ_OK) return r;
        if ((r = ygz_hip_ba_linearize_resident(ctx, W, 1)) != YGZ_OK) return r;
        double chi2 = 0;
        if ((r = ygz_hip_ba_download(ctx, W, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &chi2)) != YGZ_OK) return r;
        *cost = 0.5 * chi2;
        if (opt.fail_behind_camera) {
            int nb = 0;
            if ((r = ygz_hip_ba_behind_camera(ctx, W, &nb)) != YGZ_OK) return r;
            if (nb > 0) return 1;
        }
        return std::isfinite(chi2) ? YGZ_OK : 1;
    };
    auto fetch_blocks = [&]() -> int {
        return ygz_hip_ba_download(ctx, W, Hpp.data(), bp.data(), Hll.data(), bl.data(), Hpl.data(), nullptr, nullptr, nullptr);
    };
    auto norm_and_gradient = [&]() {
        double s2 = 0; gmax = 0;
        for (int k = 0; k < K; ++k) if (fidx[k] >= 0) for (int d = 0; d < 6; ++d) { const double v = poses[(size_t)k * 6 + d]; s2 += v * v; gmax = std::max(gmax, fabs(bp[(size_t)k * 6 + d])); }
        for (int l = 0; l < P; ++l) if (BS.lfree[l]) for (int d = 0; d < 3; ++d) { const double v = points[(size_t)l * 3 + d]; s2 += v * v; gmax = std::max(gmax, fabs(bl[(size_t)l * 3 + d])); }
        x_norm = sqrt(s2);
    };

    do {   // single-pass block so that every exit path reaches the common epilogue
        // IterationZero
        rc = evaluate(poses, points, &x_cost);
        if (rc < 0) { hard_error = true; break; }
        if (rc == 1) { term = YGZ_CERES_FAILURE; break; }
        if ((rc = fetch_blocks()) != YGZ_OK) { hard_error = true; break; }
        S.initial_cost = x_cost;
        if (opt.jacobi_scaling) {
            for (int k = 0; k < K; ++k) for (int d = 0; d < 6; ++d) scp[(size_t)k * 6 + d] = 1.0 / (1.0 + sqrt(Hpp[(size_t)k * 36 + 7 * d]));
            for (int l = 0; l < P; ++l) for (int d = 0; d < 3; ++d) scl[(size_t)l * 3 + d] = 1.0 / (1.0 + sqrt(Hll[(size_t)l * 9 + 4 * d]));
        }
        norm_and_gradient();

        for (;;) {
            if (S.iterations >= opt.max_num_iterations) { term = YGZ_CERES_NO_CONVERGENCE; break; }
            if (gmax <= opt.gradient_tolerance) { term = YGZ_CERES_GRADIENT_TOLERANCE; break; }
            if (radius <= opt.min_trust_region_radius) { term = YGZ_CERES_MIN_RADIUS; break; }
            ++S.iterations;
            // the column-scaled system
            auto clampd = [&](double v) { return std::min(std::max(v, opt.min_lm_diagonal), opt.max_lm_diagonal); };
            if (!dogleg || !dl_reuse) {
                for (int k = 0; k < K; ++k) for (int r = 0; r < 6; ++r) {
                    for (int c = 0; c < 6; ++c) sHpp[(size_t)k * 36 + 6 * r + c] = Hpp[(size_t)k * 36 + 6 * r + c] * scp[(size_t)k * 6 + r] * scp[(size_t)k * 6 + c];
                    sbp[(size_t)k * 6 + r] = bp[(size_t)k * 6 + r] * scp[(size_t)k * 6 + r];
                }
                for (int l = 0; l < P; ++l) for (int r = 0; r < 3; ++r) {
                    for (int c = 0; c < 3; ++c) sHll[(size_t)l * 9 + 3 * r + c] = Hll[(size_t)l * 9 + 3 * r + c] * scl[(size_t)l * 3 + r] * scl[(size_t)l * 3 + c];
                    sbl[(size_t)l * 3 + r] = bl[(size_t)l * 3 + r] * scl[(size_t)l * 3 + r];
                }
                for (int e = 0; e < E; ++e) {
                    const int ip = pb->edge_pose[e], il = pb->edge_point[e];
                    for (int r = 0; r < 6; ++r) for (int c = 0; c < 3; ++c)
                        sHpl[(size_t)e * 18 + 3 * r + c] = Hpl[(size_t)e * 18 + 3 * r + c] * scp[(size_t)ip * 6 + r] * scl[(size_t)il * 3 + c];
                }
            }
            bool valid;
            if (!dogleg) {
                // LevenbergMarquardtStrategy::ComputeStep
                for (int k = 0; k < K; ++k) for (int r = 0; r < 6; ++r) dp[(size_t)k * 6 + r] = clampd(sHpp[(size_t)k * 36 + 7 * r]) / radius;
                for (int l = 0; l < P; ++l) for (int r = 0; r < 3; ++r) dl[(size_t)l * 3 + r] = clampd(sHll[(size_t)l * 9 + 4 * r]) / radius;
                valid = block_solve(BS, sHpp.data(), sHll.data(), sHpl.data(), sbp.data(), sbl.data(), dp.data(), dl.data(), xp, xl);
            } else {
                // DoglegStrategy::ComputeStep
                if (!dl_reuse) {
                    dl_reuse = true;
                    double g2 = 0;
                    for (int k = 0; k < K; ++k) for (int r = 0; r < 6; ++r) {
                        dgp[(size_t)k * 6 + r] = sqrt(clampd(sHpp[(size_t)k * 36 + 7 * r]));
                        grp[(size_t)k * 6 + r] = fidx[k] >= 0 ? -sbp[(size_t)k * 6 + r] / dgp[(size_t)k * 6 + r] : 0.0;      // J^T r = -b
                        g2 += grp[(size_t)k * 6 + r] * grp[(size_t)k * 6 + r];
                    }
                    for (int l = 0; l < P; ++l) for (int r = 0; r < 3; ++r) {
                        dgl[(size_t)l * 3 + r] = sqrt(clampd(sHll[(size_t)l * 9 + 4 * r]));
                        grl[(size_t)l * 3 + r] = BS.lfree[l] ? -sbl[(size_t)l * 3 + r] / dgl[(size_t)l * 3 + r] : 0.0;
                        g2 += grl[(size_t)l * 3 + r] * grl[(size_t)l * 3 + r];
                    }
                    // ComputeCauchyPoint: |J_scaled v|^2 with v = D^-1 gradient_, as the quadratic form v^T (J_s^T J_s) v of the blocks (the oracle multiplies
                    // the per-edge Jacobians: same quantity, another order of the sums)
                    std::vector<double> vp((size_t)K * 6), vl((size_t)P * 3);
                    for (size_t i = 0; i < vp.size(); ++i) vp[i] = grp[i] / dgp[i];
                    for (size_t i = 0; i < vl.size(); ++i) vl[i] = grl[i] / dgl[i];
                    double jg2 = 0;
                    for (int k = 0; k < K; ++k) if (fidx[k] >= 0) for (int r = 0; r < 6; ++r) {
                        double hv = 0; for (int c2 = 0; c2 < 6; ++c2) hv += sHpp[(size_t)k * 36 + 6 * r + c2] * vp[(size_t)k * 6 + c2];
                        jg2 += vp[(size_t)k * 6 + r] * hv;
                    }
                    for (int l = 0; l < P; ++l) if (BS.lfree[l]) for (int r = 0; r < 3; ++r) {
                        double hv = 0; for (int c2 = 0; c2 < 3; ++c2) hv += sHll[(size_t)l * 9 + 3 * r + c2] * vl[(size_t)l * 3 + c2];
                        jg2 += vl[(size_t)l * 3 + r] * hv;
                    }
                    for (int e = 0; e < E; ++e) {
                        const double *h = &sHpl[(size_t)e * 18], *a = &vp[(size_t)pb->edge_pose[e] * 6], *b2 = &vl[(size_t)pb->edge_point[e] * 3];
                        for (int r = 0; r < 6; ++r) jg2 += 2.0 * a[r] * (h[3 * r] * b2[0] + h[3 * r + 1] * b2[1] + h[3 * r + 2] * b2[2]);
                    }
                    dl_alpha = g2 / jg2;
                    // ComputeGaussNewtonStep: (J^T J + mu_ D^2) y = J^T r, mu_ x 10 while the factorisation fails
                    dl_gn_ok = false;
                    while (dl_mu < 1.0) {
                        for (size_t i = 0; i < dp.size(); ++i) dp[i] = dgp[i] * dgp[i] * dl_mu;
                        for (size_t i = 0; i < dl.size(); ++i) dl[i] = dgl[i] * dgl[i] * dl_mu;
                        bool ok = block_solve(BS, sHpp.data(), sHll.data(), sHpl.data(), sbp.data(), sbl.data(), dp.data(), dl.data(), xp, xl);
                        if (ok) { for (double v : xp) if (!std::isfinite(v)) ok = false; for (double v : xl) if (!std::isfinite(v)) ok = false; }
                        if (!ok) { dl_mu *= 10.0; continue; }
                        dl_gn_ok = true;
                        break;
                    }
                    if (dl_gn_ok) {
                        for (int k = 0; k < K; ++k) for (int d = 0; d < 6; ++d) gnp[(size_t)k * 6 + d] = fidx[k] >= 0 ? xp[(size_t)fidx[k] * 6 + d] * dgp[(size_t)k * 6 + d] : 0.0;
                        for (int l = 0; l < P; ++l) for (int d = 0; d < 3; ++d) gnl[(size_t)l * 3 + d] = BS.lfree[l] ? xl[(size_t)l * 3 + d] * dgl[(size_t)l * 3 + d] : 0.0;
                    }
                }
                valid = dl_gn_ok;
                if (valid) {
                    // ComputeTraditionalDoglegStep
                    double g2 = 0, n2 = 0, gdn = 0;
                    for (size_t i = 0; i < grp.size(); ++i) { g2 += grp[i] * grp[i]; n2 += gnp[i] * gnp[i]; gdn += grp[i] * gnp[i]; }
                    for (size_t i = 0; i < grl.size(); ++i) { g2 += grl[i] * grl[i]; n2 += gnl[i] * gnl[i]; gdn += grl[i] * gnl[i]; }
                    const double gradient_norm = sqrt(g2), gauss_newton_norm = sqrt(n2);
                    double cg, cn;
                    if (gauss_newton_norm <= radius) { cg = 0.0; cn = 1.0; dl_step_norm = gauss_newton_norm; }
                    else if (gradient_norm * dl_alpha >= radius) { cg = -(radius / gradient_norm); cn = 0.0; dl_step_norm = radius; }
                    else {
                        const double b_dot_a = -dl_alpha * gdn, a_squared_norm = (dl_alpha * gradient_norm) * (dl_alpha * gradient_norm);
                        const double b_minus_a_squared_norm = a_squared_norm - 2 * b_dot_a + gauss_newton_norm * gauss_newton_norm;
                        const double c3 = b_dot_a - a_squared_norm;
                        const double d3 = sqrt(c3 * c3 + b_minus_a_squared_norm * (radius * radius - a_squared_norm));
                        const double beta = (c3 <= 0) ? (d3 - c3) / b_minus_a_squared_norm : (radius * radius - a_squared_norm) / (d3 + c3);
                        cg = -dl_alpha * (1.0 - beta); cn = beta;
                        double s2 = 0;
                        for (size_t i = 0; i < grp.size(); ++i) { const double v = cg * grp[i] + cn * gnp[i]; s2 += v * v; }
                        for (size_t i = 0; i < grl.size(); ++i) { const double v = cg * grl[i] + cn * gnl[i]; s2 += v * v; }
                        dl_step_norm = sqrt(s2);
                    }
                    // the step in the minimizer's (column-scaled) coordinates, laid out as block_solve leaves its solution
                    xp.assign((size_t)std::max(BS.Kf, 1) * 6, 0.0); xl.assign((size_t)P * 3, 0.0);
                    for (int k = 0; k < K; ++k) if (fidx[k] >= 0) for (int d = 0; d < 6; ++d)
                        xp[(size_t)fidx[k] * 6 + d] = (cg * grp[(size_t)k * 6 + d] + cn * gnp[(size_t)k * 6 + d]) / dgp[(size_t)k * 6 + d];
                    for (int l = 0; l < P; ++l) if (BS.lfree[l]) for (int d = 0; d < 3; ++d)
                        xl[(size_t)l * 3 + d] = (cg * grl[(size_t)l * 3 + d] + cn * gnl[(size_t)l * 3 + d]) / dgl[(size_t)l * 3 + d];
                }
            }
            double model_cost_change = 0;
            if (valid) {
                std::fill(dxp.begin(), dxp.end(), 0.0);
                for (int k = 0; k < K; ++k) if (fidx[k] >= 0) for (int d = 0; d < 6; ++d) {
                    const double v = xp[(size_t)fidx[k] * 6 + d] * scp[(size_t)k * 6 + d];
                    if (!std::isfinite(v)) valid = false;
                    dxp[(size_t)k * 6 + d] = v;
                }
                for (int l = 0; l < P; ++l) for (int d = 0; d < 3; ++d) {
                    const double v = xl[(size_t)l * 3 + d] * scl[(size_t)l * 3 + d];
                    if (!std::isfinite(v)) valid = false;
                    dxl[(size_t)l * 3 + d] = v;
                }
            }
            if (valid) {      // model_cost_change = -(J d).(r + J d / 2) = d.b - d.H d / 2, from the blocks
                for (int k = 0; k < K; ++k) if (fidx[k] >= 0) for (int r = 0; r < 6; ++r) {
                    double hd = 0;
                    for (int c = 0; c < 6; ++c) hd += Hpp[(size_t)k * 36 + 6 * r + c] * dxp[(size_t)k * 6 + c];
                    model_cost_change += dxp[(size_t)k * 6 + r] * (bp[(size_t)k * 6 + r] - 0.5 * hd);
                }
                for (int l = 0; l < P; ++l) if (BS.lfree[l]) for (int r = 0; r < 3; ++r) {
                    double hd = 0;
                    for (int c = 0; c < 3; ++c) hd += Hll[(size_t)l * 9 + 3 * r + c] * dxl[(size_t)l * 3 + c];
                    model_cost_change += dxl[(size_t)l * 3 + r] * (bl[(size_t)l * 3 + r] - 0.5 * hd);
                }
                for (int e = 0; e < E; ++e) {
                    const double *h = &Hpl[(size_t)e * 18], *a = &dxp[(size_t)pb->edge_pose[e] * 6], *b = &dxl[(size_t)pb->edge_point[e] * 3];
                    for (int r = 0; r < 6; ++r) model_cost_change -= a[r] * (h[3 * r] * b[0] + h[3 * r + 1] * b[1] + h[3 * r + 2] * b[2]);
                }
                if (!(model_cost_change > 0)) valid = false;
            }
            if (!valid) {     // HandleInvalidStep
                if (++invalid_run >= opt.max_num_consecutive_invalid_steps) { term = YGZ_CERES_FAILURE; break; }
                if (dogleg) { dl_mu *= 10.0; dl_reuse = false; }        // DoglegStrategy::StepIsInvalid
                else radius *= 0.5;
                ++S.unsuccessful_steps;
                continue;
            }
            invalid_run = 0;
            cposes = poses; cpoints = points;
            double step2 = 0;
            for (size_t i = 0; i < cposes.size(); ++i) { cposes[i] += dxp[i]; step2 += dxp[i] * dxp[i]; }
            for (size_t i = 0; i < cpoints.size(); ++i) { cpoints[i] += dxl[i]; step2 += dxl[i] * dxl[i]; }
            double cand_cost = DBL_MAX;
            rc = evaluate(cposes, cpoints, &cand_cost);
            if (rc < 0) { hard_error = true; break; }
            if (rc != YGZ_OK) cand_cost = DBL_MAX;              // evaluation failure = a step of very high cost
            if (sqrt(step2) <= opt.parameter_tolerance * (x_norm + opt.parameter_tolerance)) { term = YGZ_CERES_PARAMETER_TOLERANCE; break; }
            const double cost_change = x_cost - cand_cost;
            if (fabs(cost_change) <= opt.function_tolerance * x_cost) { term = YGZ_CERES_FUNCTION_TOLERANCE; break; }
            const double relative_decrease = cost_change / model_cost_change;
            if (relative_decrease > opt.min_relative_decrease) {              // HandleSuccessfulStep
                poses = cposes; points = cpoints; x_cost = cand_cost;        // the window already holds this state's blocks
                if ((rc = fetch_blocks()) != YGZ_OK) { hard_error = true; break; }
                norm_and_gradient();
                if (dogleg) {                                                 // DoglegStrategy::StepAccepted
                    if (relative_decrease < 0.25) radius *= 0.5;
                    if (relative_decrease > 0.75) radius = std::max(radius, 3.0 * dl_step_norm);
                    radius = std::min(radius, opt.max_trust_region_radius);
                    dl_mu = std::max(1e-8, 2.0 * dl_mu / 10.0);
                    dl_reuse = false;
                } else {
                    double t = 2.0 * relative_decrease - 1.0;
                    t = 1.0 - t * t * t;
                    radius = std::min(radius / std::max(1.0 / 3.0, t), opt.max_trust_region_radius);
                    decrease_factor = 2.0;
                }
                ++S.successful_steps;
            } else {                                                          // StepRejected
                if (dogleg) { radius *= 0.5; dl_reuse = true; }
                else { radius = radius / decrease_factor; decrease_factor *= 2.0; }
                ++S.unsuccessful_steps;
            }
        }
    } while (0);
    if (hard_error) return rc;
    S.termination = term; S.final_cost = x_cost; S.final_radius = radius;
    memcpy(poses_io, poses.data(), poses.size() * 8); memcpy(points_io, points.data(), points.size() * 8);
    if (summary) *summary = S;
    return YGZ_OK;
}

// which loop the last ygz_hip_ba_optimize / ygz_hip_ba_solve_ceres of this context ran: YGZ_BA_PATH_RESIDENT (the whole loop on the GPU) or
// YGZ_BA_PATH_HOST_LOOP | reason bits (linearisations on the GPU, reduced system on the host: about ten times slower); 0: none yet
extern "C" int ygz_hip_ba_light_barrier(const ygz_hip_ctx *ctx) { return ctx ? ctx->lm_light_barrier : -1; }
extern "C" int ygz_hip_ba_last_path(const ygz_hip_ctx *ctx)
{
    return ctx ? ctx->ba_last_path : 0;
}
