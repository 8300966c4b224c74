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
#include "se3_dev.h"
#include <vector>
#include <cmath>
#include <cfloat>
#include <string.h>

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

extern "C" int ygz_hip_ba_optimize(ygz_hip_ctx *ctx, const ygz_ba_problem *pb, double *poses_io, double *points_io,
                                   int max_iterations, ygz_ba_stats *stats)
{
    if (!ctx || !pb || !poses_io || !points_io || max_iterations < 0) return YGZ_E_INVALID;
    if (pb->formulation != 0) return YGZ_E_INVALID;       // the g2o path of the live tree
    const int K = pb->n_poses, P = pb->n_points, E = pb->n_edges, W = 1022;
    ygz_ba_problem prob = *pb;
    prob.poses = poses_io; prob.points = points_io;
    int rc = ygz_hip_ba_upload(ctx, W, &prob);
    if (rc != YGZ_OK) return rc;
    std::vector<int> free_idx(K, -1);
    int Kf = 0;
    for (int k = 0; k < K; ++k) if (!(pb->pose_fixed && pb->pose_fixed[k])) free_idx[k] = Kf++;
    std::vector<std::vector<int>> pt_edges(P);
    for (int e = 0; e < E; ++e) pt_edges[pb->edge_point[e]].push_back(e);
    std::vector<double> Hpp((size_t)K * 36), bp((size_t)K * 6), Hll((size_t)P * 9), bl((size_t)P * 3), Hpl((size_t)E * 18);
    std::vector<double> poses(poses_io, poses_io + (size_t)K * 6), points(points_io, points_io + (size_t)P * 3);
    std::vector<double> poses_bk, points_bk, xp((size_t)Kf * 6), xl((size_t)P * 3), Dinv((size_t)P * 9);
    double lambda = 0, ni = 2, currentChi = 0;
    ygz_ba_stats st; memset(&st, 0, sizeof(st));
    const int n = 6 * Kf;
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
            // ---- solve (Hpp + lambda I, Hll + lambda I) by Schur complement
            bool ok2 = true;
            std::vector<double> S((size_t)n * n, 0.0), bs((size_t)n, 0.0);
            for (int k = 0; k < K; ++k) if (free_idx[k] >= 0) {
                const int a = free_idx[k];
                for (int r = 0; r < 6; ++r) { for (int c = 0; c < 6; ++c) S[(size_t)(6 * a + r) * n + 6 * a + c] = Hpp[(size_t)k * 36 + 6 * r + c]; bs[6 * a + r] = bp[(size_t)k * 6 + r]; }
                for (int d = 0; d < 6; ++d) S[(size_t)(6 * a + d) * n + 6 * a + d] += lambda;
            }
            for (int l = 0; l < P && ok2; ++l) {
                double D[9]; memcpy(D, &Hll[(size_t)l * 9], sizeof(D));
                D[0] += lambda; D[4] += lambda; D[8] += lambda;
                if (!inv3(D, &Dinv[(size_t)l * 9])) { ok2 = false; break; }
                const double *Di = &Dinv[(size_t)l * 9];
                for (int ei : pt_edges[l]) {
                    const int a = free_idx[pb->edge_pose[ei]];
                    if (a < 0) continue;
                    const double *Bi = &Hpl[(size_t)ei * 18];           // 6x3
                    double BD[18];
                    for (int r = 0; r < 6; ++r) for (int c = 0; c < 3; ++c) BD[3 * r + c] = Bi[3 * r] * Di[c] + Bi[3 * r + 1] * Di[3 + c] + Bi[3 * r + 2] * Di[6 + c];
                    for (int r = 0; r < 6; ++r) bs[6 * a + r] -= BD[3 * r] * bl[(size_t)l * 3] + BD[3 * r + 1] * bl[(size_t)l * 3 + 1] + BD[3 * r + 2] * bl[(size_t)l * 3 + 2];
                    for (int ej : pt_edges[l]) {
                        const int b2 = free_idx[pb->edge_pose[ej]];
                        if (b2 < 0) continue;
                        const double *Bj = &Hpl[(size_t)ej * 18];
                        for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c)
                            S[(size_t)(6 * a + r) * n + 6 * b2 + c] -= BD[3 * r] * Bj[3 * c] + BD[3 * r + 1] * Bj[3 * c + 1] + BD[3 * r + 2] * Bj[3 * c + 2];
                    }
                }
            }
            if (ok2 && n > 0) { xp = bs; ok2 = chol_solve(S, xp, n); }
            if (ok2) {
                for (int l = 0; l < P; ++l) {
                    double r3[3] = { bl[(size_t)l * 3], bl[(size_t)l * 3 + 1], bl[(size_t)l * 3 + 2] };
                    for (int ei : pt_edges[l]) {
                        const int a = free_idx[pb->edge_pose[ei]];
                        if (a < 0) continue;
                        const double *Bi = &Hpl[(size_t)ei * 18];
                        for (int c = 0; c < 3; ++c) for (int r = 0; r < 6; ++r) r3[c] -= Bi[3 * r + c] * xp[6 * a + r];
                    }
                    const double *Di = &Dinv[(size_t)l * 9];
                    for (int c = 0; c < 3; ++c) xl[(size_t)l * 3 + c] = Di[3 * c] * r3[0] + Di[3 * c + 1] * r3[1] + Di[3 * c + 2] * r3[2];
                }
                // ---- _optimizer->update(x)
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
