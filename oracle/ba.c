/* ORACLE (test infrastructure) -- local-BA reprojection residual / Jacobian / JtJ build.
 * Restates include/ygz/G2oTypes.h:13-217 (live tree, pixel residual), the legacy
 * include/ygz/g2o_types.h:12-86 (normalised-plane residual, what src/optimizer.cpp
 * builds on) and the per-edge accumulation g2o performs when src/Algorithm/BA.cpp:501-502
 * calls optimize() [frozen spec of g2o BaseBinaryEdge::constructQuadraticForm +
 * RobustKernelHuber::robustify: rho'(e2)=1 if e2<=delta^2 else delta/sqrt(e2);
 * H += rho' J^T J, b += -rho' J^T e, information = I (BA.cpp:446)].
 * See ygz_oracle.h for the rules. */
#include "ygz_oracle.h"
#include <math.h>
#include <string.h>

/* vertex estimate is [omega; t]; Sophus wants [t; omega] -- G2oTypes.h:88-90 */
static void pose_to_se3(const double pose[6], yo_se3 *T)
{
    const double est[6] = { pose[3], pose[4], pose[5], pose[0], pose[1], pose[2] };
    yo_se3_exp(est, T);
}

/* EdgeSophusSE3ProjectXYZ::computeError + camProject -- G2oTypes.h:84-91,134-144 */
void yo_ba_edge_error(const double pose[6], const double pt[3], const double obs[2],
                      double fx, double fy, double cx, double cy, double err[2])
{
    yo_se3 T; double p[3];
    pose_to_se3(pose, &T);
    yo_se3_act(&T, pt, p);
    const double proj0 = p[0] / p[2], proj1 = p[1] / p[2];
    err[0] = obs[0] - (proj0 * fx + cx);
    err[1] = obs[1] - (proj1 * fy + cy);
}

/* EdgeSophusSE3ProjectXYZ::linearizeOplus -- G2oTypes.h:93-132 */
void yo_ba_edge_jacobians(const double pose[6], const double pt[3],
                          double fx, double fy, double Jp[6], double Jx[12])
{
    yo_se3 T; double p[3], R[9];
    pose_to_se3(pose, &T);
    yo_se3_act(&T, pt, p);
    yo_quat_to_R(T.q, R);
    const double x = p[0], y = p[1], z = p[2], z_2 = z * z;
    const double tmp[6] = { fx, 0, -x / z * fx, 0, fy, -y / z * fy };
    /* _jacobianOplusXi = -1./z * tmp * R   (Eigen evaluates (-1./z * tmp) * R) */
    double s[6];
    for (int i = 0; i < 6; ++i) s[i] = -1. / z * tmp[i];
    for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c)
        Jp[3 * r + c] = s[3 * r] * R[c] + s[3 * r + 1] * R[3 + c] + s[3 * r + 2] * R[6 + c];
    Jx[0] = x * y / z_2 * fx;          Jx[1] = -(1 + (x * x / z_2)) * fx;  Jx[2] = y / z * fx;
    Jx[3] = -1. / z * fx;              Jx[4] = 0;                          Jx[5] = x / z_2 * fx;
    Jx[6] = (1 + y * y / z_2) * fy;    Jx[7] = -x * y / z_2 * fy;          Jx[8] = -x / z * fy;
    Jx[9] = 0;                         Jx[10] = -1. / z * fy;              Jx[11] = y / z_2 * fy;
}

/* legacy EdgeSophusSE3ProjectXYZ::computeError -- include/ygz/g2o_types.h:45-51 */
void yo_ba_edge_error_norm(const double pose_tw[6], const double pt[3], const double obs_n[2],
                           double err[2])
{
    yo_se3 T; double p[3];
    yo_se3_exp(pose_tw, &T);
    yo_se3_act(&T, pt, p);
    err[0] = obs_n[0] - p[0] / p[2];
    err[1] = obs_n[1] - p[1] / p[2];
}

/* legacy linearizeOplus -- include/ygz/g2o_types.h:53-86 (Jx == cvutils::JacobXYZ2Cam, CVUtils.h:77-99) */
void yo_ba_edge_jacobians_norm(const double pose_tw[6], const double pt[3], double Jp[6], double Jx[12])
{
    yo_se3 T; double p[3], R[9];
    yo_se3_exp(pose_tw, &T);
    yo_se3_act(&T, pt, p);
    yo_quat_to_R(T.q, R);
    const double x = p[0], y = p[1], z = p[2], z_inv = 1. / z, z_inv_2 = z_inv * z_inv;
    const double tmp[6] = { z_inv, 0, -x * z_inv_2, 0, z_inv, -y * z_inv_2 };
    for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c)
        Jp[3 * r + c] = -tmp[3 * r] * R[c] + -tmp[3 * r + 1] * R[3 + c] + -tmp[3 * r + 2] * R[6 + c];
    Jx[0] = -z_inv;  Jx[1] = 0.0;     Jx[2] = x * z_inv_2;  Jx[3] = y * Jx[2];
    Jx[4] = -(1.0 + x * Jx[2]);       Jx[5] = y * z_inv;
    Jx[6] = 0.0;     Jx[7] = -z_inv;  Jx[8] = y * z_inv_2;  Jx[9] = 1.0 + y * Jx[8];
    Jx[10] = -Jx[3]; Jx[11] = -x * z_inv;
}

double yo_ba_linearize(const yo_ba_problem *pb, double *Hpp, double *bp, double *Hll, double *bl,
                       double *Hpl, double *err, double *chi2_edge)
{
    memset(Hpp, 0, sizeof(double) * 36 * (size_t)pb->n_poses);
    memset(bp, 0, sizeof(double) * 6 * (size_t)pb->n_poses);
    memset(Hll, 0, sizeof(double) * 9 * (size_t)pb->n_points);
    memset(bl, 0, sizeof(double) * 3 * (size_t)pb->n_points);
    if (Hpl) memset(Hpl, 0, sizeof(double) * 18 * (size_t)pb->n_edges);
    const double delta = pb->huber_delta, dsqr = delta * delta;
    double total = 0;
    for (int e = 0; e < pb->n_edges; ++e) {
        const int ip = pb->edge_pose[e], il = pb->edge_point[e];
        const double *pose = pb->poses + 6 * (size_t)ip, *pt = pb->points + 3 * (size_t)il;
        double r[2], Jp[6], Jx[12];
        yo_ba_edge_error(pose, pt, pb->obs + 2 * (size_t)e, pb->fx, pb->fy, pb->cx, pb->cy, r);
        yo_ba_edge_jacobians(pose, pt, pb->fx, pb->fy, Jp, Jx);
        const double e2 = r[0] * r[0] + r[1] * r[1];
        double rho0 = e2, rho1 = 1.0;
        if (delta > 0 && e2 > dsqr) {            /* RobustKernelHuber::robustify */
            const double sqrte = sqrt(e2);
            rho0 = 2 * sqrte * delta - dsqr;
            rho1 = delta / sqrte;
        }
        total += rho0;
        if (err) { err[2 * e] = r[0]; err[2 * e + 1] = r[1]; }
        if (chi2_edge) chi2_edge[e] = e2;
        /* point block (never fixed, BA.cpp:426-433) */
        double *hl = Hll + 9 * (size_t)il, *gl = bl + 3 * (size_t)il;
        for (int a = 0; a < 3; ++a) {
            for (int b = 0; b < 3; ++b) hl[3 * a + b] += rho1 * (Jp[a] * Jp[b] + Jp[3 + a] * Jp[3 + b]);
            gl[a] += -rho1 * (Jp[a] * r[0] + Jp[3 + a] * r[1]);
        }
        if (pb->pose_fixed && pb->pose_fixed[ip]) continue;
        double *hp = Hpp + 36 * (size_t)ip, *gp = bp + 6 * (size_t)ip;
        for (int a = 0; a < 6; ++a) {
            for (int b = 0; b < 6; ++b) hp[6 * a + b] += rho1 * (Jx[a] * Jx[b] + Jx[6 + a] * Jx[6 + b]);
            gp[a] += -rho1 * (Jx[a] * r[0] + Jx[6 + a] * r[1]);
            if (Hpl) for (int b = 0; b < 3; ++b)
                Hpl[18 * (size_t)e + 3 * a + b] = rho1 * (Jx[a] * Jp[b] + Jx[6 + a] * Jp[3 + b]);
        }
    }
    return total;
}

/* VertexSE3Sophus::oplusImpl -- G2oTypes.h:38-45 */
void yo_ba_pose_oplus(double pose[6], const double upd[6])
{
    const double v[6] = { upd[3], upd[4], upd[5], upd[0], upd[1], upd[2] };
    const double est[6] = { pose[3], pose[4], pose[5], pose[0], pose[1], pose[2] };
    yo_se3 A, B, C; double r[6];
    yo_se3_exp(v, &A); yo_se3_exp(est, &B);
    yo_se3_mul(&A, &B, &C);
    yo_se3_log(&C, r);
    pose[0] = r[3]; pose[1] = r[4]; pose[2] = r[5]; pose[3] = r[0]; pose[4] = r[1]; pose[5] = r[2];
}
