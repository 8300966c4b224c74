// Device-side building blocks of the local-BA edge stack, shared by the linearisation kernels (ba.hip) and the resident
// Levenberg-Marquardt kernel (ba_resident_lm.hip).  See ba.hip for the reference citations.
#ifndef YGZ_BA_DEV_H_
#define YGZ_BA_DEV_H_
#include "ygz_internal.h"
#include "se3_dev.h"

struct ygz_hip_ctx::BaWindow {
    int K = 0, P = 0, E = 0, formulation = 0;
    double fx = 0, fy = 0, cx = 0, cy = 0, huber = 0;
    void *blob = nullptr;            // one allocation
    double *poses, *points, *obs, *posed, *edge_tmp, *rho0, *edge_huber;
    double *Hpp, *bp, *Hll, *bl, *Hpl, *err, *chi2_edge, *chi2;
    int32_t *edge_pose, *edge_point, *pt_off, *pt_edges, *pose_off, *pose_edges, *n_behind;
    uint8_t *fixed, *point_fixed, *edge_enable;
    // work space of the resident Levenberg-Marquardt kernel (ba_resident_lm.hip)
    int Kf = 0;                      // free poses
    double *poses_bk, *points_bk, *Y, *Dinv, *xl;
    int32_t *free_idx, *free_pose, *pt_pose_edge;
};
#define BA_POSED 32      // doubles per prepared pose: q(4) t(3) R(9) J_l(9)

struct BaDev {
    int K, P, E, formulation;
    double fx, fy, cx, cy, huber;
    const double *poses, *points, *obs; double *posed, *edge_tmp, *rho0; const double *edge_huber;
    double *Hpp, *bp, *Hll, *bl, *Hpl, *err, *chi2_edge, *chi2;
    const int32_t *edge_pose, *edge_point, *pt_off, *pt_edges, *pose_off, *pose_edges; int32_t *n_behind;
    const uint8_t *fixed, *point_fixed, *edge_enable;
    int Kf;
    double *poses_w, *points_w;      // the same state arrays, writable (LM update / restore)
    double *poses_bk, *points_bk, *Y, *Dinv, *xl;
    const int32_t *free_idx, *free_pose, *pt_pose_edge;     // pose -> free index or -1; free index -> pose; [P][Kf] edge of (point, free pose) or -1
};
const BaDev *ygz_ba_table(ygz_hip_ctx *ctx, int *rc);        // device table of all uploaded windows, rebuilt when dirty

// SE3::exp once per pose instead of once per edge: (q, t, R) and, for the ceres formulation, J_l
__device__ __forceinline__ void ba_pose_prep_one(const BaDev &B, int k)
{
    const double *p = B.poses + 6 * (size_t)k;
    double *o = B.posed + BA_POSED * (size_t)k;
    if (B.formulation == 2) {        // [t; angle-axis]: R and J_l as ceres::AngleAxisRotatePoint defines the rotation
        const double ax = p[3], ay = p[4], az = p[5], theta2 = ax * ax + ay * ay + az * az;
        double *R = o + 7, *Jl = o + 16;
        o[0] = o[1] = o[2] = 0; o[3] = 1; o[4] = p[0]; o[5] = p[1]; o[6] = p[2];
        if (theta2 > 2.220446049250313e-16) {
            const double theta = sqrt(theta2), c = cos(theta), s = sin(theta), ti = 1.0 / theta;
            const double w[3] = { ax * ti, ay * ti, az * ti }, c1 = 1.0 - c, sa = s * ti, cb = c1 * ti;
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
                R[3 * i + j] = c1 * w[i] * w[j] + (i == j ? c : 0.0);
                Jl[3 * i + j] = (1.0 - sa) * w[i] * w[j] + (i == j ? sa : 0.0);
            }
            R[1] -= s * w[2]; R[2] += s * w[1]; R[3] += s * w[2]; R[5] -= s * w[0]; R[6] -= s * w[1]; R[7] += s * w[0];
            Jl[1] -= cb * w[2]; Jl[2] += cb * w[1]; Jl[3] += cb * w[2]; Jl[5] -= cb * w[0]; Jl[6] -= cb * w[1]; Jl[7] += cb * w[0];
        } else {                     // first-order branch: p + aa x p
            R[0] = 1; R[1] = -az; R[2] = ay; R[3] = az; R[4] = 1; R[5] = -ax; R[6] = -ay; R[7] = ax; R[8] = 1;
            for (int i = 0; i < 9; ++i) Jl[i] = (i % 4 == 0) ? 1.0 : 0.0;
        }
        return;
    }
    double est[6];
    if (B.formulation == 0) { est[0] = p[3]; est[1] = p[4]; est[2] = p[5]; est[3] = p[0]; est[4] = p[1]; est[5] = p[2]; }   // [omega;t] -> [t;omega], G2oTypes.h:88-90
    else { for (int i = 0; i < 6; ++i) est[i] = p[i]; }
    Se3 T;
    se3_exp_d(est, &T);
    for (int i = 0; i < 4; ++i) o[i] = T.q[i];
    for (int i = 0; i < 3; ++i) o[4 + i] = T.t[i];
    quat_to_R_d(T.q, o + 7);
}

// pd = the prepared pose (q, t, R, J_l), read only by formulation 2
__device__ __forceinline__ void ba_pose_jac(int formulation, double x, double y, double z, double fx, double fy,
                                            const double *__restrict__ pd, double Jx[12])
{
    if (formulation == 2) {          // d r / d [t; aa] of the ceres functor: [-A, A [R p]x J_l]
        const double zi = 1. / z, xz = x * zi * zi, yz = y * zi * zi;
        const double a = x - pd[4], b = y - pd[5], c = z - pd[6];           // R p_w = p_c - t
        const double *Jl = pd + 16;
        double M[9];                                                         // [R p]x J_l
        for (int j = 0; j < 3; ++j) {
            M[j] = -c * Jl[3 + j] + b * Jl[6 + j];
            M[3 + j] = c * Jl[j] - a * Jl[6 + j];
            M[6 + j] = -b * Jl[j] + a * Jl[3 + j];
        }
        Jx[0] = -zi; Jx[1] = 0.0; Jx[2] = xz;
        Jx[6] = 0.0; Jx[7] = -zi; Jx[8] = yz;
        for (int j = 0; j < 3; ++j) { Jx[3 + j] = zi * M[j] - xz * M[6 + j]; Jx[9 + j] = zi * M[3 + j] - yz * M[6 + j]; }
    } else if (formulation == 0) {          // G2oTypes.h:119-131, columns [rot(3), trans(3)]
        const double z_2 = z * z;
        Jx[0] = x * y / z_2 * fx;          Jx[1] = -(1 + (x * x / z_2)) * fx;  Jx[2] = y / z * fx;
        Jx[3] = -1. / z * fx;              Jx[4] = 0;                          Jx[5] = x / z_2 * fx;
        Jx[6] = (1 + y * y / z_2) * fy;    Jx[7] = -x * y / z_2 * fy;          Jx[8] = -x / z * fy;
        Jx[9] = 0;                         Jx[10] = -1. / z * fy;              Jx[11] = y / z_2 * fy;
    } else {                         // g2o_types.h:72-84 (== cvutils::JacobXYZ2Cam), columns [trans, rot]
        const double z_inv = 1. / z, z_inv_2 = z_inv * z_inv;
        Jx[0] = -z_inv;  Jx[1] = 0.0;     Jx[2] = x * z_inv_2;  Jx[3] = y * Jx[2];
        Jx[4] = -(1.0 + x * Jx[2]);       Jx[5] = y * z_inv;
        Jx[6] = 0.0;     Jx[7] = -z_inv;  Jx[8] = y * z_inv_2;  Jx[9] = 1.0 + y * Jx[8];
        Jx[10] = -Jx[3]; Jx[11] = -x * z_inv;
    }
}

// camera-frame point and residual of one edge (computeError)
__device__ __forceinline__ void ba_edge_residual(const BaDev &B, int e, const double pt[3], double p[3], double r[2])
{
    const double *pd = B.posed + BA_POSED * (size_t)B.edge_pose[e];
    const double *R = pd + 7;
    if (B.formulation == 2) {
        for (int i = 0; i < 3; ++i) p[i] = R[3 * i] * pt[0] + R[3 * i + 1] * pt[1] + R[3 * i + 2] * pt[2];
    } else {
        const double q[4] = { pd[0], pd[1], pd[2], pd[3] };
        quat_rotate_d(q, pt, p);
    }
    p[0] += pd[4]; p[1] += pd[5]; p[2] += pd[6];
    if (B.formulation == 0) {
        const double proj0 = p[0] / p[2], proj1 = p[1] / p[2];             // camProject, G2oTypes.h:134-144
        r[0] = B.obs[2 * (size_t)e] - (proj0 * B.fx + B.cx);
        r[1] = B.obs[2 * (size_t)e + 1] - (proj1 * B.fy + B.cy);
    } else {
        r[0] = B.obs[2 * (size_t)e] - p[0] / p[2];
        r[1] = B.obs[2 * (size_t)e + 1] - p[1] / p[2];
    }
}

// computeActiveErrors for the edges of map point il: the point's share of the robustified chi2, nothing is written
__device__ __forceinline__ double ba_point_chi2(const BaDev &B, int il)
{
    const double pt[3] = { B.points[3 * (size_t)il], B.points[3 * (size_t)il + 1], B.points[3 * (size_t)il + 2] };
    double sum = 0.0;
    for (int c = B.pt_off[il]; c < B.pt_off[il + 1]; ++c) {
        const int e = B.pt_edges[c];
        if (!B.edge_enable[e]) continue;
        double p[3], r[2];
        ba_edge_residual(B, e, pt, p, r);
        const double e2 = r[0] * r[0] + r[1] * r[1], hub = B.edge_huber[e], dsqr = hub * hub;
        sum += (hub > 0 && e2 > dsqr) ? 2 * sqrt(e2) * hub - dsqr : e2;
    }
    return sum;
}

// all edges of map point il, in edge order: residual, robust weight, Hll / bl (registers), the 6x3 Hpl block per edge,
// (p_cam, rho', r) for the pose pass.  Returns the point's share of the robustified chi2.
__device__ __forceinline__ double ba_point_edges(const BaDev &B, int il)
{
    double chi_sum = 0.0;
    const double pt[3] = { B.points[3 * (size_t)il], B.points[3 * (size_t)il + 1], B.points[3 * (size_t)il + 2] };
    double hl[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 }, gl[3] = { 0, 0, 0 };
    const bool lfree = B.point_fixed[il] == 0;
    for (int c = B.pt_off[il]; c < B.pt_off[il + 1]; ++c) {
        const int e = B.pt_edges[c];
        const int ip = B.edge_pose[e];
        const double *pd = B.posed + BA_POSED * (size_t)ip;
        const double *R = pd + 7;
        double p[3];
        if (B.formulation == 2) {
            for (int i = 0; i < 3; ++i) p[i] = R[3 * i] * pt[0] + R[3 * i + 1] * pt[1] + R[3 * i + 2] * pt[2];
        } else {
            const double q[4] = { pd[0], pd[1], pd[2], pd[3] };
            quat_rotate_d(q, pt, p);
        }
        p[0] += pd[4]; p[1] += pd[5]; p[2] += pd[6];
        const double x = p[0], y = p[1], z = p[2];
        double r[2], Jp[6];
        if (B.formulation == 0) {
            const double proj0 = x / z, proj1 = y / z;                       // camProject, G2oTypes.h:134-144
            r[0] = B.obs[2 * (size_t)e] - (proj0 * B.fx + B.cx);
            r[1] = B.obs[2 * (size_t)e + 1] - (proj1 * B.fy + B.cy);
            const double tmp[6] = { B.fx, 0, -x / z * B.fx, 0, B.fy, -y / z * B.fy };
            double s[6];
            for (int i = 0; i < 6; ++i) s[i] = -1. / z * tmp[i];
            for (int a = 0; a < 2; ++a) for (int b = 0; b < 3; ++b)
                Jp[3 * a + b] = s[3 * a] * R[b] + s[3 * a + 1] * R[3 + b] + s[3 * a + 2] * R[6 + b];
        } else {                                                             // formulations 1 and 2 share residual and point Jacobian
            r[0] = B.obs[2 * (size_t)e] - x / z;
            r[1] = B.obs[2 * (size_t)e + 1] - y / z;
            const double z_inv = 1. / z, z_inv_2 = z_inv * z_inv;
            const double tmp[6] = { z_inv, 0, -x * z_inv_2, 0, z_inv, -y * z_inv_2 };
            for (int a = 0; a < 2; ++a) for (int b = 0; b < 3; ++b)
                Jp[3 * a + b] = -tmp[3 * a] * R[b] + -tmp[3 * a + 1] * R[3 + b] + -tmp[3 * a + 2] * R[6 + b];
        }
        double *et = B.edge_tmp + 6 * (size_t)e;
        double *hpl = B.Hpl + 18 * (size_t)e;
        if (!B.edge_enable[e]) {                                             // SetEnable(false): residual and Jacobians are zero
            B.err[2 * (size_t)e] = 0.0; B.err[2 * (size_t)e + 1] = 0.0; B.chi2_edge[e] = 0.0; B.rho0[e] = 0.0;
            et[0] = x; et[1] = y; et[2] = z; et[3] = 0.0; et[4] = 0.0; et[5] = 0.0;
            for (int i = 0; i < 18; ++i) hpl[i] = 0.0;
            continue;
        }
        if (z < 0) atomicAdd(B.n_behind, 1);
        const double e2 = r[0] * r[0] + r[1] * r[1];
        double rho0 = e2, rho1 = 1.0;
        const double hub = B.edge_huber[e], dsqr = hub * hub;
        if (hub > 0 && e2 > dsqr) {                                          // RobustKernelHuber::robustify == ceres::HuberLoss + Corrector
            const double sqrte = sqrt(e2);
            rho0 = 2 * sqrte * hub - dsqr;
            rho1 = hub / sqrte;
        }
        B.err[2 * (size_t)e] = r[0]; B.err[2 * (size_t)e + 1] = r[1];
        B.chi2_edge[e] = e2; B.rho0[e] = rho0; chi_sum += rho0;
        et[0] = x; et[1] = y; et[2] = z; et[3] = rho1; et[4] = r[0]; et[5] = r[1];
        if (!lfree) { for (int i = 0; i < 18; ++i) hpl[i] = 0.0; continue; }   // constant point: no point block, no cross block
        for (int a = 0; a < 3; ++a) {
            for (int b = 0; b < 3; ++b) hl[3 * a + b] += rho1 * (Jp[a] * Jp[b] + Jp[3 + a] * Jp[3 + b]);
            gl[a] += -rho1 * (Jp[a] * r[0] + Jp[3 + a] * r[1]);
        }
        if (B.fixed[ip]) { for (int i = 0; i < 18; ++i) hpl[i] = 0.0; }
        else {
            double Jx[12];
            ba_pose_jac(B.formulation, x, y, z, B.fx, B.fy, pd, Jx);
            for (int a = 0; a < 6; ++a) for (int b = 0; b < 3; ++b)
                hpl[3 * a + b] = rho1 * (Jx[a] * Jp[b] + Jx[6 + a] * Jp[3 + b]);
        }
    }
    for (int i = 0; i < 9; ++i) B.Hll[9 * (size_t)il + i] = hl[i];
    for (int i = 0; i < 3; ++i) B.bl[3 * (size_t)il + i] = gl[i];
    return chi_sum;
}


#endif
