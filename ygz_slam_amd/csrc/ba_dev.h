// Device-side building blocks of the local-BA edge stack, shared by the linearisation kernels (ba.hip) and the resident
// Levenberg-Marquardt kernel (ba_resident_lm.hip).  See ba.hip for the reference citations.
//
// HBM layout of a window.  The unit of parallelism is the map point (lane = point), so every per-edge array is stored the way
// the lanes touch it: points are taken 64 at a time (a chunk = a wavefront), the c-th edge of every point of a chunk forms
// a ROW, and a row keeps each of its NC components as 64 consecutive doubles:
//     element(l, c, k) = base[((slot_off[l >> 6] + c) * NC + k) * 64 + (l & 63)]
// A store of component k by the 64 lanes is one contiguous 512-byte write and a load one contiguous read -- the edge-major
// AoS form ([E][18] doubles written 8 bytes at a time with a ~1.3 KB lane stride) cost 1.6x write amplification plus a
// read-for-ownership of the same size in HBM.  Per-point arrays (Hll, bl) use the same idea with c = 0: [chunk][NC][64].
// The ABI keeps edge order: ygz_hip_ba_download gathers through the (row, lane) of every edge.
#ifndef YGZ_BA_DEV_H_
#define YGZ_BA_DEV_H_
#include "ygz_internal.h"
#include "se3_dev.h"
#include <vector>

struct ygz_hip_ctx::BaWindow {
    int K = 0, P = 0, E = 0, formulation = 0, Kf = 0, R = 0, Q = 0;       // R rows, Q = ceil(P / 64) chunks
    double fx = 0, fy = 0, cx = 0, cy = 0, huber = 0;
    void *blob = nullptr;            // one allocation
    size_t blob_bytes = 0;           // its size (an upload into the same slot reuses it when it is large enough)
    double *poses, *points, *posed;
    double *obs_c, *huber_c;         // [R][2][64], [R][64]
    int32_t *pose_c;                 // [R][64] pose of the c-th edge of the lane's point, -1: no such edge
    uint8_t *enable_c;               // [R][64]
    int32_t *slot_off;               // [Q + 1] first row of each chunk
    int16_t *ppc;                    // [P][Kf] c of the (first) edge from point l to free pose a, or -1
    int16_t *dupn;                   // [R][64] c of the NEXT edge of the lane's point to the same pose, or -1 (two features of one frame
                                     // observing one map point: ceres / g2o count both residual blocks)
    bool has_dup = false;
    int32_t *edge_rl;                // [E] row * 64 + lane of edge e (gather / scatter between ABI order and rows)
    uint8_t *fixed, *point_fixed;
    int32_t *free_idx, *free_pose, *n_behind;
    double *Hpp, *bp, *chi2;         // [K][36], [K][6], [1]
    double *Hll_c, *bl_c;            // [Q][9][64], [Q][3][64]
    double *Hpl_c, *err_c, *chi2e_c; // [R][18][64], [R][2][64], [R][64]
    double *part_pose, *part_chi;    // [Q][Kf][27], [Q]: per-wavefront partial sums of the pose blocks / chi2
    // work space of the resident Levenberg-Marquardt kernel (ba_resident_lm.hip)
    double *poses_bk, *points_bk, *Y_c, *Dinv, *xl;       // Y_c [R][18][64]
    double *sc_p, *sc_l;             // [K][6], [P][3] Jacobi column scales of the resident trust-region loop
    double *lm_out;                  // [16]: ygz_ba_stats of the last resident LM run on this window (iterations < 0: none, or a team member timed out)
    // host side: where each edge lives (for ygz_hip_ba_set_enable)
    std::vector<int32_t> h_edge_rl;
    // windows whose graph is built on the device (window.hip): the blob is carved for the capacities below, the kernels read the actual
    // K / Kf / P / E / R / Q from the DEVICE table entry, which k_win_edges patches; the host fields above keep the capacities
    bool device_built = false;
    int cap_K = 0, cap_P = 0;
    bool table_dirty = true;         // the device table entry must be re-uploaded from the host fields
};
#define BA_POSED 32      // doubles per prepared pose: q(4) t(3) R(9) J_l(9)

struct BaDev {
    int K, P, E, formulation, Kf, R, Q;
    double fx, fy, cx, cy, huber;
    const double *poses, *points; double *posed;
    const double *obs_c, *huber_c; const int32_t *pose_c; const uint8_t *enable_c; const int32_t *slot_off; const int16_t *ppc, *dupn;
    const int32_t *edge_rl;
    const uint8_t *fixed, *point_fixed; const int32_t *free_idx, *free_pose; int32_t *n_behind;
    double *Hpp, *bp, *chi2, *Hll_c, *bl_c, *Hpl_c, *err_c, *chi2e_c, *part_pose, *part_chi;
    double *poses_w, *points_w;      // the same state arrays, writable (LM update / restore)
    double *poses_bk, *points_bk, *Y_c, *Dinv, *xl;
    double *sc_p, *sc_l;
    double *lm_out;
};
const BaDev *ygz_ba_table(ygz_hip_ctx *ctx, int *rc);        // device table of all uploaded windows, rebuilt when dirty

// element k of row `row` for lane `lane` in a chunked per-edge array with NC components
#define BA_EC(base, row, NC, k, lane) ((base)[(((size_t)(row) * (NC)) + (k)) * 64 + (lane)])
// element k of point l in a chunked per-point array with NC components
// the per-edge outputs are written once and read by a later kernel: streaming stores
#ifdef YGZ_BA_PLAIN_STORES
#define BA_ST(p, v) (*(p) = (v))
#define BA_LD(p) (*(p))
#else
#define BA_LD(p) (*(p))      /* the inputs are read again by the pose pass: streaming loads measured 15 % slower */
#define BA_ST(p, v) __builtin_nontemporal_store((v), (p))
#endif
#define BA_PC(base, l, NC, k) ((base)[((size_t)((l) >> 6) * (NC) + (k)) * 64 + ((l) & 63)])

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

// Divisions (round 5): an edge divides by its camera depth ONCE (zi = 1 / z, IEEE); the quotients x / z, y / z, 1 / z, . / z^2 of the reference's
// expressions (G2oTypes.h:119-144) are products with zi.  A quotient differs from the reference's by at most one more rounding (1e-16 relative;
// the path's bar is 1e-5, the tests hold 1e-9) -- an FP64 division is ~28 instructions at half rate, and the two passes over an edge
// (point blocks, pose blocks) held sixteen of them.
// pd = the prepared pose (q, t, R, J_l), read only by formulation 2
__device__ __forceinline__ void ba_pose_jac(int formulation, double x, double y, double z, double zi, double fx, double fy,
                                            const double *__restrict__ pd, double Jx[12])
{
    if (formulation == 2) {          // d r / d [t; aa] of the ceres functor: [-A, A [R p]x J_l]
        const double xz = x * zi * zi, yz = y * zi * zi;
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
        const double zi2 = zi * zi, xz = x * zi, yz = y * zi;
        Jx[0] = x * y * zi2 * fx;          Jx[1] = -(1 + (x * x * zi2)) * fx;  Jx[2] = yz * fx;
        Jx[3] = -zi * fx;                  Jx[4] = 0;                          Jx[5] = x * zi2 * fx;
        Jx[6] = (1 + y * y * zi2) * fy;    Jx[7] = -x * y * zi2 * fy;          Jx[8] = -xz * fy;
        Jx[9] = 0;                         Jx[10] = -zi * fy;                  Jx[11] = y * zi2 * fy;
    } else {                         // g2o_types.h:72-84 (== cvutils::JacobXYZ2Cam), columns [trans, rot]
        const double z_inv = zi, z_inv_2 = z_inv * z_inv;
        Jx[0] = -z_inv;  Jx[1] = 0.0;     Jx[2] = x * z_inv_2;  Jx[3] = y * Jx[2];
        Jx[4] = -(1.0 + x * Jx[2]);       Jx[5] = y * z_inv;
        Jx[6] = 0.0;     Jx[7] = -z_inv;  Jx[8] = y * z_inv_2;  Jx[9] = 1.0 + y * Jx[8];
        Jx[10] = -Jx[3]; Jx[11] = -x * z_inv;
    }
}


// camera-frame point of map point pt seen from prepared pose pd, the reciprocal of its depth, and the residual against obs (computeError)
__device__ __forceinline__ double ba_project_zi(const BaDev &B, const double *__restrict__ pd, const double pt[3], double ox, double oy,
                                                double p[3], double r[2])
{
    const double *R = pd + 7;
    if (B.formulation == 2) {
        for (int i = 0; i < 3; ++i) p[i] = R[3 * i] * pt[0] + R[3 * i + 1] * pt[1] + R[3 * i + 2] * pt[2];
    } else {
        const double q[4] = { pd[0], pd[1], pd[2], pd[3] };
        quat_rotate_d(q, pt, p);
    }
    p[0] += pd[4]; p[1] += pd[5]; p[2] += pd[6];
    const double zi = 1.0 / p[2];
    if (B.formulation == 0) {
        const double proj0 = p[0] * zi, proj1 = p[1] * zi;                 // camProject, G2oTypes.h:134-144
        r[0] = ox - (proj0 * B.fx + B.cx);
        r[1] = oy - (proj1 * B.fy + B.cy);
    } else {                                                               // formulations 1 and 2 share the residual
        r[0] = ox - p[0] * zi;
        r[1] = oy - p[1] * zi;
    }
    return zi;
}
__device__ __forceinline__ void ba_project(const BaDev &B, const double *__restrict__ pd, const double pt[3], double ox, double oy,
                                           double p[3], double r[2])
{ (void)ba_project_zi(B, pd, pt, ox, oy, p, r); }

// RobustKernelHuber::robustify == ceres::HuberLoss + Corrector
__device__ __forceinline__ void ba_robust(double e2, double hub, double *rho0, double *rho1)
{
    *rho0 = e2; *rho1 = 1.0;
    const double dsqr = hub * hub;
    if (hub > 0 && e2 > dsqr) { const double sqrte = sqrt(e2); *rho0 = 2 * sqrte * hub - dsqr; *rho1 = hub / sqrte; }
}

// computeActiveErrors for the edges of map point il: the point's share of the robustified chi2, nothing is written
__device__ __forceinline__ double ba_point_chi2(const BaDev &B, int il)
{
    const int lane = il & 63, row0 = B.slot_off[il >> 6], rows = B.slot_off[(il >> 6) + 1] - row0;
    const double pt[3] = { B.points[3 * (size_t)il], B.points[3 * (size_t)il + 1], B.points[3 * (size_t)il + 2] };
    double sum = 0.0;
    for (int c = 0; c < rows; ++c) {
        const int row = row0 + c, ip = B.pose_c[(size_t)row * 64 + lane];
        if (ip < 0 || !B.enable_c[(size_t)row * 64 + lane]) continue;
        double p[3], r[2], rho0, rho1;
        ba_project(B, B.posed + BA_POSED * (size_t)ip, pt, BA_EC(B.obs_c, row, 2, 0, lane), BA_EC(B.obs_c, row, 2, 1, lane), p, r);
        ba_robust(r[0] * r[0] + r[1] * r[1], B.huber_c[(size_t)row * 64 + lane], &rho0, &rho1);
        sum += rho0;
    }
    return sum;
}

// all edges of map point il, in edge order: residual, robust weight, Hll / bl (registers), the 6x3 Hpl block per edge.
// Returns the point's share of the robustified chi2.
__device__ __forceinline__ double ba_point_edges(const BaDev &B, int il)
{
    const int lane = il & 63, row0 = B.slot_off[il >> 6], rows = B.slot_off[(il >> 6) + 1] - row0;
    double chi_sum = 0.0;
    const double pt[3] = { B.points[3 * (size_t)il], B.points[3 * (size_t)il + 1], B.points[3 * (size_t)il + 2] };
    double hl[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 }, gl[3] = { 0, 0, 0 };
    const bool lfree = B.point_fixed[il] == 0;
    // the inputs of row c + 1 are in flight while row c is computed (the loop is a chain of dependent loads otherwise)
    int n_ip = -1, n_en = 0; double n_ox = 0, n_oy = 0, n_hub = 0;
#define BA_FETCH_(row_)                                                                                          \
    { n_ip = BA_LD(&B.pose_c[(size_t)(row_) * 64 + lane]); n_en = BA_LD(&B.enable_c[(size_t)(row_) * 64 + lane]);  \
      n_ox = BA_LD(&BA_EC(B.obs_c, row_, 2, 0, lane)); n_oy = BA_LD(&BA_EC(B.obs_c, row_, 2, 1, lane)); n_hub = BA_LD(&B.huber_c[(size_t)(row_) * 64 + lane]); }
    if (rows > 0) BA_FETCH_(row0)
    for (int c = 0; c < rows; ++c) {
        const int row = row0 + c, ip = n_ip, en = n_en;
        const double ox = n_ox, oy = n_oy, hub = n_hub;
        if (c + 1 < rows) BA_FETCH_(row + 1)
        if (ip < 0) continue;                                                // this point has fewer edges than the chunk's longest
        const double *pd = B.posed + BA_POSED * (size_t)ip;
        const double *R = pd + 7;
        double p[3], r[2], Jp[6];
        const double zi = ba_project_zi(B, pd, pt, ox, oy, p, r);
        const double x = p[0], y = p[1], z = p[2];
        if (B.formulation == 0) {
            const double tmp[6] = { B.fx, 0, -(x * zi) * B.fx, 0, B.fy, -(y * zi) * B.fy };
            double s[6];
            for (int i = 0; i < 6; ++i) s[i] = -zi * tmp[i];
            for (int a = 0; a < 2; ++a) for (int b = 0; b < 3; ++b)
                Jp[3 * a + b] = s[3 * a] * R[b] + s[3 * a + 1] * R[3 + b] + s[3 * a + 2] * R[6 + b];
        } else {                                                             // formulations 1 and 2 share the point Jacobian
            const double z_inv = zi, z_inv_2 = z_inv * z_inv;
            const double tmp[6] = { z_inv, 0, -x * z_inv_2, 0, z_inv, -y * z_inv_2 };
            for (int a = 0; a < 2; ++a) for (int b = 0; b < 3; ++b)
                Jp[3 * a + b] = -tmp[3 * a] * R[b] + -tmp[3 * a + 1] * R[3 + b] + -tmp[3 * a + 2] * R[6 + b];
        }
        if (!en) {                                                           // SetEnable(false): residual and Jacobians are zero
            BA_EC(B.err_c, row, 2, 0, lane) = 0.0; BA_EC(B.err_c, row, 2, 1, lane) = 0.0; B.chi2e_c[(size_t)row * 64 + lane] = 0.0;
            for (int i = 0; i < 18; ++i) BA_EC(B.Hpl_c, row, 18, i, lane) = 0.0;
            continue;
        }
        if (z < 0) atomicAdd(B.n_behind, 1);
        const double e2 = r[0] * r[0] + r[1] * r[1];
        double rho0, rho1;
        ba_robust(e2, hub, &rho0, &rho1);
        BA_ST(&BA_EC(B.err_c, row, 2, 0, lane), r[0]); BA_ST(&BA_EC(B.err_c, row, 2, 1, lane), r[1]);
        B.chi2e_c[(size_t)row * 64 + lane] = e2; chi_sum += rho0;
        if (!lfree) { for (int i = 0; i < 18; ++i) BA_EC(B.Hpl_c, row, 18, i, lane) = 0.0; continue; }   // constant point: no point / cross block
        for (int a = 0; a < 3; ++a) {
            for (int b = 0; b < 3; ++b) hl[3 * a + b] += rho1 * (Jp[a] * Jp[b] + Jp[3 + a] * Jp[3 + b]);
            gl[a] += -rho1 * (Jp[a] * r[0] + Jp[3 + a] * r[1]);
        }
        if (B.fixed[ip]) { for (int i = 0; i < 18; ++i) BA_EC(B.Hpl_c, row, 18, i, lane) = 0.0; }
        else {
            double Jx[12];
            ba_pose_jac(B.formulation, x, y, z, zi, B.fx, B.fy, pd, Jx);
            for (int a = 0; a < 6; ++a) for (int b = 0; b < 3; ++b)
                BA_ST(&BA_EC(B.Hpl_c, row, 18, 3 * a + b, lane), rho1 * (Jx[a] * Jp[b] + Jx[6 + a] * Jp[3 + b]));
        }
    }
#undef BA_FETCH_
    for (int i = 0; i < 9; ++i) BA_PC(B.Hll_c, il, 9, i) = hl[i];
    for (int i = 0; i < 3; ++i) BA_PC(B.bl_c, il, 3, i) = gl[i];
    return chi_sum;
}

// contribution of map point il to the blocks of free pose a (Hpp upper triangle 21 + bp 6), added to acc; the edge's camera
// point, residual and weight are recomputed exactly as in ba_point_edges (cheaper than storing 48 bytes per edge and reading
// them back through a pose-major gather).
__device__ __forceinline__ void ba_pose_contrib(const BaDev &B, int il, int a, double acc[27])
{
    int c = B.ppc[(size_t)il * B.Kf + a];
    if (c < 0) return;
    const int lane = il & 63, row0 = B.slot_off[il >> 6];
    const int k = B.free_pose[a];
    const double *pd = B.posed + BA_POSED * (size_t)k;
    const double pt[3] = { B.points[3 * (size_t)il], B.points[3 * (size_t)il + 1], B.points[3 * (size_t)il + 2] };
    for (; c >= 0; c = B.dupn[(size_t)(row0 + c) * 64 + lane]) {            // almost always one edge; the chain holds repeated (point, pose) pairs
        const int row = row0 + c;
        if (!B.enable_c[(size_t)row * 64 + lane]) continue;
        double p[3], r[2], rho0, rho1, Jx[12];
        const double zi = ba_project_zi(B, pd, pt, BA_EC(B.obs_c, row, 2, 0, lane), BA_EC(B.obs_c, row, 2, 1, lane), p, r);
        ba_robust(r[0] * r[0] + r[1] * r[1], B.huber_c[(size_t)row * 64 + lane], &rho0, &rho1);
        ba_pose_jac(B.formulation, p[0], p[1], p[2], zi, B.fx, B.fy, pd, Jx);
        int q = 0;
#pragma unroll
        for (int u = 0; u < 6; ++u) {
#pragma unroll
            for (int v = u; v < 6; ++v) acc[q++] += rho1 * (Jx[u] * Jx[v] + Jx[6 + u] * Jx[6 + v]);
        }
#pragma unroll
        for (int u = 0; u < 6; ++u) acc[21 + u] += -rho1 * (Jx[u] * r[0] + Jx[6 + u] * r[1]);
    }
}

// Sums of 27 (padded to 32) per-lane FP64 values over the 64 lanes of a wavefront, all at once: a butterfly in which every
// level halves the number of values a lane still carries -- it keeps one half and hands the other to its partner
// (lane ^ 1, ^ 2, ^ 4, ^ 8, ^ 16), so 16 + 8 + 4 + 2 + 1 exchanges replace 27 separate 6-step reductions (~210 instead of ~620
// instructions; the pose blocks of k_ba_points need 9 such sums per chunk).  On return lane l (and l + 32) holds the total
// of value *idx = bit-reversed low five bits of l.  The order of the additions is fixed.
__device__ __forceinline__ double ba_dpp_d(double v, const int ctrl_is_xor1)
{   // partner's value for lane ^ 1 (quad_perm [1,0,3,2]) or lane ^ 2 (quad_perm [2,3,0,1])
    const int lo = ctrl_is_xor1 ? __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xF, 0xF, false)
                                : __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x4E, 0xF, 0xF, false);
    const int hi = ctrl_is_xor1 ? __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xB1, 0xF, 0xF, false)
                                : __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x4E, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double ba_ror8_d(double v)
{   // lane ^ 8 inside a row of 16: row_ror:8
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x128, 0xF, 0xF, false),
                            __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x128, 0xF, 0xF, false));
}
__device__ __forceinline__ double ba_reduce32(const double (&acc)[27], int lane, int *idx)
{
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8, b4 = lane & 16;
    double v1[16], v2[8], v3[4], v4[2];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double lo = acc[j], hi = (16 + j < 27) ? acc[16 + j] : 0.0;
        const double keep = b0 ? hi : lo, send = b0 ? lo : hi;
        v1[j] = keep + ba_dpp_d(send, 1);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const double keep = b1 ? v1[8 + j] : v1[j], send = b1 ? v1[j] : v1[8 + j];
        v2[j] = keep + ba_dpp_d(send, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double keep = b2 ? v2[4 + j] : v2[j], send = b2 ? v2[j] : v2[4 + j];
        v3[j] = keep + __shfl_xor(send, 4);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const double keep = b3 ? v3[2 + j] : v3[j], send = b3 ? v3[j] : v3[2 + j];
        v4[j] = keep + ba_ror8_d(send);
    }
    const double keep = b4 ? v4[1] : v4[0], send = b4 ? v4[0] : v4[1];
    double v5 = keep + __shfl_xor(send, 16);
    v5 = v5 + __shfl_xor(v5, 32);
    *idx = (b0 ? 16 : 0) + (b1 ? 8 : 0) + (b2 ? 4 : 0) + (b3 ? 2 : 0) + (b4 ? 1 : 0);
    return v5;
}


#endif
