// B1-B5 -- local-BA reprojection residual / Jacobian / JtJ block build.
// Replaces, per LM iteration of ba::LocalBAG2O (src/Algorithm/BA.cpp:386-543, optimize() at :501-502),
// EdgeSophusSE3ProjectXYZ::computeError + linearizeOplus (include/ygz/G2oTypes.h:84-132) and g2o's
// BaseBinaryEdge::constructQuadraticForm with RobustKernelHuber for every edge; formulation 1 is the
// legacy normalised-plane edge (include/ygz/g2o_types.h:33-86) that src/optimizer.cpp builds on; formulation 2 is
// the ceres side (B3): the functors of include/ygz/Ceres/CeresReprojectionError*.h with pose = [t; angle-axis] and an
// additive update -- the Jacobian AutoDiffCostFunction hands to the solver is reproduced in closed form:
// d r / d t = -A, d r / d aa = A [R p]x J_l(aa), d r / d p_w = -A R, A = d(x/z, y/z)/d p_c.
//
// FP64 VALU work; the block contraction has inner dimension 2 (two residual rows), far too thin
// for MFMA.  The accumulation is organised so that no floating-point atomics are needed and every
// sum has a fixed order:
//  k_ba_pose_prep  lane = pose: SE3::exp once per pose (q, t, R) instead of once per edge.
//  k_ba_points     lane = map point: walks the point's edges in edge order (CSR built at upload),
//                  accumulates Hll / bl in registers exactly in the oracle's order, writes the
//                  unique 6x3 Hpl block, the residual and chi2 of each edge once (coalesced by edge
//                  when edges are sorted by point, as BA.cpp:421-493 generates them), and leaves
//                  (p_cam, rho', r) per edge for the pose pass.
//  k_ba_poses      workgroup = keyframe pose: lanes stride over the pose's edge list, rebuild the
//                  2x6 pose Jacobian from the stored camera point (12 values, ~20 flops: cheaper than
//                  reading it back), accumulate 21+6 sums, fixed-order tree reduction.
//  k_ba_chi2       fixed-order sum of the robustified chi2.
#include "ygz_internal.h"
#include "se3_dev.h"
#include <vector>
#include <string.h>

struct ygz_hip_ctx::BaWindow {
    int K = 0, P = 0, E = 0, formulation = 0;
    double fx = 0, fy = 0, cx = 0, cy = 0, huber = 0;
    void *blob = nullptr;            // one allocation
    double *poses, *points, *obs, *posed, *edge_tmp, *rho0, *edge_huber;
    double *Hpp, *bp, *Hll, *bl, *Hpl, *err, *chi2_edge, *chi2;
    int32_t *edge_pose, *edge_point, *pt_off, *pt_edges, *pose_off, *pose_edges, *n_behind;
    uint8_t *fixed, *point_fixed, *edge_enable;
};
#define BA_POSED 32      // doubles per prepared pose: q(4) t(3) R(9) J_l(9)

struct BaDev {
    int K, P, E, formulation;
    double fx, fy, cx, cy, huber;
    const double *poses, *points, *obs; double *posed, *edge_tmp, *rho0; const double *edge_huber;
    double *Hpp, *bp, *Hll, *bl, *Hpl, *err, *chi2_edge, *chi2;
    const int32_t *edge_pose, *edge_point, *pt_off, *pt_edges, *pose_off, *pose_edges; int32_t *n_behind;
    const uint8_t *fixed, *point_fixed, *edge_enable;
};

__global__ __launch_bounds__(64) void k_ba_pose_prep(const BaDev *__restrict__ wins)
{
    const BaDev B = wins[blockIdx.y];
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k == 0) *B.n_behind = 0;
    if (k >= B.K) return;
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

__global__ __launch_bounds__(128) void k_ba_points(const BaDev *__restrict__ wins)
{
    const BaDev B = wins[blockIdx.y];
    const int il = blockIdx.x * 128 + threadIdx.x;
    if (il >= B.P) return;
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
        B.chi2_edge[e] = e2; B.rho0[e] = rho0;
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
}

__global__ __launch_bounds__(256) void k_ba_poses(const BaDev *__restrict__ wins)
{
    __shared__ double red[4][27];
    const BaDev B = wins[blockIdx.y];
    if ((int)blockIdx.x >= B.K) return;                // block-uniform
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double acc[27];
#pragma unroll
    for (int i = 0; i < 27; ++i) acc[i] = 0.0;
    const bool fixed = B.fixed[k] != 0;
    const double *pd = B.posed + BA_POSED * (size_t)k;
    if (!fixed) {
        for (int c = B.pose_off[k] + tid; c < B.pose_off[k + 1]; c += 256) {
            const int e = B.pose_edges[c];
            const double *et = B.edge_tmp + 6 * (size_t)e;
            const double rho1 = et[3], r0 = et[4], r1 = et[5];
            double Jx[12];
            ba_pose_jac(B.formulation, et[0], et[1], et[2], B.fx, B.fy, pd, Jx);
            int q = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
#pragma unroll
                for (int b = a; b < 6; ++b) acc[q++] += rho1 * (Jx[a] * Jx[b] + Jx[6 + a] * Jx[6 + b]);
            }
#pragma unroll
            for (int a = 0; a < 6; ++a) acc[21 + a] += -rho1 * (Jx[a] * r0 + Jx[6 + a] * r1);
        }
    }
#pragma unroll
    for (int i = 0; i < 27; ++i) {
        double v = acc[i];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) red[wv][i] = v;
    }
    __syncthreads();
    if (tid < 27) {
        const double s = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
        if (tid < 21) {
            int a = 0, rem = tid;                      // unpack upper-triangular index
            while (rem >= 6 - a) { rem -= 6 - a; ++a; }
            const int b = a + rem;
            B.Hpp[36 * (size_t)k + 6 * a + b] = s; B.Hpp[36 * (size_t)k + 6 * b + a] = s;
        } else B.bp[6 * (size_t)k + (tid - 21)] = s;
    }
}

__global__ __launch_bounds__(1024) void k_ba_chi2(const BaDev *__restrict__ wins)
{
    __shared__ double red[16];
    const double *rho0 = wins[blockIdx.x].rho0; const int E = wins[blockIdx.x].E; double *out = wins[blockIdx.x].chi2;
    double v = 0.0;
    for (int e = threadIdx.x; e < E; e += 1024) v += rho0[e];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) { double s = 0; for (int i = 0; i < 16; ++i) s += red[i]; *out = s; }
}

static void ba_free(ygz_hip_ctx::BaWindow *w) { if (w) { if (w->blob) (void)hipFree(w->blob); delete w; } }

extern "C" void ygz_hip_ba_free_all(ygz_hip_ctx *ctx)
{
    for (auto *w : ctx->ba) ba_free(w);
    ctx->ba.clear();
    if (ctx->ba_table) { (void)hipFree(ctx->ba_table); ctx->ba_table = nullptr; }
}

static BaDev ba_dev(const ygz_hip_ctx::BaWindow *w)
{
    BaDev B;
    B.K = w->K; B.P = w->P; B.E = w->E; B.formulation = w->formulation;
    B.fx = w->fx; B.fy = w->fy; B.cx = w->cx; B.cy = w->cy; B.huber = w->huber;
    B.poses = w->poses; B.points = w->points; B.obs = w->obs; B.posed = w->posed; B.edge_tmp = w->edge_tmp; B.rho0 = w->rho0;
    B.edge_huber = w->edge_huber; B.n_behind = w->n_behind; B.point_fixed = w->point_fixed; B.edge_enable = w->edge_enable;
    B.Hpp = w->Hpp; B.bp = w->bp; B.Hll = w->Hll; B.bl = w->bl; B.Hpl = w->Hpl; B.err = w->err; B.chi2_edge = w->chi2_edge; B.chi2 = w->chi2;
    B.edge_pose = w->edge_pose; B.edge_point = w->edge_point; B.pt_off = w->pt_off; B.pt_edges = w->pt_edges;
    B.pose_off = w->pose_off; B.pose_edges = w->pose_edges; B.fixed = w->fixed;
    return B;
}

extern "C" {

int ygz_hip_ba_upload(ygz_hip_ctx *ctx, int window, const ygz_ba_problem *pb)
{
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || !pb || window < 0 || window > 1023) return YGZ_E_INVALID;
    const int K = pb->n_poses, P = pb->n_points, E = pb->n_edges;
    if (K < 1 || P < 1 || E < 0 || !pb->poses || !pb->points || (E > 0 && (!pb->edge_pose || !pb->edge_point || !pb->obs)))
        return YGZ_E_INVALID;
    if (pb->formulation < 0 || pb->formulation > 2) return YGZ_E_INVALID;
    for (int e = 0; e < E; ++e)
        if (pb->edge_pose[e] < 0 || pb->edge_pose[e] >= K || pb->edge_point[e] < 0 || pb->edge_point[e] >= P) return YGZ_E_INVALID;
    if ((int)ctx->ba.size() <= window) ctx->ba.resize(window + 1, nullptr);
    if (ctx->ba[window]) { YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); ba_free(ctx->ba[window]); ctx->ba[window] = nullptr; }
    auto *w = new ygz_hip_ctx::BaWindow();
    w->K = K; w->P = P; w->E = E; w->formulation = pb->formulation;
    w->fx = pb->fx; w->fy = pb->fy; w->cx = pb->cx; w->cy = pb->cy; w->huber = pb->huber_delta;
    // CSR by point and by pose, edges in ascending edge order inside each row
    std::vector<int32_t> pt_off(P + 1, 0), pose_off(K + 1, 0), pt_edges(E > 0 ? E : 1), pose_edges(E > 0 ? E : 1);
    for (int e = 0; e < E; ++e) { pt_off[pb->edge_point[e] + 1]++; pose_off[pb->edge_pose[e] + 1]++; }
    for (int i = 0; i < P; ++i) pt_off[i + 1] += pt_off[i];
    for (int i = 0; i < K; ++i) pose_off[i + 1] += pose_off[i];
    { std::vector<int32_t> c1(pt_off.begin(), pt_off.end() - 1), c2(pose_off.begin(), pose_off.end() - 1);
      for (int e = 0; e < E; ++e) { pt_edges[c1[pb->edge_point[e]]++] = e; pose_edges[c2[pb->edge_pose[e]]++] = e; } }
    // one blob: doubles first, then int32, then bytes
    const size_t Ez = (size_t)(E > 0 ? E : 1);
    const size_t nd = (size_t)K * 6 + (size_t)P * 3 + Ez * 2 + (size_t)K * BA_POSED + Ez * 6 + Ez + Ez
                    + (size_t)K * 36 + (size_t)K * 6 + (size_t)P * 9 + (size_t)P * 3 + Ez * 18 + Ez * 2 + Ez + 1;
    const size_t ni = Ez * 4 + (size_t)P + 1 + (size_t)K + 1 + 1;
    const size_t bytes = nd * 8 + ni * 4 + (size_t)K + (size_t)P + Ez + 64;
    hipError_t he = hipMalloc(&w->blob, bytes);
    if (he != hipSuccess) { ctx->last_hip_error = (int)he; delete w; return YGZ_E_HIP; }
    double *d = (double *)w->blob;
    w->poses = d; d += (size_t)K * 6; w->points = d; d += (size_t)P * 3; w->obs = d; d += Ez * 2;
    w->posed = d; d += (size_t)K * BA_POSED; w->edge_tmp = d; d += Ez * 6; w->rho0 = d; d += Ez; w->edge_huber = d; d += Ez;
    w->Hpp = d; d += (size_t)K * 36; w->bp = d; d += (size_t)K * 6; w->Hll = d; d += (size_t)P * 9; w->bl = d; d += (size_t)P * 3;
    w->Hpl = d; d += Ez * 18; w->err = d; d += Ez * 2; w->chi2_edge = d; d += Ez; w->chi2 = d; d += 1;
    int32_t *ii = (int32_t *)d;
    w->edge_pose = ii; ii += Ez; w->edge_point = ii; ii += Ez; w->pt_edges = ii; ii += Ez; w->pose_edges = ii; ii += Ez;
    w->pt_off = ii; ii += (size_t)P + 1; w->pose_off = ii; ii += (size_t)K + 1; w->n_behind = ii; ii += 1;
    w->fixed = (uint8_t *)ii; w->point_fixed = w->fixed + K; w->edge_enable = w->point_fixed + P;
    ctx->ba[window] = w;
    ctx->ba_table_dirty = true;
    std::vector<uint8_t> fixed(K, 0), pfixed(P, 0), enable(Ez, 1);
    std::vector<double> hub(Ez, pb->huber_delta);
    if (pb->pose_fixed) memcpy(fixed.data(), pb->pose_fixed, K);
    if (pb->point_fixed) memcpy(pfixed.data(), pb->point_fixed, P);
    if (pb->edge_enable && E > 0) memcpy(enable.data(), pb->edge_enable, E);
    if (pb->edge_huber && E > 0) memcpy(hub.data(), pb->edge_huber, (size_t)E * 8);
    YGZ_HIPCHK(ctx, hipMemcpyAsync(w->poses, pb->poses, (size_t)K * 48, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(w->points, pb->points, (size_t)P * 24, hipMemcpyHostToDevice, ctx->stream));
    if (E > 0) {
        YGZ_HIPCHK(ctx, hipMemcpyAsync(w->obs, pb->obs, (size_t)E * 16, hipMemcpyHostToDevice, ctx->stream));
        YGZ_HIPCHK(ctx, hipMemcpyAsync(w->edge_pose, pb->edge_pose, (size_t)E * 4, hipMemcpyHostToDevice, ctx->stream));
        YGZ_HIPCHK(ctx, hipMemcpyAsync(w->edge_point, pb->edge_point, (size_t)E * 4, hipMemcpyHostToDevice, ctx->stream));
        YGZ_HIPCHK(ctx, hipMemcpyAsync(w->pt_edges, pt_edges.data(), (size_t)E * 4, hipMemcpyHostToDevice, ctx->stream));
        YGZ_HIPCHK(ctx, hipMemcpyAsync(w->pose_edges, pose_edges.data(), (size_t)E * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    YGZ_HIPCHK(ctx, hipMemcpyAsync(w->pt_off, pt_off.data(), ((size_t)P + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(w->pose_off, pose_off.data(), ((size_t)K + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(w->fixed, fixed.data(), (size_t)K, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(w->point_fixed, pfixed.data(), (size_t)P, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(w->edge_enable, enable.data(), Ez, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(w->edge_huber, hub.data(), Ez * 8, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));     // host vectors go out of scope
    return YGZ_OK;
}

int ygz_hip_ba_set_state(ygz_hip_ctx *ctx, int window, const double *poses, const double *points)
{
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || window < 0 || window >= (int)ctx->ba.size() || !ctx->ba[window]) return YGZ_E_INVALID;
    auto *w = ctx->ba[window];
    if (poses) YGZ_HIPCHK(ctx, hipMemcpyAsync(w->poses, poses, (size_t)w->K * 48, hipMemcpyHostToDevice, ctx->stream));
    if (points) YGZ_HIPCHK(ctx, hipMemcpyAsync(w->points, points, (size_t)w->P * 24, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return YGZ_OK;
}

// device-pointer variant: the new state is already in HBM (e.g. a torch tensor filled by an RCCL broadcast)
int ygz_hip_ba_set_state_device(ygz_hip_ctx *ctx, int window, const double *d_poses, const double *d_points)
{
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || window < 0 || window >= (int)ctx->ba.size() || !ctx->ba[window]) return YGZ_E_INVALID;
    auto *w = ctx->ba[window];
    if (d_poses) YGZ_HIPCHK(ctx, hipMemcpyAsync(w->poses, d_poses, (size_t)w->K * 48, hipMemcpyDeviceToDevice, ctx->stream));
    if (d_points) YGZ_HIPCHK(ctx, hipMemcpyAsync(w->points, d_points, (size_t)w->P * 24, hipMemcpyDeviceToDevice, ctx->stream));
    return YGZ_OK;
}

int ygz_hip_ba_linearize_resident(ygz_hip_ctx *ctx, int window_begin, int n_windows)
{
    if (!ctx || window_begin < 0 || n_windows < 1 || window_begin + n_windows > (int)ctx->ba.size()) return YGZ_E_INVALID;
    for (int i = window_begin; i < window_begin + n_windows; ++i) if (!ctx->ba[i]) return YGZ_E_INVALID;
    if (ctx->ba_table_dirty) { int rj = ygz_join(ctx); if (rj != YGZ_OK) return rj; }
    if (ctx->ba_table_dirty) {                       // descriptor table of all windows (changes only at upload time)
        std::vector<BaDev> tab(1024);
        memset(tab.data(), 0, tab.size() * sizeof(BaDev));
        ctx->ba_max_K = ctx->ba_max_P = 0;
        for (size_t i = 0; i < ctx->ba.size() && i < 1024; ++i)
            if (ctx->ba[i]) {
                tab[i] = ba_dev(ctx->ba[i]);
                if (ctx->ba[i]->K > ctx->ba_max_K) ctx->ba_max_K = ctx->ba[i]->K;
                if (ctx->ba[i]->P > ctx->ba_max_P) ctx->ba_max_P = ctx->ba[i]->P;
            }
        if (!ctx->ba_table) YGZ_HIPCHK(ctx, hipMalloc(&ctx->ba_table, 1024 * sizeof(BaDev)));
        YGZ_HIPCHK(ctx, hipMemcpyAsync(ctx->ba_table, tab.data(), tab.size() * sizeof(BaDev), hipMemcpyHostToDevice, ctx->stream));
        YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        ctx->ba_table_dirty = false;
    }
    const BaDev *tab = reinterpret_cast<const BaDev *>(ctx->ba_table) + window_begin;
    YgzAuxScope aux(ctx, 1);
    YGZ_LAUNCH(ctx, KID_BA_POSE_PREP, k_ba_pose_prep, dim3(ygz_div_up(ctx->ba_max_K, 64), n_windows), dim3(64), tab);
    YGZ_LAUNCH(ctx, KID_BA_POINTS, k_ba_points, dim3(ygz_div_up(ctx->ba_max_P, 128), n_windows), dim3(128), tab);
    YGZ_LAUNCH(ctx, KID_BA_POSES, k_ba_poses, dim3(ctx->ba_max_K, n_windows), dim3(256), tab);
    YGZ_LAUNCH(ctx, KID_BA_CHI2, k_ba_chi2, dim3(n_windows), dim3(1024), tab);
    YGZ_HIPCHK(ctx, hipGetLastError());
    return YGZ_OK;
}

int ygz_hip_ba_download(ygz_hip_ctx *ctx, int window, double *Hpp, double *bp, double *Hll, double *bl, double *Hpl,
                        double *err, double *chi2_edge, double *chi2)
{
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || window < 0 || window >= (int)ctx->ba.size() || !ctx->ba[window]) return YGZ_E_INVALID;
    auto *w = ctx->ba[window];
    const size_t K = w->K, P = w->P, E = w->E;
#define DL_(dst, src, n) if ((dst) && (n) > 0) YGZ_HIPCHK(ctx, hipMemcpyAsync((dst), (src), (n) * 8, hipMemcpyDeviceToHost, ctx->stream))
    DL_(Hpp, w->Hpp, K * 36); DL_(bp, w->bp, K * 6); DL_(Hll, w->Hll, P * 9); DL_(bl, w->bl, P * 3);
    DL_(Hpl, w->Hpl, E * 18); DL_(err, w->err, E * 2); DL_(chi2_edge, w->chi2_edge, E); DL_(chi2, w->chi2, (size_t)1);
#undef DL_
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return YGZ_OK;
}

int ygz_hip_ba_behind_camera(ygz_hip_ctx *ctx, int window, int *n_behind)
{
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || !n_behind || window < 0 || window >= (int)ctx->ba.size() || !ctx->ba[window]) return YGZ_E_INVALID;
    int32_t v = 0;
    YGZ_HIPCHK(ctx, hipMemcpyAsync(&v, ctx->ba[window]->n_behind, 4, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *n_behind = v;
    return YGZ_OK;
}

int ygz_hip_ba_set_enable(ygz_hip_ctx *ctx, int window, const uint8_t *edge_enable)
{
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || !edge_enable || window < 0 || window >= (int)ctx->ba.size() || !ctx->ba[window]) return YGZ_E_INVALID;
    auto *w = ctx->ba[window];
    if (w->E > 0) YGZ_HIPCHK(ctx, hipMemcpyAsync(w->edge_enable, edge_enable, (size_t)w->E, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return YGZ_OK;
}

int ygz_hip_ba_linearize(ygz_hip_ctx *ctx, const ygz_ba_problem *pb, double *Hpp, double *bp, double *Hll, double *bl,
                         double *Hpl, double *err, double *chi2_edge, double *chi2)
{
    if (!ctx) return YGZ_E_INVALID;
    const int window = 1023;          // private slot for the one-shot form
    int rc = ygz_hip_ba_upload(ctx, window, pb);
    if (rc == YGZ_OK) rc = ygz_hip_ba_linearize_resident(ctx, window, 1);
    if (rc == YGZ_OK) rc = ygz_hip_ba_download(ctx, window, Hpp, bp, Hll, bl, Hpl, err, chi2_edge, chi2);
    return rc;
}

}  // extern "C"
