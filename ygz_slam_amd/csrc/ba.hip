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
#include "ba_dev.h"
#include <vector>
#include <string.h>

__global__ __launch_bounds__(64) void k_ba_pose_prep(const BaDev *__restrict__ wins)
{
    const BaDev B = wins[blockIdx.y];
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k == 0) *B.n_behind = 0;
    if (k >= B.K) return;
    ba_pose_prep_one(B, k);
}

__global__ __launch_bounds__(128) void k_ba_points(const BaDev *__restrict__ wins)
{
    const BaDev B = wins[blockIdx.y];
    const int il = blockIdx.x * 128 + threadIdx.x;
    if (il >= B.P) return;
    (void)ba_point_edges(B, il);
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
    B.Kf = w->Kf; B.poses_w = w->poses; B.points_w = w->points; B.poses_bk = w->poses_bk; B.points_bk = w->points_bk;
    B.Y = w->Y; B.Dinv = w->Dinv; B.xl = w->xl; B.free_idx = w->free_idx; B.free_pose = w->free_pose; B.pt_pose_edge = w->pt_pose_edge;
    return B;
}

// descriptor table of all windows (changes only at upload time)
const BaDev *ygz_ba_table(ygz_hip_ctx *ctx, int *rc)
{
#define TAB_CHK_(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { ctx->last_hip_error = (int)e_; *rc = YGZ_E_HIP; return nullptr; } } while (0)
    if (ctx->ba_table_dirty) { const int rj = ygz_join(ctx); if (rj != YGZ_OK) { *rc = rj; return nullptr; } }
    if (ctx->ba_table_dirty) {
        std::vector<BaDev> tab(1024);
        memset(tab.data(), 0, tab.size() * sizeof(BaDev));
        ctx->ba_max_K = ctx->ba_max_P = 0;
        for (size_t i = 0; i < ctx->ba.size() && i < 1024; ++i)
            if (ctx->ba[i]) {
                tab[i] = ba_dev(ctx->ba[i]);
                if (ctx->ba[i]->K > ctx->ba_max_K) ctx->ba_max_K = ctx->ba[i]->K;
                if (ctx->ba[i]->P > ctx->ba_max_P) ctx->ba_max_P = ctx->ba[i]->P;
            }
        if (!ctx->ba_table) TAB_CHK_(hipMalloc(&ctx->ba_table, 1024 * sizeof(BaDev)));
        TAB_CHK_(hipMemcpyAsync(ctx->ba_table, tab.data(), tab.size() * sizeof(BaDev), hipMemcpyHostToDevice, ctx->stream));
        TAB_CHK_(hipStreamSynchronize(ctx->stream));
        ctx->ba_table_dirty = false;
    }
#undef TAB_CHK_
    *rc = YGZ_OK;
    return reinterpret_cast<const BaDev *>(ctx->ba_table);
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
    std::vector<int32_t> free_idx(K, -1), free_pose(K, -1);
    int Kf = 0;
    for (int k = 0; k < K; ++k) if (!(pb->pose_fixed && pb->pose_fixed[k])) { free_idx[k] = Kf; free_pose[Kf] = k; ++Kf; }
    const size_t Kfz = (size_t)(Kf > 0 ? Kf : 1);
    const size_t nd = (size_t)K * 6 + (size_t)P * 3 + Ez * 2 + (size_t)K * BA_POSED + Ez * 6 + Ez + Ez
                    + (size_t)K * 36 + (size_t)K * 6 + (size_t)P * 9 + (size_t)P * 3 + Ez * 18 + Ez * 2 + Ez + 1
                    + (size_t)K * 6 + (size_t)P * 3 + Ez * 18 + (size_t)P * 9 + (size_t)P * 3;          // LM: backups, Y, Dinv, xl
    const size_t ni = Ez * 4 + (size_t)P + 1 + (size_t)K + 1 + 1 + 2 * (size_t)K + (size_t)P * Kfz;
    const size_t bytes = nd * 8 + ni * 4 + (size_t)K + (size_t)P + Ez + 64;
    hipError_t he = hipMalloc(&w->blob, bytes);
    if (he != hipSuccess) { ctx->last_hip_error = (int)he; delete w; return YGZ_E_HIP; }
    double *d = (double *)w->blob;
    w->poses = d; d += (size_t)K * 6; w->points = d; d += (size_t)P * 3; w->obs = d; d += Ez * 2;
    w->posed = d; d += (size_t)K * BA_POSED; w->edge_tmp = d; d += Ez * 6; w->rho0 = d; d += Ez; w->edge_huber = d; d += Ez;
    w->Hpp = d; d += (size_t)K * 36; w->bp = d; d += (size_t)K * 6; w->Hll = d; d += (size_t)P * 9; w->bl = d; d += (size_t)P * 3;
    w->Hpl = d; d += Ez * 18; w->err = d; d += Ez * 2; w->chi2_edge = d; d += Ez; w->chi2 = d; d += 1;
    w->poses_bk = d; d += (size_t)K * 6; w->points_bk = d; d += (size_t)P * 3; w->Y = d; d += Ez * 18; w->Dinv = d; d += (size_t)P * 9;
    w->xl = d; d += (size_t)P * 3;
    int32_t *ii = (int32_t *)d;
    w->edge_pose = ii; ii += Ez; w->edge_point = ii; ii += Ez; w->pt_edges = ii; ii += Ez; w->pose_edges = ii; ii += Ez;
    w->pt_off = ii; ii += (size_t)P + 1; w->pose_off = ii; ii += (size_t)K + 1; w->n_behind = ii; ii += 1;
    w->free_idx = ii; ii += K; w->free_pose = ii; ii += K; w->pt_pose_edge = ii; ii += (size_t)P * Kfz;
    w->Kf = Kf;
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
    {   // edge of (point, free pose): first such edge if a pair is observed twice
        std::vector<int32_t> ppe((size_t)P * Kfz, -1);
        for (int e = E - 1; e >= 0; --e) { const int a = free_idx[pb->edge_pose[e]]; if (a >= 0) ppe[(size_t)pb->edge_point[e] * Kfz + a] = e; }
        YGZ_HIPCHK(ctx, hipMemcpyAsync(w->pt_pose_edge, ppe.data(), ppe.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        YGZ_HIPCHK(ctx, hipMemcpyAsync(w->free_idx, free_idx.data(), (size_t)K * 4, hipMemcpyHostToDevice, ctx->stream));
        YGZ_HIPCHK(ctx, hipMemcpyAsync(w->free_pose, free_pose.data(), (size_t)K * 4, hipMemcpyHostToDevice, ctx->stream));
        YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
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
    int trc = YGZ_OK;
    const BaDev *table = ygz_ba_table(ctx, &trc);
    if (!table) return trc;
    const BaDev *tab = table + window_begin;
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
