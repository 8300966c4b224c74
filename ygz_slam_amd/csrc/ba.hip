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
// sum has a fixed order, and the data so that every per-edge access is a contiguous 512-byte row (ba_dev.h):
//  k_ba_pose_prep  lane = pose: SE3::exp once per pose (q, t, R) instead of once per edge.
//  k_ba_points     lane = map point, wavefront = chunk of 64 points.  Pass 1 walks the point's edges in edge order:
//                  Hll / bl in registers exactly in the oracle's order, the unique 6x3 Hpl block, residual and chi2 of each
//                  edge.  Pass 2 runs over the free poses: lanes whose point sees the pose rebuild the 2x6 pose Jacobian
//                  (cheaper than storing and re-reading it), the 21 + 6 sums are reduced inside the wavefront in a fixed order
//                  and written as one partial per (chunk, pose).
//  k_ba_final      Hpp / bp / chi2 = the partials summed over the chunks in chunk order.
//  k_ba_unpack     edge-order / point-order copies of the chunked arrays for the ABI (download path only).
#include "ba_dev.h"
#include <vector>
#include <string.h>

#define ba_wave_sum ygz_wave_sum_d

__global__ __launch_bounds__(64) void k_ba_pose_prep(const BaDev *__restrict__ wins)
{
    const BaDev B = wins[blockIdx.y];
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k == 0) *B.n_behind = 0;
    if (k >= B.K) return;
    ba_pose_prep_one(B, k);
}

__global__ __launch_bounds__(128) void k_ba_points(const BaDev *__restrict__ wins, int prio)
{
    ygz_raise_prio(prio);                                   // HBM-write bound: keeps its pace beside a VALU-bound kernel
    const BaDev B = wins[blockIdx.y];
    const int il = blockIdx.x * 128 + threadIdx.x, lane = threadIdx.x & 63, q = il >> 6;
    if (q >= B.Q) return;                                   // wavefront-uniform
    const bool live = il < B.P;
    double chi = live ? ba_point_edges(B, il) : 0.0;
    chi = ba_wave_sum(chi);
    if (lane == 0) B.part_chi[q] = chi;
    for (int a = 0; a < B.Kf; ++a) {
        double acc[27];
#pragma unroll
        for (int i = 0; i < 27; ++i) acc[i] = 0.0;
        if (live) ba_pose_contrib(B, il, a, acc);
        double *out = B.part_pose + ((size_t)q * B.Kf + a) * 27;
        int idx;
        const double tot = ba_reduce32(acc, lane, &idx);
        if (lane < 32 && idx < 27) out[idx] = tot;
    }
}

__global__ __launch_bounds__(256) void k_ba_final(const BaDev *__restrict__ wins)
{
    const BaDev B = wins[blockIdx.x];
    for (int t = threadIdx.x; t < B.Kf * 27; t += 256) {
        const int a = t / 27, i = t - 27 * a, k = B.free_pose[a];
        double s = 0.0;
        for (int q = 0; q < B.Q; ++q) s += B.part_pose[((size_t)q * B.Kf + a) * 27 + i];
        if (i < 21) {
            int u = 0, rem = i;                                // unpack upper-triangular index
            while (rem >= 6 - u) { rem -= 6 - u; ++u; }
            const int v = u + rem;
            B.Hpp[36 * (size_t)k + 6 * u + v] = s; B.Hpp[36 * (size_t)k + 6 * v + u] = s;
        } else B.bp[6 * (size_t)k + (i - 21)] = s;
    }
    if (threadIdx.x == 0) { double s = 0.0; for (int q = 0; q < B.Q; ++q) s += B.part_chi[q]; *B.chi2 = s; }
}

// out[e][k] = chunked[(row_e * NC + k) * 64 + lane_e]   (edge order for the ABI)
__global__ __launch_bounds__(256) void k_ba_unpack_edges(const double *__restrict__ src, const int32_t *__restrict__ edge_rl, int E, int NC,
                                                         double *__restrict__ out)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= E * NC) return;
    const int e = t / NC, k = t - e * NC, rl = edge_rl[e];
    out[t] = src[((size_t)(rl >> 6) * NC + k) * 64 + (rl & 63)];
}
__global__ __launch_bounds__(256) void k_ba_unpack_points(const double *__restrict__ src, int P, int NC, double *__restrict__ out)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= P * NC) return;
    const int l = t / NC, k = t - l * NC;
    out[t] = src[((size_t)(l >> 6) * NC + k) * 64 + (l & 63)];
}

bool ygz_ba_window_has_dup(const ygz_hip_ctx *ctx, int window)
{ return window >= 0 && window < (int)ctx->ba.size() && ctx->ba[window] && ctx->ba[window]->has_dup; }

static void ba_free(ygz_hip_ctx::BaWindow *w) { if (w) { if (w->blob) (void)hipFree(w->blob); delete w; } }

extern "C" void ygz_hip_ba_free_all(ygz_hip_ctx *ctx)
{
    YgzDeviceGuard dg_(ctx);
    for (auto *w : ctx->ba) ba_free(w);
    ctx->ba.clear();
    if (ctx->ba_table) { (void)hipFree(ctx->ba_table); ctx->ba_table = nullptr; }
    if (ctx->ba_table_host) { delete static_cast<std::vector<BaDev> *>(ctx->ba_table_host); ctx->ba_table_host = nullptr; }
}

static BaDev ba_dev(const ygz_hip_ctx::BaWindow *w)
{
    BaDev B;
    B.K = w->K; B.P = w->P; B.E = w->E; B.formulation = w->formulation; B.Kf = w->Kf; B.R = w->R; B.Q = w->Q;
    B.fx = w->fx; B.fy = w->fy; B.cx = w->cx; B.cy = w->cy; B.huber = w->huber;
    B.poses = w->poses; B.points = w->points; B.posed = w->posed;
    B.obs_c = w->obs_c; B.huber_c = w->huber_c; B.pose_c = w->pose_c; B.enable_c = w->enable_c; B.slot_off = w->slot_off; B.ppc = w->ppc; B.dupn = w->dupn;
    B.edge_rl = w->edge_rl; B.fixed = w->fixed; B.point_fixed = w->point_fixed; B.free_idx = w->free_idx; B.free_pose = w->free_pose;
    B.n_behind = w->n_behind;
    B.Hpp = w->Hpp; B.bp = w->bp; B.chi2 = w->chi2; B.Hll_c = w->Hll_c; B.bl_c = w->bl_c; B.Hpl_c = w->Hpl_c; B.err_c = w->err_c;
    B.chi2e_c = w->chi2e_c; B.part_pose = w->part_pose; B.part_chi = w->part_chi;
    B.poses_w = w->poses; B.points_w = w->points; B.poses_bk = w->poses_bk; B.points_bk = w->points_bk;
    B.Y_c = w->Y_c; B.Dinv = w->Dinv; B.xl = w->xl; B.sc_p = w->sc_p; B.sc_l = w->sc_l; B.lm_out = w->lm_out;
    return B;
}

// descriptor table of all windows: an entry changes when its window is uploaded / reserved (host) or rebuilt on the device (the build
// kernel patches the sizes in place), so only the entries of windows marked dirty are copied
const BaDev *ygz_ba_table(ygz_hip_ctx *ctx, int *rc)
{
#define TAB_CHK_(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { ctx->last_hip_error = (int)e_; *rc = YGZ_E_HIP; return nullptr; } } while (0)
    if (ctx->ba_table_dirty) { const int rj = ygz_join(ctx); if (rj != YGZ_OK) { *rc = rj; return nullptr; } }
    if (ctx->ba_table_dirty) {
        if (!ctx->ba_table_host) { auto *v = new std::vector<BaDev>(1024); memset(v->data(), 0, v->size() * sizeof(BaDev)); ctx->ba_table_host = v; }
        std::vector<BaDev> &tab = *static_cast<std::vector<BaDev> *>(ctx->ba_table_host);
        if (!ctx->ba_table) {
            TAB_CHK_(hipMalloc(&ctx->ba_table, 1024 * sizeof(BaDev)));
            TAB_CHK_(hipMemsetAsync(ctx->ba_table, 0, 1024 * sizeof(BaDev), ctx->stream));
        }
        ctx->ba_max_K = ctx->ba_max_P = 0;
        for (size_t i = 0; i < ctx->ba.size() && i < 1024; ++i) {
            auto *w = ctx->ba[i];
            if (!w) continue;
            if (w->K > ctx->ba_max_K) ctx->ba_max_K = w->K;
            if (w->P > ctx->ba_max_P) ctx->ba_max_P = w->P;
            if (!w->table_dirty) continue;
            tab[i] = ba_dev(w);
            TAB_CHK_(hipMemcpyAsync((BaDev *)ctx->ba_table + i, &tab[i], sizeof(BaDev), hipMemcpyHostToDevice, ctx->stream));
            w->table_dirty = false;
        }
        TAB_CHK_(hipStreamSynchronize(ctx->stream));
        ctx->ba_table_dirty = false;
    }
#undef TAB_CHK_
    *rc = YGZ_OK;
    return reinterpret_cast<const BaDev *>(ctx->ba_table);
}

// one allocation per window: doubles, then int32, then int16, then bytes; R rows, E edges, Q chunks, Kf free poses (each >= 1 here)
static int ba_carve(ygz_hip_ctx *ctx, ygz_hip_ctx::BaWindow *w, size_t K, size_t P, size_t Rz, size_t Ez, size_t Q, size_t Kfz)
{
    const size_t nd = K * 6 + P * 3 + K * BA_POSED                                                  // poses, points, posed
                    + Rz * 128 + Rz * 64                                                            // obs_c, huber_c
                    + K * 36 + K * 6 + 1 + Q * 9 * 64 + Q * 3 * 64                                  // Hpp, bp, chi2, Hll_c, bl_c
                    + Rz * 18 * 64 + Rz * 2 * 64 + Rz * 64                                          // Hpl_c, err_c, chi2e_c
                    + Q * Kfz * 27 + Q                                                              // partials
                    + K * 6 + P * 3 + Rz * 18 * 64 + P * 9 + P * 3                                  // LM: backups, Y_c, Dinv, xl
                    + K * 6 + P * 3                                                                 // trust-region loop: Jacobi scales
                    + 16;                                                                           // lm_out
    const size_t ni = Rz * 64 + Q + 1 + Ez + 2 * K + 1;
    const size_t ns = P * Kfz + 1 + Rz * 64;
    const size_t bytes = nd * 8 + ni * 4 + ((ns * 2 + 3) & ~(size_t)3) + K + P + Rz * 64 + 64;
    if (!w->blob || w->blob_bytes < bytes) {             // w->blob: the allocation of the window this one replaces (ygz_hip_ba_upload)
        if (w->blob) { (void)hipFree(w->blob); w->blob = nullptr; w->blob_bytes = 0; }
        const size_t cap = bytes + bytes / 4;                // head room: the next window of this slot is about this size
        hipError_t he = hipMalloc(&w->blob, cap);
        if (he != hipSuccess) { ctx->last_hip_error = (int)he; return YGZ_E_HIP; }
        w->blob_bytes = cap;
    }
    double *d = (double *)w->blob;
    w->poses = d; d += K * 6; w->points = d; d += P * 3; w->posed = d; d += K * BA_POSED;
    w->obs_c = d; d += Rz * 128; w->huber_c = d; d += Rz * 64;
    w->Hpp = d; d += K * 36; w->bp = d; d += K * 6; w->chi2 = d; d += 1;
    w->Hll_c = d; d += Q * 9 * 64; w->bl_c = d; d += Q * 3 * 64;
    w->Hpl_c = d; d += Rz * 18 * 64; w->err_c = d; d += Rz * 2 * 64; w->chi2e_c = d; d += Rz * 64;
    w->part_pose = d; d += Q * Kfz * 27; w->part_chi = d; d += Q;
    w->poses_bk = d; d += K * 6; w->points_bk = d; d += P * 3; w->Y_c = d; d += Rz * 18 * 64;
    w->Dinv = d; d += P * 9; w->xl = d; d += P * 3;
    w->sc_p = d; d += K * 6; w->sc_l = d; d += P * 3;
    w->lm_out = d; d += 16;
    int32_t *ii = (int32_t *)d;
    w->pose_c = ii; ii += Rz * 64; w->slot_off = ii; ii += Q + 1; w->edge_rl = ii; ii += Ez;
    w->free_idx = ii; ii += K; w->free_pose = ii; ii += K; w->n_behind = ii; ii += 1;
    w->ppc = (int16_t *)ii; w->dupn = w->ppc + P * Kfz + 1;
    uint8_t *bb = (uint8_t *)ii + ((ns * 2 + 3) & ~(size_t)3);
    w->fixed = bb; bb += K; w->point_fixed = bb; bb += P; w->enable_c = bb;
    YGZ_HIPCHK(ctx, hipMemsetAsync(w->lm_out, 0xFF, 16 * 8, ctx->stream));        // iterations = -1: no resident LM has run
    return YGZ_OK;
}

// bytes of the region [Hpp, part_pose) that the kernels never write for constant poses / padding lanes: zeroed at upload / build
size_t ygz_ba_zero_bytes(const ygz_hip_ctx::BaWindow *w) { return (size_t)((const uint8_t *)w->part_pose - (const uint8_t *)w->Hpp); }

// Capacity window for a graph that ygz_hip_ba_build_windows assembles on the device (window.hip): K poses of which pose 0 is constant,
// up to P points, every point seen by up to K poses.
int ygz_ba_reserve_window(ygz_hip_ctx *ctx, int window, int K, int P, double huber)
{
    if (window < 0 || window > 1022 || K < 2 || K > 16 || P < 1) return YGZ_E_INVALID;
    if ((int)ctx->ba.size() <= window) ctx->ba.resize(window + 1, nullptr);
    if (ctx->ba[window]) {
        auto *o = ctx->ba[window];
        if (o->device_built && o->cap_K == K && o->cap_P == P && o->huber == huber) return YGZ_OK;
        YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); ba_free(o); ctx->ba[window] = nullptr;
    }
    auto *w = new ygz_hip_ctx::BaWindow();
    const int Q = (P + 63) / 64, R = Q * K;
    w->K = K; w->P = P; w->E = P * K; w->formulation = 0; w->Kf = K - 1; w->R = R; w->Q = Q;
    w->fx = ctx->prm.fx; w->fy = ctx->prm.fy; w->cx = ctx->prm.cx; w->cy = ctx->prm.cy; w->huber = huber;
    w->device_built = true; w->cap_K = K; w->cap_P = P;
    const int rc = ba_carve(ctx, w, (size_t)K, (size_t)P, (size_t)R, (size_t)P * K, (size_t)Q, (size_t)(K - 1));
    if (rc != YGZ_OK) { delete w; return rc; }
    ctx->ba[window] = w;
    w->table_dirty = true; ctx->ba_table_dirty = true;
    return YGZ_OK;
}

extern "C" {

int ygz_hip_ba_upload(ygz_hip_ctx *ctx, int window, const ygz_ba_problem *pb)
{
    YgzDeviceGuard dg_(ctx);
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || !pb || window < 0 || window > 1023) return YGZ_E_INVALID;
    const int K = pb->n_poses, P = pb->n_points, E = pb->n_edges;
    if (K < 1 || P < 1 || E < 0 || !pb->poses || !pb->points || (E > 0 && (!pb->edge_pose || !pb->edge_point || !pb->obs)))
        return YGZ_E_INVALID;
    if (pb->formulation < 0 || pb->formulation > 2) return YGZ_E_INVALID;
    for (int e = 0; e < E; ++e)
        if (pb->edge_pose[e] < 0 || pb->edge_pose[e] >= K || pb->edge_point[e] < 0 || pb->edge_point[e] >= P) return YGZ_E_INVALID;
    if ((int)ctx->ba.size() <= window) ctx->ba.resize(window + 1, nullptr);
    // the allocation of the window this one replaces is kept when it is large enough (LocalBAG2O uploads a window of about the same size for
    // every keyframe: a hipFree + hipMalloc pair per call otherwise)
    void *old_blob = nullptr; size_t old_bytes = 0;
    if (ctx->ba[window]) {
        YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        auto *o = ctx->ba[window];
        old_blob = o->blob; old_bytes = o->blob_bytes; o->blob = nullptr;
        ba_free(o); ctx->ba[window] = nullptr;
    }
    auto *w = new ygz_hip_ctx::BaWindow();
    w->blob = old_blob; w->blob_bytes = old_bytes;
    w->K = K; w->P = P; w->E = E; w->formulation = pb->formulation;
    w->fx = pb->fx; w->fy = pb->fy; w->cx = pb->cx; w->cy = pb->cy; w->huber = pb->huber_delta;
    // ---- rows: the c-th edge (ascending edge order) of every point of a 64-point chunk
    const int Q = (P + 63) / 64;
    std::vector<int32_t> cnt(P, 0), slot_off_v(Q + 1, 0);
    for (int e = 0; e < E; ++e) cnt[pb->edge_point[e]]++;
    for (int q = 0; q < Q; ++q) {
        int mx = 0;
        for (int l = 64 * q; l < std::min(P, 64 * q + 64); ++l) mx = std::max(mx, cnt[l]);
        if (mx > 32767) { ba_free(w); return YGZ_E_CAPACITY; }
        slot_off_v[q + 1] = slot_off_v[q] + mx;
    }
    const int R = slot_off_v[Q];
    const size_t Rz = (size_t)(R > 0 ? R : 1), Ez = (size_t)(E > 0 ? E : 1);
    int Kf = 0;
    for (int k = 0; k < K; ++k) if (!(pb->pose_fixed && pb->pose_fixed[k])) ++Kf;
    const size_t Kfz = (size_t)(Kf > 0 ? Kf : 1);
    w->Kf = Kf; w->R = R; w->Q = Q;
    { const int rcv = ba_carve(ctx, w, (size_t)K, (size_t)P, Rz, Ez, (size_t)Q, Kfz); if (rcv != YGZ_OK) { ba_free(w); return rcv; } }
    // ---- the host image of the five uploaded regions, assembled in ONE page-locked block in the order the blob keeps them (ba_carve), so that
    // each region goes up as one copy: [poses | points], [obs_c | huber_c], [pose_c | slot_off | edge_rl | free_idx | free_pose],
    // [ppc | dupn], [fixed | point_fixed | enable_c]
    const size_t b0 = ((size_t)K * 6 + (size_t)P * 3) * 8, b1 = Rz * 192 * 8, b2 = (Rz * 64 + (size_t)Q + 1 + Ez + 2 * (size_t)K) * 4,
                 b3 = ((size_t)P * Kfz + 1 + Rz * 64) * 2, b4 = (size_t)K + (size_t)P + Rz * 64;
    const size_t o1 = (b0 + 63) & ~(size_t)63, o2 = o1 + ((b1 + 63) & ~(size_t)63), o3 = o2 + ((b2 + 63) & ~(size_t)63), o4 = o3 + ((b3 + 63) & ~(size_t)63);
    uint8_t *hs = (uint8_t *)ygz_stage(ctx, o4 + b4);
    if (!hs) { ba_free(w); return YGZ_E_HIP; }
    double *h_poses = (double *)hs, *h_points = h_poses + (size_t)K * 6;
    double *obs_c = (double *)(hs + o1), *huber_c = obs_c + Rz * 128;
    int32_t *pose_c = (int32_t *)(hs + o2), *slot_off = pose_c + Rz * 64, *edge_rl = slot_off + Q + 1, *free_idx = edge_rl + Ez, *free_pose = free_idx + K;
    int16_t *ppc = (int16_t *)(hs + o3), *dupn = ppc + (size_t)P * Kfz + 1;
    uint8_t *fixed = hs + o4, *pfixed = fixed + K, *enable_c = pfixed + P;
    memcpy(h_poses, pb->poses, (size_t)K * 48); memcpy(h_points, pb->points, (size_t)P * 24);
    memset(obs_c, 0, b1);
    memset(pose_c, 0xFF, Rz * 64 * 4);                                        // -1: no such edge
    memcpy(slot_off, slot_off_v.data(), ((size_t)Q + 1) * 4);
    memset(edge_rl, 0, Ez * 4);
    memset(free_idx, 0xFF, (size_t)K * 8);                                    // free_idx and free_pose: -1
    memset(ppc, 0xFF, b3);                                                    // ppc, the pad element and dupn: -1
    memset(fixed, 0, b4);
    if (pb->pose_fixed) memcpy(fixed, pb->pose_fixed, K);
    if (pb->point_fixed) memcpy(pfixed, pb->point_fixed, P);
    { int a = 0; for (int k = 0; k < K; ++k) if (!fixed[k]) { free_idx[k] = a; free_pose[a] = k; ++a; } }
    std::vector<int16_t> last_c((size_t)P * Kfz, -1);
    std::fill(cnt.begin(), cnt.end(), 0);
    for (int e = 0; e < E; ++e) {
        const int l = pb->edge_point[e], c = cnt[l]++, row = slot_off[l >> 6] + c, lane = l & 63, ip = pb->edge_pose[e];
        edge_rl[e] = row * 64 + lane;
        pose_c[(size_t)row * 64 + lane] = ip;
        obs_c[((size_t)row * 2) * 64 + lane] = pb->obs[2 * (size_t)e]; obs_c[((size_t)row * 2 + 1) * 64 + lane] = pb->obs[2 * (size_t)e + 1];
        huber_c[(size_t)row * 64 + lane] = pb->edge_huber ? pb->edge_huber[e] : pb->huber_delta;
        enable_c[(size_t)row * 64 + lane] = pb->edge_enable ? (pb->edge_enable[e] ? 1 : 0) : 1;
        const int a = free_idx[ip];
        if (a >= 0) {
            const size_t pa = (size_t)l * Kfz + a;
            if (ppc[pa] < 0) ppc[pa] = (int16_t)c;
            else { dupn[(size_t)(slot_off[l >> 6] + last_c[pa]) * 64 + lane] = (int16_t)c; w->has_dup = true; }    // a repeated (point, free pose) pair
            last_c[pa] = (int16_t)c;
        }
    }
    w->h_edge_rl.assign(edge_rl, edge_rl + Ez);
    ctx->ba[window] = w;
    w->table_dirty = true; ctx->ba_table_dirty = true;
#define UP_(dst, src, n) do { const int rk_ = ygz_kcopy(ctx, (void *)(dst), (src), (n), (int)hipMemcpyHostToDevice); if (rk_ != YGZ_OK) return rk_; } while (0)
    UP_(w->poses, hs, b0); UP_(w->obs_c, hs + o1, b1); UP_(w->pose_c, hs + o2, b2); UP_(w->ppc, hs + o3, b3); UP_(w->fixed, hs + o4, b4);
#undef UP_
    // blocks of constant poses / padding lanes are never written by the kernels: zero once
    YGZ_HIPCHK(ctx, hipMemsetAsync(w->Hpp, 0, ygz_ba_zero_bytes(w), ctx->stream));
    // no synchronisation: the staging block stays valid until the arena wraps (ygz_stage synchronises the stream before it recycles)
    return YGZ_OK;
}

int ygz_hip_ba_set_state(ygz_hip_ctx *ctx, int window, const double *poses, const double *points)
{
    YgzDeviceGuard dg_(ctx);
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
    YgzDeviceGuard dg_(ctx);
    // only a pending BA stage reads what this overwrites; sparse alignment / matcher stages of the last step keep running
    if (ctx) { int rj_ = ygz_join(ctx, ~(1u << YGZ_AUX_BA)); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || window < 0 || window >= (int)ctx->ba.size() || !ctx->ba[window]) return YGZ_E_INVALID;
    auto *w = ctx->ba[window];
    if (d_poses) YGZ_HIPCHK(ctx, hipMemcpyAsync(w->poses, d_poses, (size_t)w->K * 48, hipMemcpyDeviceToDevice, ctx->stream));
    if (d_points) YGZ_HIPCHK(ctx, hipMemcpyAsync(w->points, d_points, (size_t)w->P * 24, hipMemcpyDeviceToDevice, ctx->stream));
    return YGZ_OK;
}

int ygz_hip_ba_linearize_resident(ygz_hip_ctx *ctx, int window_begin, int n_windows)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || window_begin < 0 || n_windows < 1 || window_begin + n_windows > (int)ctx->ba.size()) return YGZ_E_INVALID;
    for (int i = window_begin; i < window_begin + n_windows; ++i) if (!ctx->ba[i]) return YGZ_E_INVALID;
    int trc = YGZ_OK;
    const BaDev *table = ygz_ba_table(ctx, &trc);
    if (!table) return trc;
    const BaDev *tab = table + window_begin;
    YgzAuxScope aux(ctx, 1);
    YGZ_LAUNCH(ctx, KID_BA_POSE_PREP, k_ba_pose_prep, dim3(ygz_div_up(ctx->ba_max_K, 64), n_windows), dim3(64), tab);
    YGZ_LAUNCH(ctx, KID_BA_POINTS, k_ba_points, dim3(ygz_div_up(ctx->ba_max_P, 128), n_windows), dim3(128), tab, (ctx->wave_prio_mask >> 2) & 1);
    YGZ_LAUNCH(ctx, KID_BA_POSES, k_ba_final, dim3(n_windows), dim3(256), tab);
    YGZ_HIPCHK(ctx, hipGetLastError());
    return YGZ_OK;
}

int ygz_hip_ba_download(ygz_hip_ctx *ctx, int window, double *Hpp, double *bp, double *Hll, double *bl, double *Hpl,
                        double *err, double *chi2_edge, double *chi2)
{
    YgzDeviceGuard dg_(ctx);
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || window < 0 || window >= (int)ctx->ba.size() || !ctx->ba[window]) return YGZ_E_INVALID;
    auto *w = ctx->ba[window];
    size_t K = w->K, P = w->P, E = w->E;
    if (w->device_built) {                                   // the actual sizes live in the device table entry (k_win_edges)
        int dims[7];
        if (!ctx->ba_table || w->table_dirty) return YGZ_E_STATE;
        YGZ_HIPCHK(ctx, hipMemcpyAsync(dims, (const BaDev *)ctx->ba_table + window, sizeof(dims), hipMemcpyDeviceToHost, ctx->stream));
        YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        K = (size_t)dims[0]; P = (size_t)dims[1]; E = (size_t)dims[2];
    }
#define DL_(dst, src, n) if ((dst) && (n) > 0) YGZ_HIPCHK(ctx, hipMemcpyAsync((dst), (src), (n) * 8, hipMemcpyDeviceToHost, ctx->stream))
    DL_(Hpp, w->Hpp, K * 36); DL_(bp, w->bp, K * 6); DL_(chi2, w->chi2, (size_t)1);
    // chunked -> ABI order through a staging buffer
    const size_t need = std::max(E * 18, P * 9);
    double *stage = nullptr;
    if ((Hll || bl || Hpl || err || chi2_edge) && need > 0) {
        int rc = ygz_scratch(ctx, SCR_BA_0, need * 8, (void **)&stage);
        if (rc != YGZ_OK) return rc;
    }
#define EDGE_(dst, src, NC) if ((dst) && E > 0) { k_ba_unpack_edges<<<dim3(ygz_div_up((int)(E * (NC)), 256)), dim3(256), 0, ctx->stream>>>((src), w->edge_rl, (int)E, (NC), stage); \
                                                  YGZ_HIPCHK(ctx, hipMemcpyAsync((dst), stage, E * (NC) * 8, hipMemcpyDeviceToHost, ctx->stream)); \
                                                  YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); }
#define POINT_(dst, src, NC) if (dst) { k_ba_unpack_points<<<dim3(ygz_div_up((int)(P * (NC)), 256)), dim3(256), 0, ctx->stream>>>((src), (int)P, (NC), stage); \
                                        YGZ_HIPCHK(ctx, hipMemcpyAsync((dst), stage, P * (NC) * 8, hipMemcpyDeviceToHost, ctx->stream)); \
                                        YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); }
    POINT_(Hll, w->Hll_c, 9) POINT_(bl, w->bl_c, 3)
    EDGE_(Hpl, w->Hpl_c, 18) EDGE_(err, w->err_c, 2) EDGE_(chi2_edge, w->chi2e_c, 1)
#undef EDGE_
#undef POINT_
#undef DL_
    YGZ_HIPCHK(ctx, hipGetLastError());
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return YGZ_OK;
}

}  // extern "C"
// What LocalBAG2O reads after its optimisation -- the statistics record of the resident LM run (d_stats), the optimised state and the per-edge chi2 of
// the linearisation that followed -- gathered into ONE page-locked block and waited for once (three transfers with a wait each before: 0.13 ms per call).
int ygz_ba_fetch_result(ygz_hip_ctx *ctx, int window, const void *d_stats, ygz_ba_stats *stats, double *poses, double *points, double *chi2_edge)
{
    { int rj = ygz_join(ctx); if (rj != YGZ_OK) return rj; }
    if (window < 0 || window >= (int)ctx->ba.size() || !ctx->ba[window] || ctx->ba[window]->device_built) return YGZ_E_INVALID;
    auto *w = ctx->ba[window];
    const size_t K = w->K, P = w->P, E = w->E;
    double *stage = nullptr;
    int rc = YGZ_OK;
    if (chi2_edge && E > 0) {
        if ((rc = ygz_scratch(ctx, SCR_GEN_0 + 9, E * 8, (void **)&stage)) != YGZ_OK) return rc;
        k_ba_unpack_edges<<<dim3(ygz_div_up((int)E, 256)), dim3(256), 0, ctx->stream>>>(w->chi2e_c, w->edge_rl, (int)E, 1, stage);
        YGZ_HIPCHK(ctx, hipGetLastError());
    }
    YgzPack pk;
    if ((rc = ygz_pack_begin(ctx, &pk, sizeof(ygz_ba_stats) + K * 48 + P * 24 + E * 8 + 64, SCR_GEN_0 + 8)) != YGZ_OK) return rc;
    void *h_st = ygz_pack_add(&pk, d_stats, sizeof(ygz_ba_stats)), *h_po = ygz_pack_add(&pk, w->poses, K * 48), *h_pt = ygz_pack_add(&pk, w->points, P * 24);
    void *h_c = stage ? ygz_pack_add(&pk, stage, E * 8) : nullptr;
    if (!h_st || !h_po || !h_pt || (stage && !h_c)) return YGZ_E_CAPACITY;
    if ((rc = ygz_pack_fetch(ctx, &pk)) != YGZ_OK) return rc;
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (stats) memcpy(stats, h_st, sizeof(ygz_ba_stats));
    if (poses) memcpy(poses, h_po, K * 48);
    if (points) memcpy(points, h_pt, P * 24);
    if (h_c) memcpy(chi2_edge, h_c, E * 8);
    return YGZ_OK;
}
extern "C" {

int ygz_hip_ba_behind_camera(ygz_hip_ctx *ctx, int window, int *n_behind)
{
    YgzDeviceGuard dg_(ctx);
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
    YgzDeviceGuard dg_(ctx);
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || !edge_enable || window < 0 || window >= (int)ctx->ba.size() || !ctx->ba[window]) return YGZ_E_INVALID;
    auto *w = ctx->ba[window];
    if (w->device_built) return YGZ_E_STATE;                 // the host does not know where the edges of a device-built graph live
    std::vector<uint8_t> en((size_t)(w->R > 0 ? w->R : 1) * 64, 0);
    for (int e = 0; e < w->E; ++e) en[w->h_edge_rl[e]] = edge_enable[e] ? 1 : 0;
    YGZ_HIPCHK(ctx, hipMemcpyAsync(w->enable_c, en.data(), en.size(), hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return YGZ_OK;
}

int ygz_hip_ba_linearize(ygz_hip_ctx *ctx, const ygz_ba_problem *pb, double *Hpp, double *bp, double *Hll, double *bl,
                         double *Hpl, double *err, double *chi2_edge, double *chi2)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx) return YGZ_E_INVALID;
    const int window = 1023;          // private slot for the one-shot form
    int rc = ygz_hip_ba_upload(ctx, window, pb);
    if (rc == YGZ_OK) rc = ygz_hip_ba_linearize_resident(ctx, window, 1);
    if (rc == YGZ_OK) rc = ygz_hip_ba_download(ctx, window, Hpp, bp, Hll, bl, Hpl, err, chi2_edge, chi2);
    return rc;
}

}  // extern "C"
