// Resident, batched tracking state: the per-frame path of VisualOdometry::AddFrame / LocalMapping::TrackLocalMap
// (src/Module/VisualOdometry.cpp:38-107, src/Module/LocalMapping.cpp:24-120) run for MANY frame pairs per launch
// with no host round trip between the stages.  A "track set" per pair holds the reference features
// (pixel, level, depth, has-map-point); it is filled on the device from the keypoints the extractor left in HBM
// (k_track_load) or from host arrays by the single-pair entry points of the ABI.
#include "ygz_internal.h"
#include "se3_dev.h"
#include <stdlib.h>
#include <string.h>

int ygz_track_ensure(ygz_hip_ctx *ctx)
{
    if (ctx->trk_alloc) return YGZ_OK;
    const size_t F = (size_t)ctx->prm.max_frames, Cn = (size_t)ctx->cells;
    hipError_t e = hipSuccess;
#define A_(ptr, bytes) if (e == hipSuccess) e = hipMalloc((void **)&(ptr), (bytes))
    A_(ctx->trk_n, F * 4); A_(ctx->trk_px, F * Cn * 16); A_(ctx->trk_level, F * Cn * 4); A_(ctx->trk_depth, F * Cn * 8);
    A_(ctx->trk_has_mp, F * Cn); A_(ctx->pair_T, F * 14 * 8); A_(ctx->kp_depth, F * Cn * 8); A_(ctx->kp_has_mp, F * Cn);
    A_(ctx->klt_pts, F * Cn * 8); A_(ctx->klt_err, F * Cn * 4); A_(ctx->klt_status, F * Cn);
    A_(ctx->fdp_px, F * Cn * 16); A_(ctx->fdp_level, F * Cn * 4); A_(ctx->fdp_ok, F * Cn); A_(ctx->sa_out, F * 16 * 8);
    // sparse-align work per pair: jac_cache 768 B + patch_cache 64 B + r2 64 B + visible 1 B per feature
    ctx->sa_work_stride = ((Cn * (768 + 64 + 64 + 4 + 4 + 16 + 8 + 2) + 255) / 256) * 256;  // caches | patch | chain terms | ctot | pre | fmap | pmap | visible, used
    A_(ctx->sa_work, F * ctx->sa_work_stride);
    A_(ctx->fdp_cand, F * Cn); A_(ctx->po_pw, F * Cn * 24); A_(ctx->po_pose, F * 48); A_(ctx->po_T, F * 56); A_(ctx->po_depth, F * Cn * 8);
    A_(ctx->po_bad, F * Cn); A_(ctx->po_cnt, F * 8);
#undef A_
    if (e != hipSuccess) { ctx->last_hip_error = (int)e; return YGZ_E_HIP; }
    YGZ_HIPCHK(ctx, hipMemsetAsync(ctx->trk_n, 0, F * 4, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemsetAsync(ctx->kp_depth, 0, F * Cn * 8, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemsetAsync(ctx->kp_has_mp, 0, F * Cn, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemsetAsync(ctx->fdp_cand, 1, F * Cn, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemsetAsync(ctx->po_pose, 0, F * 48, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemsetAsync(ctx->po_T, 0, F * 56, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemsetAsync(ctx->po_cnt, 0, F * 8, ctx->stream));
    ctx->trk_alloc = true;
    return YGZ_OK;
}

int ygz_track_set_pairs(ygz_hip_ctx *ctx, const int32_t *cur_slot, const int32_t *ref_slot, const double *T_cur,
                        const double *T_ref, int n_pairs, YgzPack *pk)
{
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (n_pairs < 1 || n_pairs > ctx->prm.max_frames) return YGZ_E_CAPACITY;
    for (int i = 0; i < n_pairs; ++i)
        if (cur_slot[i] < 0 || cur_slot[i] >= ctx->prm.max_frames || ref_slot[i] < 0 || ref_slot[i] >= ctx->prm.max_frames) return YGZ_E_INVALID;
    int rc = ygz_track_ensure(ctx);
    if (rc != YGZ_OK) return rc;
    const int F = ctx->prm.max_frames;
    // page-locked copies of the caller's tables (ygz_stage): no wait for the stream, the call can be enqueued behind running work
    if (!ctx->klt_slots) YGZ_HIPCHK(ctx, hipMalloc((void **)&ctx->klt_slots, (size_t)F * 8));
    int32_t *h_q, *h_t, *h_lst; double *T;
    if (pk) {                                                    // slices of the caller's packed upload (a single-frame call: one transfer for everything)
        h_q = (int32_t *)ygz_pack_add(pk, ctx->pair_q, (size_t)n_pairs * 4); h_t = (int32_t *)ygz_pack_add(pk, ctx->pair_t, (size_t)n_pairs * 4);
        T = (double *)ygz_pack_add(pk, ctx->pair_T, (size_t)n_pairs * 14 * 8); h_lst = (int32_t *)ygz_pack_add(pk, ctx->klt_slots, (size_t)F * 8);
        if (!h_q || !h_t || !T || !h_lst) return YGZ_E_CAPACITY;
    } else {
        uint8_t *st = (uint8_t *)ygz_stage(ctx, (size_t)n_pairs * (8 + 14 * 8) + (size_t)F * 8);
        if (!st) return YGZ_E_HIP;
        h_q = (int32_t *)st; h_t = h_q + n_pairs; h_lst = h_t + n_pairs;
        T = (double *)(st + (((size_t)n_pairs * 8 + (size_t)F * 8 + 7) & ~(size_t)7));
    }
    memcpy(h_q, cur_slot, (size_t)n_pairs * 4); memcpy(h_t, ref_slot, (size_t)n_pairs * 4);
    for (int i = 0; i < n_pairs; ++i)
        for (int k = 0; k < 7; ++k) { T[14 * i + k] = T_ref[7 * i + k]; T[14 * i + 7 + k] = T_cur[7 * i + k]; }
    if (!pk) {
        YGZ_HIPCHK(ctx, hipMemcpyAsync(ctx->pair_q, h_q, (size_t)n_pairs * 4, hipMemcpyHostToDevice, ctx->stream));
        YGZ_HIPCHK(ctx, hipMemcpyAsync(ctx->pair_t, h_t, (size_t)n_pairs * 4, hipMemcpyHostToDevice, ctx->stream));
        YGZ_HIPCHK(ctx, hipMemcpyAsync(ctx->pair_T, T, (size_t)n_pairs * 14 * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    {   // distinct slots of the table (a frame is usually the current frame of one pair and the reference of the next): the
        // tracker prepares its working images once per slot
        std::vector<uint8_t> any(F, 0), isref(F, 0);
        for (int i = 0; i < n_pairs; ++i) { any[cur_slot[i]] = 1; any[ref_slot[i]] = 1; isref[ref_slot[i]] = 1; }
        int n = 0;
        for (int s = 0; s < F; ++s) if (any[s]) h_lst[n++] = s;
        ctx->n_klt_slots = n;
        for (int s = 0; s < F; ++s) if (isref[s]) h_lst[n++] = s;
        ctx->n_klt_refs = n - ctx->n_klt_slots;
        if (n > 0 && !pk) YGZ_HIPCHK(ctx, hipMemcpyAsync(ctx->klt_slots, h_lst, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
        ctx->klt_slots_host.assign(h_lst, h_lst + ctx->n_klt_slots);
    }
    ctx->n_pairs = n_pairs;
    ctx->klt_prep_valid = false;
    return YGZ_OK;
}

struct LoadArgs {
    const int32_t *pair_t, *n_kp; int cells;
    const double *kp_px; const int32_t *kp_level; const double *kp_depth; const uint8_t *kp_has_mp;
    const double *pair_T;
    int32_t *trk_n; double *trk_px; int32_t *trk_level; double *trk_depth; uint8_t *trk_has_mp;
    float *klt_pts; double *fdp_px; double *sa_out; uint8_t *fdp_cand;
    float fx, fy, cx, cy; int predict;
};

__global__ __launch_bounds__(256) void k_track_load(LoadArgs A)
{
    const int p = blockIdx.y, ref = A.pair_t[p];
    const int n = A.n_kp[ref];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) {
        A.trk_n[p] = n;
        for (int k = 0; k < 7; ++k) A.sa_out[16 * (size_t)p + k] = A.pair_T[14 * (size_t)p + k];   // current->_TCW = ref->_TCW (Matcher.cpp:471)
    }
    if (i >= n) return;
    const size_t s = (size_t)ref * A.cells + i, d = (size_t)p * A.cells + i;
    const double x = A.kp_px[2 * s], y = A.kp_px[2 * s + 1], dep = A.kp_depth[s];
    A.trk_px[2 * d] = x; A.trk_px[2 * d + 1] = y;
    A.trk_level[d] = A.kp_level[s]; A.trk_depth[d] = dep; A.trk_has_mp[d] = A.kp_has_mp[s];
    A.klt_pts[2 * d] = (float)x; A.klt_pts[2 * d + 1] = (float)y;            // Tracker::SetReference (Tracker.cpp:27-31)
    double ox = x, oy = y;
    if (A.predict && dep > 0) {
        // candidate projection with the current pose estimate (LocalMapping::FindCandidates, LocalMapping.cpp:47-80)
        Se3 Tr, Tc, Tri, TCR;
        for (int k = 0; k < 4; ++k) { Tr.q[k] = A.pair_T[14 * (size_t)p + k]; Tc.q[k] = A.pair_T[14 * (size_t)p + 7 + k]; }
        for (int k = 0; k < 3; ++k) { Tr.t[k] = A.pair_T[14 * (size_t)p + 4 + k]; Tc.t[k] = A.pair_T[14 * (size_t)p + 11 + k]; }
        se3_inv_d(&Tr, &Tri); se3_mul_d(&Tc, &Tri, &TCR);
        const double pr[3] = { (x - A.cx) * dep / A.fx, (y - A.cy) * dep / A.fy, dep };
        double pc[3];
        se3_act_d(&TCR, pr, pc);
        ox = A.fx * pc[0] / pc[2] + A.cx; oy = A.fy * pc[1] / pc[2] + A.cy;
    }
    A.fdp_px[2 * d] = ox; A.fdp_px[2 * d + 1] = oy;
    A.fdp_cand[d] = 1;
}

// VisualOdometry::TrackRefFrame -> TrackLocalMap hand-over for every pair: the pose sparse alignment left in sa_out becomes the
// pair's current pose (_curr_frame->_TCW = _TCR_estimated * _ref_frame->_TCW, VisualOdometry.cpp:293) and each reference feature's
// map point is projected with it (LocalMapping::FindCandidates, LocalMapping.cpp:47-79: World2Camera, Camera2Pixel; behind the
// camera or outside InFrame(px, 20) -> no candidate).  lane = feature.
struct AdoptArgs {
    const int32_t *trk_n; int cells, w, h;
    double *pair_T; const double *sa_out;
    const double *trk_px, *trk_depth; const uint8_t *trk_has_mp; double *fdp_px; uint8_t *fdp_cand;
    double fx, fy, cx, cy;
};
__global__ __launch_bounds__(256) void k_track_adopt_pose(AdoptArgs A)
{
    const int p = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) for (int k = 0; k < 7; ++k) A.pair_T[14 * (size_t)p + 7 + k] = A.sa_out[16 * (size_t)p + k];
    if (i >= A.trk_n[p]) return;
    const size_t d = (size_t)p * A.cells + i;
    const double x = A.trk_px[2 * d], y = A.trk_px[2 * d + 1], dep = A.trk_depth[d];
    bool cand = false;
    double ox = x, oy = y;
    if (dep > 0 && A.trk_has_mp[d]) {                 // candidates come from map points (LocalMapping.cpp:52-56: the loop runs over the local map's points that are not _bad)
        Se3 Tr, Tc, Tri;
        for (int k = 0; k < 4; ++k) { Tr.q[k] = A.pair_T[14 * (size_t)p + k]; Tc.q[k] = A.sa_out[16 * (size_t)p + k]; }
        for (int k = 0; k < 3; ++k) { Tr.t[k] = A.pair_T[14 * (size_t)p + 4 + k]; Tc.t[k] = A.sa_out[16 * (size_t)p + 4 + k]; }
        se3_inv_d(&Tr, &Tri);
        const double pr[3] = { (x - A.cx) * dep / A.fx, (y - A.cy) * dep / A.fy, dep };          // Pixel2Camera, Camera.h:56-62
        double pw[3], pc[3];
        se3_act_d(&Tri, pr, pw);                      // the feature's map point (world)
        se3_act_d(&Tc, pw, pc);                       // World2Camera(_pos_world, current->_TCW)
        ox = A.fx * pc[0] / pc[2] + A.cx; oy = A.fy * pc[1] / pc[2] + A.cy;
        cand = !(pc[2] < 0) && ox >= 20 && ox < A.w - 20 && oy >= 20 && oy < A.h - 20;
    }
    A.fdp_px[2 * d] = ox; A.fdp_px[2 * d + 1] = oy;
    A.fdp_cand[d] = (uint8_t)cand;
}

// inputs of ba::OptimizeCurrentPoseOnly for every pair: pose = [t; so3.log()] of the pair's current pose (BA.cpp:190-193), the
// map point of every reference feature, and the mask of the features ProjectMapPoints created (direct-projection successes)
struct PoPrepArgs {
    const int32_t *trk_n; int cells;
    const double *pair_T, *trk_px, *trk_depth; const uint8_t *fdp_ok;
    double *po_pw, *po_pose, *po_depth;
    double fx, fy, cx, cy;
};
__global__ __launch_bounds__(256) void k_track_po_prep(PoPrepArgs A)
{
    const int p = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) {
        double th, w[3];
        so3_log_d(A.pair_T + 14 * (size_t)p + 7, w, &th);
        for (int k = 0; k < 3; ++k) { A.po_pose[6 * (size_t)p + k] = A.pair_T[14 * (size_t)p + 11 + k]; A.po_pose[6 * (size_t)p + 3 + k] = w[k]; }
    }
    if (i >= A.trk_n[p]) return;
    const size_t d = (size_t)p * A.cells + i;
    const double x = A.trk_px[2 * d], y = A.trk_px[2 * d + 1], dep = A.trk_depth[d];
    Se3 Tr, Tri;
    for (int k = 0; k < 4; ++k) Tr.q[k] = A.pair_T[14 * (size_t)p + k];
    for (int k = 0; k < 3; ++k) Tr.t[k] = A.pair_T[14 * (size_t)p + 4 + k];
    se3_inv_d(&Tr, &Tri);
    const double pr[3] = { (x - A.cx) * dep / A.fx, (y - A.cy) * dep / A.fy, dep };
    double pw[3];
    se3_act_d(&Tri, pr, pw);
    A.po_pw[3 * d] = pw[0]; A.po_pw[3 * d + 1] = pw[1]; A.po_pw[3 * d + 2] = pw[2];
    A.po_depth[d] = 0.0;
}

static int launch_load(ygz_hip_ctx *ctx, int predict)
{
    LoadArgs A;
    A.pair_t = ctx->pair_t; A.n_kp = ctx->n_kp; A.cells = ctx->cells;
    A.kp_px = ctx->kp_px; A.kp_level = ctx->kp_level; A.kp_depth = ctx->kp_depth; A.kp_has_mp = ctx->kp_has_mp;
    A.pair_T = ctx->pair_T;
    A.trk_n = ctx->trk_n; A.trk_px = ctx->trk_px; A.trk_level = ctx->trk_level; A.trk_depth = ctx->trk_depth; A.trk_has_mp = ctx->trk_has_mp;
    A.klt_pts = ctx->klt_pts; A.fdp_px = ctx->fdp_px; A.sa_out = ctx->sa_out; A.fdp_cand = ctx->fdp_cand;
    A.fx = ctx->prm.fx; A.fy = ctx->prm.fy; A.cx = ctx->prm.cx; A.cy = ctx->prm.cy; A.predict = predict;
    YGZ_LAUNCH(ctx, KID_TRACK_LOAD, k_track_load, dim3(ygz_div_up(ctx->cells, 256), ctx->n_pairs), dim3(256), A);
    YGZ_HIPCHK(ctx, hipGetLastError());
    return YGZ_OK;
}

extern "C" {

int ygz_hip_set_keypoint_depths(ygz_hip_ctx *ctx, int slot, const double *depth, const uint8_t *has_mappoint, int n)
{
    YgzDeviceGuard dg_(ctx);
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || slot < 0 || slot >= ctx->prm.max_frames || n < 0 || n > ctx->cells || (n > 0 && (!depth || !has_mappoint))) return YGZ_E_INVALID;
    int rc = ygz_track_ensure(ctx);
    if (rc != YGZ_OK) return rc;
    if (n > 0) {
        YGZ_HIPCHK(ctx, hipMemcpyAsync(ctx->kp_depth + (size_t)slot * ctx->cells, depth, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
        YGZ_HIPCHK(ctx, hipMemcpyAsync(ctx->kp_has_mp + (size_t)slot * ctx->cells, has_mappoint, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return YGZ_OK;
}

int ygz_hip_track_begin(ygz_hip_ctx *ctx, const int32_t *cur_slot, const int32_t *ref_slot, const double *T_cur,
                        const double *T_ref, int n_pairs, int predict)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || !cur_slot || !ref_slot || !T_cur || !T_ref) return YGZ_E_INVALID;
    int rc = ygz_track_set_pairs(ctx, cur_slot, ref_slot, T_cur, T_ref, n_pairs);
    if (rc != YGZ_OK) return rc;
    return launch_load(ctx, predict);
}

int ygz_hip_track_reload(ygz_hip_ctx *ctx, int predict)
{
    YgzDeviceGuard dg_(ctx);
    // the track sets are built from pixel / level / depth of the keypoints: neither a BA build nor descriptors still being computed
    // on the matcher's stream (describe_aside) are read or written here
    if (ctx) { int rj_ = ygz_join(ctx, (1u << YGZ_AUX_BA) | (ctx->describe_aside && !ctx->match_aux_reads_track ? 1u << YGZ_AUX_MATCH : 0u)); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || ctx->n_pairs < 1 || !ctx->trk_alloc) return YGZ_E_STATE;
    int rc = launch_load(ctx, predict);
    if (rc != YGZ_OK) return rc;
    return rc;
}

int ygz_hip_track_klt(ygz_hip_ctx *ctx, const ygz_klt_params *prm)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || !prm) return YGZ_E_INVALID;
    if (ctx->n_pairs < 1 || !ctx->trk_alloc) return YGZ_E_STATE;
    return ygz_launch_klt(ctx, ctx->n_pairs, prm);
}

// The reflect-framed copies and Scharr images LK works on depend on the pyramids only: a pipeline that calls this right after
// ygz_hip_build_pyramid has them built on a side stream beside the extractor (memory-bound work beside VALU/LDS-bound work), and
// ygz_hip_track_klt then starts with the LK kernel itself.  Without it ygz_hip_track_klt builds them first; same results.
int ygz_hip_track_klt_prepare(ygz_hip_ctx *ctx)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx) return YGZ_E_INVALID;
    if (ctx->n_pairs < 1 || !ctx->trk_alloc) return YGZ_E_STATE;
    if (!ctx->ev_prep) YGZ_HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_prep, hipEventDisableTiming));
    YgzAuxScope aux(ctx, YGZ_AUX_BA);                        // idle at this point of a step: the BA build is issued after the extractor
    const int rc = ygz_klt_prepare_early(ctx);
    if (rc != YGZ_OK) return rc;
    YGZ_HIPCHK(ctx, hipEventRecord(ctx->ev_prep, ctx->stream));
    ctx->klt_prep_pending = true;
    return YGZ_OK;
}

int ygz_hip_track_direct(ygz_hip_ctx *ctx)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx) return YGZ_E_INVALID;
    if (ctx->n_pairs < 1 || !ctx->trk_alloc) return YGZ_E_STATE;
    YgzAuxScope aux(ctx, YGZ_AUX_MATCH);                     // independent of LK: shares the matcher's side stream
    ctx->match_aux_reads_track = aux.active;
    return ygz_launch_fdp(ctx, ctx->n_pairs);
}

int ygz_hip_track_sparse_align(ygz_hip_ctx *ctx, int max_level, int min_level, int n_iter)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || min_level < 0 || max_level < min_level || max_level >= ctx->prm.pyramid_levels || n_iter < 0) return YGZ_E_INVALID;
    if (ctx->n_pairs < 1 || !ctx->trk_alloc) return YGZ_E_STATE;
    YgzAuxScope aux(ctx, 0);
    return ygz_launch_sparse_align(ctx, ctx->n_pairs, max_level, min_level, n_iter);
}

int ygz_hip_track_adopt_pose(ygz_hip_ctx *ctx)
{
    YgzDeviceGuard dg_(ctx);
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }           // waits for the sparse alignment (and a running direct projection)
    if (!ctx) return YGZ_E_INVALID;
    if (ctx->n_pairs < 1 || !ctx->trk_alloc) return YGZ_E_STATE;
    AdoptArgs A;
    A.trk_n = ctx->trk_n; A.cells = ctx->cells; A.w = ctx->lw[0]; A.h = ctx->lh[0];
    A.pair_T = ctx->pair_T; A.sa_out = ctx->sa_out; A.trk_px = ctx->trk_px; A.trk_depth = ctx->trk_depth; A.trk_has_mp = ctx->trk_has_mp;
    A.fdp_px = ctx->fdp_px; A.fdp_cand = ctx->fdp_cand;
    A.fx = ctx->prm.fx; A.fy = ctx->prm.fy; A.cx = ctx->prm.cx; A.cy = ctx->prm.cy;
    YGZ_LAUNCH(ctx, KID_TRACK_AUX, k_track_adopt_pose, dim3(ygz_div_up(ctx->cells, 256), ctx->n_pairs), dim3(256), A);
    YGZ_HIPCHK(ctx, hipGetLastError());
    return YGZ_OK;
}

int ygz_hip_track_pose_only(ygz_hip_ctx *ctx)
{
    YgzDeviceGuard dg_(ctx);
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }           // the direct projection runs on a side stream
    if (!ctx) return YGZ_E_INVALID;
    if (ctx->n_pairs < 1 || !ctx->trk_alloc) return YGZ_E_STATE;
    PoPrepArgs P;
    P.trk_n = ctx->trk_n; P.cells = ctx->cells; P.pair_T = ctx->pair_T; P.trk_px = ctx->trk_px; P.trk_depth = ctx->trk_depth;
    P.fdp_ok = ctx->fdp_ok; P.po_pw = ctx->po_pw; P.po_pose = ctx->po_pose; P.po_depth = ctx->po_depth;
    P.fx = ctx->prm.fx; P.fy = ctx->prm.fy; P.cx = ctx->prm.cx; P.cy = ctx->prm.cy;
    YGZ_LAUNCH(ctx, KID_TRACK_AUX, k_track_po_prep, dim3(ygz_div_up(ctx->cells, 256), ctx->n_pairs), dim3(256), P);
    YGZ_HIPCHK(ctx, hipGetLastError());
    YgzPoDev d;
    d.off = nullptr; d.cnt = ctx->trk_n; d.stride = ctx->cells; d.use = ctx->fdp_ok;
    d.px = ctx->fdp_px; d.pw = ctx->po_pw; d.poses = ctx->po_pose; d.bad = ctx->po_bad; d.depth = ctx->po_depth;
    d.inliers = ctx->po_cnt; d.rounds = ctx->po_cnt + ctx->prm.max_frames;
    d.T_out = ctx->po_T;
    return ygz_launch_pose_only(ctx, ctx->n_pairs, d);
}

static int pair_count(ygz_hip_ctx *ctx, int pair, int *n)
{
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (pair < 0 || pair >= ctx->n_pairs || !ctx->trk_alloc) return YGZ_E_INVALID;
    YGZ_HIPCHK(ctx, hipMemcpyAsync(n, ctx->trk_n + pair, 4, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return YGZ_OK;
}

int ygz_hip_track_get_klt(ygz_hip_ctx *ctx, int pair, float *pts, uint8_t *status, float *err, int capacity, int *n_out)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || !n_out) return YGZ_E_INVALID;
    int n = 0, rc = pair_count(ctx, pair, &n);
    if (rc != YGZ_OK) return rc;
    *n_out = n;
    if (n > capacity) return YGZ_E_CAPACITY;
    const size_t o = (size_t)pair * ctx->cells;
    if (n > 0) {
        if (pts) YGZ_HIPCHK(ctx, hipMemcpyAsync(pts, ctx->klt_pts + 2 * o, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (status) YGZ_HIPCHK(ctx, hipMemcpyAsync(status, ctx->klt_status + o, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        if (err) YGZ_HIPCHK(ctx, hipMemcpyAsync(err, ctx->klt_err + o, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
        YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return YGZ_OK;
}

int ygz_hip_track_get_direct(ygz_hip_ctx *ctx, int pair, double *px, int32_t *level, uint8_t *ok, int capacity, int *n_out)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || !n_out) return YGZ_E_INVALID;
    int n = 0, rc = pair_count(ctx, pair, &n);
    if (rc != YGZ_OK) return rc;
    *n_out = n;
    if (n > capacity) return YGZ_E_CAPACITY;
    const size_t o = (size_t)pair * ctx->cells;
    if (n > 0) {
        if (px) YGZ_HIPCHK(ctx, hipMemcpyAsync(px, ctx->fdp_px + 2 * o, (size_t)n * 16, hipMemcpyDeviceToHost, ctx->stream));
        if (level) YGZ_HIPCHK(ctx, hipMemcpyAsync(level, ctx->fdp_level + o, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (ok) YGZ_HIPCHK(ctx, hipMemcpyAsync(ok, ctx->fdp_ok + o, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return YGZ_OK;
}

int ygz_hip_track_get_pose_only(ygz_hip_ctx *ctx, int pair, double pose[6], double T[7], int *inliers, int *rounds, uint8_t *bad,
                                double *depth, int capacity, int *n_out)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || !n_out) return YGZ_E_INVALID;
    int n = 0, rc = pair_count(ctx, pair, &n);
    if (rc != YGZ_OK) return rc;
    *n_out = n;
    if (n > capacity && (bad || depth)) return YGZ_E_CAPACITY;
    double h[6], hT[7]; int32_t c[2];
    YGZ_HIPCHK(ctx, hipMemcpyAsync(h, ctx->po_pose + 6 * (size_t)pair, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(hT, ctx->po_T + 7 * (size_t)pair, sizeof(hT), hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(&c[0], ctx->po_cnt + pair, 4, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(&c[1], ctx->po_cnt + ctx->prm.max_frames + pair, 4, hipMemcpyDeviceToHost, ctx->stream));
    const size_t o = (size_t)pair * ctx->cells;
    if (n > 0 && bad) YGZ_HIPCHK(ctx, hipMemcpyAsync(bad, ctx->po_bad + o, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    if (n > 0 && depth) YGZ_HIPCHK(ctx, hipMemcpyAsync(depth, ctx->po_depth + o, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (pose) for (int k = 0; k < 6; ++k) pose[k] = h[k];
    if (T) for (int k = 0; k < 7; ++k) T[k] = hT[k];          // SE3(SO3::exp(pose.tail<3>()), pose.head<3>()), BA.cpp:254, formed on the device
    if (inliers) *inliers = c[0];
    if (rounds) *rounds = c[1];
    return YGZ_OK;
}

int ygz_hip_track_get_pose(ygz_hip_ctx *ctx, int pair, double T[7], int *n_meas, int *iters)
{
    YgzDeviceGuard dg_(ctx);
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || pair < 0 || pair >= ctx->n_pairs || !ctx->trk_alloc) return YGZ_E_INVALID;
    double h[16];
    YGZ_HIPCHK(ctx, hipMemcpyAsync(h, ctx->sa_out + 16 * (size_t)pair, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (T) for (int k = 0; k < 7; ++k) T[k] = h[k];
    if (n_meas) *n_meas = (int)(h[7] / 16);                  // run() returns n_meas_/patch_area_ (SparseImageAlign.cpp:49)
    if (iters) for (int l = 0; l < ctx->prm.pyramid_levels && l < 8; ++l) iters[l] = (int)h[8 + l];
    return YGZ_OK;
}

}  // extern "C"
