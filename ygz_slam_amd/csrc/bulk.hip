// Bulk host<->device traffic for pipelines that stream frames through the context (offline run, bench.py --stream):
// one copy per batch instead of one per frame, asynchronous on the context's stream when the host buffers are page-locked
// (ygz_hip_pinned_alloc), and a per-pair summary of the tracking results reduced on the device so that a step's
// algorithmic output leaves HBM in two copies.  Nothing here computes anything the per-frame entry points do not.
#include "ygz_internal.h"
#include "se3_dev.h"

#define SUM_DOUBLES 32

// per pair: [0..6] sparse-alignment pose, [7] n_meas / 16, [8..13] pose-only pose [t; log so3], [14] inliers, [15] rounds,
// [16] matches, [17] good matches, [18] min_dis, [19] KLT status == 1, [20] direct-projection successes, [21] reference features,
// [22] query keypoints, [23] 0, [24..30] the pose-only pose as quaternion + translation, [31] 0
struct SumArgs {
    const int32_t *pair_q, *n_kp, *trk_n; int cells, max_frames;
    const double *sa_out, *po_pose, *po_T; const int32_t *po_cnt;
    const int32_t *m_idx; const uint8_t *m_good; const int32_t *m_good_n; const double *m_min_dis;
    const uint8_t *klt_status, *fdp_ok;
    double *out;
    int have_po, have_pf;
};

__global__ __launch_bounds__(256) void k_track_summary(SumArgs A)
{
    __shared__ int red[3][4];
    const int p = blockIdx.x, tid = threadIdx.x;
    const int nq = A.n_kp[A.pair_q[p]], nr = A.trk_n[p];
    const size_t o = (size_t)p * A.cells;
    int c_match = 0, c_klt = 0, c_fdp = 0;
    for (int i = tid; i < nq; i += 256) c_match += A.m_idx[o + i] >= 0;
    for (int i = tid; i < nr; i += 256) { c_klt += A.klt_status[o + i] != 0; c_fdp += A.fdp_ok[o + i] != 0; }
    c_match = ygz_wave_sum_i(c_match); c_klt = ygz_wave_sum_i(c_klt); c_fdp = ygz_wave_sum_i(c_fdp);
    if ((tid & 63) == 0) { red[0][tid >> 6] = c_match; red[1][tid >> 6] = c_klt; red[2][tid >> 6] = c_fdp; }
    __syncthreads();
    if (tid == 0) {
        double *d = A.out + (size_t)p * SUM_DOUBLES;
        for (int k = 0; k < 7; ++k) d[k] = A.sa_out[16 * (size_t)p + k];
        d[7] = (double)((int)(A.sa_out[16 * (size_t)p + 7] / 16));
        for (int k = 0; k < 6; ++k) d[8 + k] = A.have_po ? A.po_pose[6 * (size_t)p + k] : 0.0;
        d[14] = A.have_po ? (double)A.po_cnt[p] : 0.0;
        d[15] = A.have_po ? (double)A.po_cnt[A.max_frames + p] : 0.0;
        d[16] = (double)(red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        d[17] = A.have_pf ? (double)A.m_good_n[p] : 0.0;
        d[18] = A.have_pf ? A.m_min_dis[p] : 0.0;
        d[19] = (double)(red[1][0] + red[1][1] + red[1][2] + red[1][3]);
        d[20] = (double)(red[2][0] + red[2][1] + red[2][2] + red[2][3]);
        d[21] = (double)nr; d[22] = (double)nq; d[23] = 0.0;
        for (int k = 0; k < 7; ++k) d[24 + k] = A.have_po ? A.po_T[7 * (size_t)p + k] : 0.0;
        d[31] = 0.0;
    }
}


// Feature::_depth of the keypoints of a slot sampled from the slot's depth image (the RGB-D style input of the offline run; in the
// reference the depth of a feature is the z of its map point in the camera frame, Feature.h:30 -- the depth image stands in for the
// map).  dw x dh samples cover the level-0 frame: pixel (x, y) reads sample ((int)x * dw / w, (int)y * dh / h).  kind 0: float32 metres,
// kind 1: uint16 with depth = value * scale (TUM RGB-D: scale = 1 / 5000), kind 2: float64 metres.  depth <= 0 or NaN = no map point.
struct DepthArgs {
    const void *img; int dw, dh, kind; double scale; int w, h, cells, slot_begin;
    const int32_t *n_kp; const double *kp_px; double *kp_depth; uint8_t *kp_has_mp;
};
__global__ __launch_bounds__(256) void k_kp_depth_from_image(DepthArgs A)
{
    const int slot = A.slot_begin + blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n_kp[slot]) return;
    const size_t o = (size_t)slot * A.cells + i;
    int ix = (int)(((long long)(int)A.kp_px[2 * o] * A.dw) / A.w), iy = (int)(((long long)(int)A.kp_px[2 * o + 1] * A.dh) / A.h);
    ix = ix < 0 ? 0 : (ix >= A.dw ? A.dw - 1 : ix); iy = iy < 0 ? 0 : (iy >= A.dh ? A.dh - 1 : iy);
    const size_t p = ((size_t)slot * A.dh + iy) * A.dw + ix;
    double d;
    if (A.kind == 1) d = (double)reinterpret_cast<const uint16_t *>(A.img)[p] * A.scale;
    else if (A.kind == 2) d = reinterpret_cast<const double *>(A.img)[p];
    else d = (double)reinterpret_cast<const float *>(A.img)[p];
    if (!(d > 0)) d = 0.0;
    A.kp_depth[o] = d; A.kp_has_mp[o] = (uint8_t)(d > 0);
}

extern "C" {

int ygz_hip_pinned_alloc(void **out, size_t bytes)
{
    if (!out || bytes == 0) return YGZ_E_INVALID;
    *out = nullptr;
    return hipHostMalloc(out, bytes, hipHostMallocDefault) == hipSuccess ? YGZ_OK : YGZ_E_HIP;
}

int ygz_hip_pinned_free(void *p)
{
    if (!p) return YGZ_OK;
    return hipHostFree(p) == hipSuccess ? YGZ_OK : YGZ_E_HIP;
}

int ygz_hip_upload_bgr_batch(ygz_hip_ctx *ctx, int slot_begin, int n_slots, const uint8_t *bgr, int wait)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || !bgr || slot_begin < 0 || n_slots < 1 || slot_begin + n_slots > ctx->prm.max_frames) return YGZ_E_INVALID;
    { int rj = ygz_join(ctx); if (rj != YGZ_OK) return rj; }
    const size_t fb = (size_t)ctx->lw[0] * ctx->lh[0] * 3;
    if (!ctx->bgr) YGZ_HIPCHK(ctx, hipMalloc((void **)&ctx->bgr, (size_t)ctx->prm.max_frames * fb + 64));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(ctx->bgr + (size_t)slot_begin * fb, bgr, (size_t)n_slots * fb, hipMemcpyHostToDevice, ctx->stream));
    if (wait) YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int s = slot_begin; s < slot_begin + n_slots; ++s) { ctx->pyr_valid[s] = 0; ctx->pad_levels[s] = 0; }
    return YGZ_OK;
}

int ygz_hip_upload_gray_batch(ygz_hip_ctx *ctx, int slot_begin, int n_slots, const uint8_t *gray, int wait)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || !gray || slot_begin < 0 || n_slots < 1 || slot_begin + n_slots > ctx->prm.max_frames) return YGZ_E_INVALID;
    { int rj = ygz_join(ctx); if (rj != YGZ_OK) return rj; }
    const size_t fb = (size_t)ctx->lw[0] * ctx->lh[0];
    YGZ_HIPCHK(ctx, hipMemcpyAsync(ctx->lvl[0] + (size_t)slot_begin * fb, gray, (size_t)n_slots * fb, hipMemcpyHostToDevice, ctx->stream));
    if (wait) YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int s = slot_begin; s < slot_begin + n_slots; ++s) { ctx->pyr_valid[s] = 0; ctx->pad_levels[s] = 0; }
    return YGZ_OK;
}

int ygz_hip_get_keypoint_pixels_batch(ygz_hip_ctx *ctx, int slot_begin, int n_slots, double *px, int32_t *count, int wait)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || !px || !count || slot_begin < 0 || n_slots < 1 || slot_begin + n_slots > ctx->prm.max_frames) return YGZ_E_INVALID;
    { int rj = ygz_join(ctx); if (rj != YGZ_OK) return rj; }
    const size_t Cn = (size_t)ctx->cells;
    YGZ_HIPCHK(ctx, hipMemcpyAsync(px, ctx->kp_px + (size_t)slot_begin * Cn * 2, (size_t)n_slots * Cn * 16, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(count, ctx->n_kp + slot_begin, (size_t)n_slots * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (wait) YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return YGZ_OK;
}

int ygz_hip_get_keypoints_batch(ygz_hip_ctx *ctx, int slot_begin, int n_slots, ygz_kpt_soa *out, int32_t *count, int wait)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || !out || !count || slot_begin < 0 || n_slots < 1 || slot_begin + n_slots > ctx->prm.max_frames) return YGZ_E_INVALID;
    { int rj = ygz_join(ctx); if (rj != YGZ_OK) return rj; }
    const size_t Cn = (size_t)ctx->cells, o = (size_t)slot_begin * Cn, N = (size_t)n_slots * Cn;
    if (out->px) YGZ_HIPCHK(ctx, hipMemcpyAsync(out->px, ctx->kp_px + o * 2, N * 16, hipMemcpyDeviceToHost, ctx->stream));
    if (out->level) YGZ_HIPCHK(ctx, hipMemcpyAsync(out->level, ctx->kp_level + o, N * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (out->score) YGZ_HIPCHK(ctx, hipMemcpyAsync(out->score, ctx->kp_score + o, N * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (out->angle) YGZ_HIPCHK(ctx, hipMemcpyAsync(out->angle, ctx->kp_angle + o, N * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (out->desc) YGZ_HIPCHK(ctx, hipMemcpyAsync(out->desc, ctx->kp_desc + o * 8, N * 32, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(count, ctx->n_kp + slot_begin, (size_t)n_slots * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (wait) YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return YGZ_OK;
}

int ygz_hip_set_keypoint_depths_batch(ygz_hip_ctx *ctx, int slot_begin, int n_slots, const double *depth, const uint8_t *has_mappoint, int wait)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || !depth || !has_mappoint || slot_begin < 0 || n_slots < 1 || slot_begin + n_slots > ctx->prm.max_frames) return YGZ_E_INVALID;
    { int rj = ygz_join(ctx); if (rj != YGZ_OK) return rj; }
    int rc = ygz_track_ensure(ctx);
    if (rc != YGZ_OK) return rc;
    const size_t Cn = (size_t)ctx->cells, o = (size_t)slot_begin * Cn, N = (size_t)n_slots * Cn;
    YGZ_HIPCHK(ctx, hipMemcpyAsync(ctx->kp_depth + o, depth, N * 8, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipMemcpyAsync(ctx->kp_has_mp + o, has_mappoint, N, hipMemcpyHostToDevice, ctx->stream));
    if (wait) YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return YGZ_OK;
}

int ygz_hip_upload_depth_batch(ygz_hip_ctx *ctx, int slot_begin, int n_slots, const void *depth, int dw, int dh, int kind, double scale, int wait)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || !depth || slot_begin < 0 || n_slots < 1 || slot_begin + n_slots > ctx->prm.max_frames || dw < 1 || dh < 1 ||
        dw > ctx->lw[0] || dh > ctx->lh[0] || kind < 0 || kind > 2 || !(scale > 0)) return YGZ_E_INVALID;
    { int rj = ygz_join(ctx); if (rj != YGZ_OK) return rj; }
    const size_t esz = kind == 1 ? 2 : (kind == 2 ? 8 : 4), fb = (size_t)dw * dh * esz;
    if (ctx->depth_img && (ctx->depth_w != dw || ctx->depth_h != dh || ctx->depth_kind != kind)) {
        YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->depth_img); ctx->depth_img = nullptr;
    }
    if (!ctx->depth_img) {
        YGZ_HIPCHK(ctx, hipMalloc(&ctx->depth_img, (size_t)ctx->prm.max_frames * fb + 64));
        YGZ_HIPCHK(ctx, hipMemsetAsync(ctx->depth_img, 0, (size_t)ctx->prm.max_frames * fb, ctx->stream));
        ctx->depth_w = dw; ctx->depth_h = dh; ctx->depth_kind = kind;
    }
    ctx->depth_scale = scale;
    YGZ_HIPCHK(ctx, hipMemcpyAsync((uint8_t *)ctx->depth_img + (size_t)slot_begin * fb, depth, (size_t)n_slots * fb, hipMemcpyHostToDevice, ctx->stream));
    if (wait) YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return YGZ_OK;
}

int ygz_hip_keypoint_depths_from_image(ygz_hip_ctx *ctx, int slot_begin, int n_slots)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || slot_begin < 0 || n_slots < 1 || slot_begin + n_slots > ctx->prm.max_frames) return YGZ_E_INVALID;
    if (!ctx->depth_img) return YGZ_E_STATE;
    { int rj = ygz_join(ctx); if (rj != YGZ_OK) return rj; }
    int rc = ygz_track_ensure(ctx);
    if (rc != YGZ_OK) return rc;
    DepthArgs A;
    A.img = ctx->depth_img; A.dw = ctx->depth_w; A.dh = ctx->depth_h; A.kind = ctx->depth_kind; A.scale = ctx->depth_scale;
    A.w = ctx->lw[0]; A.h = ctx->lh[0]; A.cells = ctx->cells; A.slot_begin = slot_begin;
    A.n_kp = ctx->n_kp; A.kp_px = ctx->kp_px; A.kp_depth = ctx->kp_depth; A.kp_has_mp = ctx->kp_has_mp;
    YGZ_LAUNCH(ctx, KID_TRACK_AUX, k_kp_depth_from_image, dim3(ygz_div_up(ctx->cells, 256), n_slots), dim3(256), A);
    YGZ_HIPCHK(ctx, hipGetLastError());
    return YGZ_OK;
}

int ygz_hip_get_keypoint_depths(ygz_hip_ctx *ctx, int slot, double *depth, uint8_t *has_mappoint, int capacity, int *n_out)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || !n_out || slot < 0 || slot >= ctx->prm.max_frames) return YGZ_E_INVALID;
    if (!ctx->trk_alloc) return YGZ_E_STATE;
    { int rj = ygz_join(ctx); if (rj != YGZ_OK) return rj; }
    int n = 0;
    YGZ_HIPCHK(ctx, hipMemcpyAsync(&n, ctx->n_kp + slot, 4, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = n;
    if (n > capacity) return YGZ_E_CAPACITY;
    const size_t o = (size_t)slot * ctx->cells;
    if (n > 0) {
        if (depth) YGZ_HIPCHK(ctx, hipMemcpyAsync(depth, ctx->kp_depth + o, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (has_mappoint) YGZ_HIPCHK(ctx, hipMemcpyAsync(has_mappoint, ctx->kp_has_mp + o, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return YGZ_OK;
}

int ygz_hip_get_keypoint_counts(ygz_hip_ctx *ctx, int slot_begin, int n_slots, int32_t *count, int wait)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || !count || slot_begin < 0 || n_slots < 1 || slot_begin + n_slots > ctx->prm.max_frames) return YGZ_E_INVALID;
    YGZ_HIPCHK(ctx, hipMemcpyAsync(count, ctx->n_kp + slot_begin, (size_t)n_slots * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (wait) YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return YGZ_OK;
}

int ygz_hip_track_get_summary(ygz_hip_ctx *ctx, double *out, int capacity_pairs, int *n_pairs, int wait)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || !out || !n_pairs) return YGZ_E_INVALID;
    if (ctx->n_pairs < 1 || !ctx->trk_alloc) return YGZ_E_STATE;
    *n_pairs = ctx->n_pairs;
    if (ctx->n_pairs > capacity_pairs) return YGZ_E_CAPACITY;
    { int rj = ygz_join(ctx); if (rj != YGZ_OK) return rj; }
    double *d = nullptr;
    int rc = ygz_scratch(ctx, SCR_GEN_0 + 2, (size_t)ctx->n_pairs * SUM_DOUBLES * 8, (void **)&d);
    if (rc != YGZ_OK) return rc;
    SumArgs A;
    A.pair_q = ctx->pair_q; A.n_kp = ctx->n_kp; A.trk_n = ctx->trk_n; A.cells = ctx->cells; A.max_frames = ctx->prm.max_frames;
    A.sa_out = ctx->sa_out; A.po_pose = ctx->po_pose; A.po_T = ctx->po_T; A.po_cnt = ctx->po_cnt;
    A.m_idx = ctx->m_idx; A.m_good = ctx->m_good; A.m_good_n = ctx->m_good_n; A.m_min_dis = ctx->m_min_dis;
    A.klt_status = ctx->klt_status; A.fdp_ok = ctx->fdp_ok; A.out = d;
    A.have_po = 1; A.have_pf = (ctx->pf_valid && ctx->m_good) ? 1 : 0;
    YGZ_LAUNCH(ctx, KID_TRACK_AUX, k_track_summary, dim3(ctx->n_pairs), dim3(256), A);
    YGZ_HIPCHK(ctx, hipGetLastError());
    YGZ_HIPCHK(ctx, hipMemcpyAsync(out, d, (size_t)ctx->n_pairs * SUM_DOUBLES * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (wait) YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return YGZ_OK;
}

}  // extern "C"
