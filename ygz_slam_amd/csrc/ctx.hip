// Context, HBM frame store and host<->device plumbing of the C ABI (include/ygz_hip.h).
#include "ygz_internal.h"
#include <string.h>
#include <stdlib.h>
#include <math.h>
#include <new>

extern "C" {

void ygz_hip_default_params(ygz_hip_params *p)
{
    memset(p, 0, sizeof(*p));
    p->image_width = 640; p->image_height = 480;      // config/default.yaml:15-16
    p->pyramid_levels = 3;                            // Basic/Frame.h:23
    p->cell_size = 10; p->fast_threshold = 15;        // default.yaml:50-51
    p->nms_tie_suppress = 0;
    p->max_frames = 8;
    p->fx = 520.9f; p->fy = 521.0f; p->cx = 325.1f; p->cy = 249.7f;   // default.yaml:32-35
    p->debug_maps = 0;
}

const char *ygz_hip_error_string(int code)
{
    switch (code) {
    case YGZ_OK: return "ok";
    case YGZ_E_INVALID: return "invalid argument";
    case YGZ_E_HIP: return "HIP runtime error";
    case YGZ_E_NO_DEVICE: return "no usable gfx950 device";
    case YGZ_E_CAPACITY: return "capacity exceeded";
    case YGZ_E_STATE: return "call order violated";
    default: return "unknown error";
    }
}

int ygz_hip_last_hip_error(const ygz_hip_ctx *ctx) { return ctx ? ctx->last_hip_error : 0; }
int ygz_hip_max_keypoints(const ygz_hip_ctx *ctx) { return ctx ? ctx->cells : 0; }

}  // extern "C"

int ygz_scratch(ygz_hip_ctx *ctx, int id, size_t bytes, void **out)
{
    if (id < 0 || id >= YGZ_N_SCRATCH) return YGZ_E_INVALID;
    if (ctx->scratch_bytes[id] < bytes) {
        if (ctx->scratch[id]) { YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->scratch[id]); ctx->scratch[id] = nullptr; }
        size_t cap = bytes + bytes / 4 + 256;
        YGZ_HIPCHK(ctx, hipMalloc(&ctx->scratch[id], cap));
        ctx->scratch_bytes[id] = cap;
    }
    *out = ctx->scratch[id];
    return YGZ_OK;
}

int ygz_scratch_mirror(ygz_hip_ctx *ctx, int id, void **host)
{
    if (id < 0 || id >= YGZ_N_SCRATCH || !ctx->scratch[id]) return YGZ_E_INVALID;
    if (ctx->scratch_host_bytes[id] < ctx->scratch_bytes[id]) {
        if (ctx->scratch_host[id]) { YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); (void)hipHostFree(ctx->scratch_host[id]); ctx->scratch_host[id] = nullptr; ctx->scratch_host_bytes[id] = 0; }
        YGZ_HIPCHK(ctx, hipHostMalloc(&ctx->scratch_host[id], ctx->scratch_bytes[id], hipHostMallocDefault));
        ctx->scratch_host_bytes[id] = ctx->scratch_bytes[id];
    }
    *host = ctx->scratch_host[id];
    return YGZ_OK;
}

void *ygz_stage(ygz_hip_ctx *ctx, size_t bytes)
{
    bytes = (bytes + 63) & ~(size_t)63;
    if (ctx->stage_used + bytes > ctx->stage_cap) {
        // everything handed out so far must have been consumed before the arena is recycled (or replaced)
        if (ctx->stage && hipStreamSynchronize(ctx->stream) != hipSuccess) return nullptr;
        ctx->stage_used = 0;
        if (bytes > ctx->stage_cap) {
            if (ctx->stage) (void)hipHostFree(ctx->stage);
            ctx->stage = nullptr; ctx->stage_cap = 0;
            const size_t cap = bytes > ((size_t)1 << 20) ? bytes : ((size_t)1 << 20);
            void *p = nullptr;
            const hipError_t e = hipHostMalloc(&p, cap, hipHostMallocDefault);
            if (e != hipSuccess) { ctx->last_hip_error = (int)e; return nullptr; }
            ctx->stage = (uint8_t *)p; ctx->stage_cap = cap;
        }
    }
    uint8_t *r = ctx->stage + ctx->stage_used;
    ctx->stage_used += bytes;
    return r;
}

// ---- packed transfers of single-frame calls (ygz_internal.h)
__global__ __launch_bounds__(256) void k_copy_segs(YgzPackSegs S)
{
    const int seg = blockIdx.y;
    if (seg >= S.n) return;
    const uint32_t bytes = S.bytes[seg];
    const uint8_t *src = S.src[seg]; uint8_t *dst = S.dst[seg];
    const bool dwords = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 3u) == 0;
    for (uint32_t i = (blockIdx.x * 256u + threadIdx.x) * 4u; i < bytes; i += gridDim.x * 1024u) {
        if (dwords && i + 4 <= bytes) *reinterpret_cast<uint32_t *>(dst + i) = *reinterpret_cast<const uint32_t *>(src + i);
        else for (uint32_t k = i; k < bytes && k < i + 4; ++k) dst[k] = src[k];
    }
}
int ygz_pack_begin(ygz_hip_ctx *ctx, YgzPack *pk, size_t capacity_bytes, int scratch_id)
{
    capacity_bytes += 64 * YGZ_PACK_MAX;
    pk->host = (uint8_t *)ygz_stage(ctx, capacity_bytes);
    if (!pk->host) return YGZ_E_HIP;
    void *d = nullptr;
    const int rc = ygz_scratch(ctx, scratch_id, capacity_bytes, &d);
    if (rc != YGZ_OK) return rc;
    pk->dev = (uint8_t *)d; pk->used = 0; pk->cap = capacity_bytes; pk->segs.n = 0;
    return YGZ_OK;
}
void *ygz_pack_add(YgzPack *pk, const void *device_ptr, size_t bytes)
{
    if (pk->segs.n >= YGZ_PACK_MAX || pk->used + bytes > pk->cap) return nullptr;
    const int k = pk->segs.n++;
    pk->segs.src[k] = pk->dev + pk->used;                 // upload: staging -> array (ygz_pack_fetch swaps the roles)
    pk->segs.dst[k] = (uint8_t *)const_cast<void *>(device_ptr);
    pk->segs.bytes[k] = (uint32_t)bytes;
    void *h = pk->host + pk->used;
    pk->used = (pk->used + bytes + 63) & ~(size_t)63;
    return h;
}
static int pack_launch(ygz_hip_ctx *ctx, const YgzPackSegs &S)
{
    uint32_t mx = 0;
    for (int k = 0; k < S.n; ++k) mx = S.bytes[k] > mx ? S.bytes[k] : mx;
    if (S.n == 0 || mx == 0) return YGZ_OK;
    const unsigned gx = (mx + 1023) / 1024 > 64 ? 64 : (mx + 1023) / 1024;
    hipLaunchKernelGGL(k_copy_segs, dim3(gx, S.n), dim3(256), 0, ctx->stream, S);
    YGZ_HIPCHK(ctx, hipGetLastError());
    return YGZ_OK;
}
// Small transfers of the single-frame calls WITHOUT the copy engine (round 6).  The device timeline of a frame of the surface loop
// (profiles/r06_surface_timeline.txt) showed every hipMemcpyAsync of a few tens of KB as an SDMA operation of 7-12 us followed by ~9 us before the
// kernel that waits for it starts -- five to six times per frame.  The page-locked staging memory is mapped into the device's address space
// (hipHostMalloc), so the scatter / gather kernel reads (writes) it directly over PCIe: one kernel instead of copy + hand-over + kernel.
// YGZ_ZERO_COPY=0: the copy-engine form.
bool ygz_zero_copy();
static bool zero_copy() { return ygz_zero_copy(); }
bool ygz_zero_copy() { static const bool z = [] { const char *e = getenv("YGZ_ZERO_COPY"); return !(e && e[0] == '0'); }(); return z; }
int ygz_pack_upload(ygz_hip_ctx *ctx, YgzPack *pk)
{
    if (pk->used == 0) return YGZ_OK;
    if (zero_copy()) {
        YgzPackSegs S = pk->segs;
        for (int k = 0; k < S.n; ++k) S.src[k] = pk->host + (S.src[k] - pk->dev);       // the host slice itself
        return pack_launch(ctx, S);
    }
    YGZ_HIPCHK(ctx, hipMemcpyAsync(pk->dev, pk->host, pk->used, hipMemcpyHostToDevice, ctx->stream));
    return pack_launch(ctx, pk->segs);
}
int ygz_pack_fetch(ygz_hip_ctx *ctx, YgzPack *pk)
{
    if (pk->used == 0) return YGZ_OK;
    YgzPackSegs S = pk->segs;
    const bool z = zero_copy();
    for (int k = 0; k < S.n; ++k) { const uint8_t *stg = z ? pk->host + (S.src[k] - pk->dev) : S.src[k]; S.src[k] = S.dst[k]; S.dst[k] = const_cast<uint8_t *>(stg); }
    const int rc = pack_launch(ctx, S);
    if (rc != YGZ_OK || z) return rc;
    YGZ_HIPCHK(ctx, hipMemcpyAsync(pk->host, pk->dev, pk->used, hipMemcpyDeviceToHost, ctx->stream));
    return YGZ_OK;
}
// one block of bytes between device memory and PAGE-LOCKED host memory (a scratch mirror, the staging arena) on the context's stream: a copy
// kernel for what the single-frame calls move (up to 1 MB), the copy engine beyond
__global__ __launch_bounds__(256) void k_kcopy(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, size_t bytes)
{
    const bool v16 = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0;
    const size_t n16 = v16 ? bytes / 16 : 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        reinterpret_cast<uint4 *>(dst)[i] = reinterpret_cast<const uint4 *>(src)[i];
    for (size_t i = n16 * 16 + (size_t)blockIdx.x * 256 + threadIdx.x; i < bytes; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
int ygz_kcopy(ygz_hip_ctx *ctx, void *dst, const void *src, size_t bytes, int kind)
{
    if (bytes == 0) return YGZ_OK;
    if (!zero_copy() || bytes > ((size_t)1 << 20)) { YGZ_HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, (hipMemcpyKind)kind, ctx->stream)); return YGZ_OK; }
    const unsigned gx = (unsigned)((bytes / 16 + 255) / 256 > 64 ? 64 : (bytes / 16 + 255) / 256 + 1);
    hipLaunchKernelGGL(k_kcopy, dim3(gx), dim3(256), 0, ctx->stream, (uint8_t *)dst, (const uint8_t *)src, bytes);
    YGZ_HIPCHK(ctx, hipGetLastError());
    return YGZ_OK;
}

int ygz_join(ygz_hip_ctx *ctx, unsigned skip_mask)
{
    for (int i = 0; i < 3; ++i)
        if (ctx->aux_pending[i] && !((skip_mask >> i) & 1u)) {
            YGZ_HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join[i], 0)); ctx->aux_pending[i] = false;
            if (i == YGZ_AUX_MATCH) ctx->match_aux_reads_track = false;
        }
    return YGZ_OK;
}

int ygz_ensure_levels(ygz_hip_ctx *ctx, int n_levels)
{
    if (n_levels > YGZ_MAX_LEVELS) return YGZ_E_INVALID;
    for (int L = ctx->n_levels_alloc; L < n_levels; ++L) {
        if (L > 0) { ctx->lw[L] = (ctx->lw[L - 1] + 1) / 2; ctx->lh[L] = (ctx->lh[L - 1] + 1) / 2; }
        // +64 bytes of slack so 4-byte loads of the last pixels stay inside the allocation
        YGZ_HIPCHK(ctx, hipMalloc(&ctx->lvl[L], (size_t)ctx->prm.max_frames * ctx->lw[L] * ctx->lh[L] + 64));
        ctx->n_levels_alloc = L + 1;
    }
    return YGZ_OK;
}

extern "C" {

int ygz_hip_create(ygz_hip_ctx **out, int device, const ygz_hip_params *prm, void *stream)
{
    if (!out || !prm) return YGZ_E_INVALID;
    *out = nullptr;
    if (prm->image_width < 32 || prm->image_height < 32 || prm->image_width > 8192 || prm->image_height > 8192 ||
        prm->pyramid_levels < 1 || prm->pyramid_levels > YGZ_MAX_LEVELS || prm->cell_size < 1 ||
        prm->max_frames < 1 || prm->fast_threshold < 0 || prm->fast_threshold > 254)
        return YGZ_E_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return YGZ_E_NO_DEVICE;
    int prev_dev = -1;
    (void)hipGetDevice(&prev_dev);
    if (hipSetDevice(device) != hipSuccess) return YGZ_E_NO_DEVICE;
    struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore_{ prev_dev == device ? -1 : prev_dev };
    ygz_hip_ctx *ctx = new (std::nothrow) ygz_hip_ctx();
    if (!ctx) return YGZ_E_INVALID;
    ctx->prm = *prm;
    ctx->device = device;
    { int ncu = 0; if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0) ctx->n_cu = ncu; }
    int rc = YGZ_OK;
    do {
        if (stream) { ctx->stream = (hipStream_t)stream; ctx->own_stream = false; }
        else {
            hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
            if (e != hipSuccess) { ctx->last_hip_error = (int)e; rc = YGZ_E_HIP; break; }
            ctx->own_stream = true;
        }
        if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) { rc = YGZ_E_HIP; break; }
        ctx->lw[0] = prm->image_width; ctx->lh[0] = prm->image_height;
        // FeatureDetector ctor / LoadParams: grid = ceil(size / cell)  (FeatureDetector.cpp:301-302,336-337)
        ctx->grid_rows = (int)ceil((double)prm->image_height / prm->cell_size);
        ctx->grid_cols = (int)ceil((double)prm->image_width / prm->cell_size);
        ctx->cells = ctx->grid_rows * ctx->grid_cols;
        // image levels: the frame pyramid plus the extra levels cv::calcOpticalFlowPyrLK builds for the default
        // tracker (21x21 window, maxLevel 4; buildOpticalFlowPyramid stops when the next level <= window)
        int klt_levels = 1;
        { int sw = prm->image_width, sh = prm->image_height;
          for (int level = 0; level <= 4; ++level) { klt_levels = level + 1; sw = (sw + 1) / 2; sh = (sh + 1) / 2; if (sw <= 21 || sh <= 21) break; } }
        if ((rc = ygz_ensure_levels(ctx, prm->pyramid_levels > klt_levels ? prm->pyramid_levels : klt_levels)) != YGZ_OK) break;
        const size_t F = (size_t)prm->max_frames, Cn = (size_t)ctx->cells;
        hipError_t e = hipSuccess;
#define A_(ptr, bytes) if (e == hipSuccess) e = hipMalloc((void **)&(ptr), (bytes))
        A_(ctx->cell_first, F * Cn * 4); A_(ctx->cell_best, F * Cn * 8); A_(ctx->occupied, F * Cn);
        A_(ctx->kp_px, F * Cn * 16); A_(ctx->kp_level, F * Cn * 4); A_(ctx->kp_score, F * Cn * 4);
        A_(ctx->kp_angle, F * Cn * 4); A_(ctx->kp_desc, F * Cn * 32); A_(ctx->n_kp, F * 4);
        A_(ctx->pair_q, F * 4); A_(ctx->pair_t, F * 4);
        A_(ctx->m_tq, F * Cn * 4); A_(ctx->m_td, F * Cn * 4); A_(ctx->m_key, F * Cn * 8);
        A_(ctx->m_idx, F * Cn * 4); A_(ctx->m_dist, F * Cn * 4); A_(ctx->m_dist2, F * Cn * 4);
        if (prm->debug_maps)
            for (int L = 0; L < prm->pyramid_levels; ++L) {
                A_(ctx->dbg_score[L], F * ctx->lw[L] * ctx->lh[L]); A_(ctx->dbg_nms[L], F * ctx->lw[L] * ctx->lh[L]);
            }
#undef A_
        if (e != hipSuccess) { ctx->last_hip_error = (int)e; rc = YGZ_E_HIP; break; }
        if (hipMemsetAsync(ctx->n_kp, 0, F * 4, ctx->stream) != hipSuccess ||
            hipMemsetAsync(ctx->occupied, 0, F * Cn, ctx->stream) != hipSuccess) { rc = YGZ_E_HIP; break; }
        ctx->pyr_valid.assign(F, 0);
        ctx->pad_levels.assign(F, 0);
    } while (0);
    if (rc != YGZ_OK) { ygz_hip_destroy(ctx); return rc; }
    *out = ctx;
    return YGZ_OK;
}

void ygz_hip_ba_free_all(ygz_hip_ctx *ctx);   // ba.hip
void ygz_hip_vocab_free(ygz_hip_ctx *ctx);    // bow.hip

void ygz_hip_destroy(ygz_hip_ctx *ctx)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (int i = 0; i < 3; ++i) if (ctx->aux[i]) (void)hipStreamSynchronize(ctx->aux[i]);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    ygz_hip_ba_free_all(ctx);
    ygz_hip_vocab_free(ctx);
    ygz_kf_store_free(ctx);
    if (ctx->stage) (void)hipHostFree(ctx->stage);
    if (ctx->depth_img) (void)hipFree(ctx->depth_img);
    if (ctx->ev_xctx) (void)hipEventDestroy(ctx->ev_xctx);
    if (ctx->ev_mark) (void)hipEventDestroy(ctx->ev_mark);
    if (ctx->ev_prep) (void)hipEventDestroy(ctx->ev_prep);
    for (int L = 0; L < YGZ_MAX_LEVELS; ++L) {
        if (ctx->lvl[L]) (void)hipFree(ctx->lvl[L]);
        if (ctx->deriv[L]) (void)hipFree(ctx->deriv[L]);
        if (ctx->klt_pad[L]) (void)hipFree(ctx->klt_pad[L]);
        if (L == 0 && ctx->klt_slots) (void)hipFree(ctx->klt_slots);
        if (ctx->dbg_score[L]) (void)hipFree(ctx->dbg_score[L]);
        if (ctx->dbg_nms[L]) (void)hipFree(ctx->dbg_nms[L]);
    }
    void *ptrs[] = { ctx->bgr, ctx->cell_first, ctx->cell_best, ctx->occupied, ctx->kp_px, ctx->kp_level, ctx->kp_score,
                     ctx->kp_angle, ctx->kp_desc, ctx->n_kp, ctx->pair_q, ctx->pair_t, ctx->m_tq, ctx->m_td, ctx->m_key,
                     ctx->m_idx, ctx->m_dist, ctx->m_dist2, ctx->m_good, ctx->m_good_n, ctx->m_min_dis, ctx->trk_n, ctx->trk_px, ctx->trk_level, ctx->trk_depth, ctx->trk_has_mp,
                     ctx->pair_T, ctx->kp_depth, ctx->kp_has_mp, ctx->klt_pts, ctx->klt_err, ctx->klt_status, ctx->fdp_px,
                     ctx->fdp_level, ctx->fdp_ok, ctx->sa_out, ctx->sa_work, ctx->fdp_cand, ctx->po_pw, ctx->po_pose, ctx->po_T, ctx->po_depth, ctx->po_bad, ctx->po_cnt };
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (int i = 0; i < YGZ_N_SCRATCH; ++i) { if (ctx->scratch[i]) (void)hipFree(ctx->scratch[i]); if (ctx->scratch_host[i]) (void)hipHostFree(ctx->scratch_host[i]); }
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    for (int i = 0; i < 3; ++i) { if (ctx->ev_join[i]) (void)hipEventDestroy(ctx->ev_join[i]); if (ctx->aux[i]) (void)hipStreamDestroy(ctx->aux[i]); }
    for (hipEvent_t e : ctx->probe_ev) (void)hipEventDestroy(e);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int ygz_hip_set_overlap(ygz_hip_ctx *ctx, int enable)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx) return YGZ_E_INVALID;
    int rc = ygz_join(ctx);
    if (rc != YGZ_OK) return rc;
    if (enable && !ctx->aux[0]) {
        YGZ_HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        for (int i = 0; i < 3; ++i) {
            YGZ_HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->aux[i], hipStreamNonBlocking));
            YGZ_HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_join[i], hipEventDisableTiming));
        }
    }
    ctx->overlap = enable ? 1 : 0;
    return YGZ_OK;
}

int ygz_hip_synchronize(ygz_hip_ctx *ctx)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx) return YGZ_E_INVALID;
    int rcj = ygz_join(ctx);
    if (rcj != YGZ_OK) return rcj;
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stage_used = 0;                                      // every staged table has been consumed
    return YGZ_OK;
}

int ygz_hip_join(ygz_hip_ctx *ctx)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx) return YGZ_E_INVALID;
    return ygz_join(ctx);
}

// Everything enqueued on `waiter` after this call runs after everything enqueued on `signaler` before it (both contexts on one
// device; no host synchronisation).  The offline run uses it to order the BA context behind the tracking lanes.
int ygz_hip_stream_wait(ygz_hip_ctx *waiter, ygz_hip_ctx *signaler)
{
    if (!waiter || !signaler || waiter->device != signaler->device) return YGZ_E_INVALID;
    if (waiter == signaler) return YGZ_OK;
    YgzDeviceGuard dg_(signaler);
    { int rj = ygz_join(signaler); if (rj != YGZ_OK) return rj; }
    if (!signaler->ev_xctx) YGZ_HIPCHK(signaler, hipEventCreateWithFlags(&signaler->ev_xctx, hipEventDisableTiming));
    YGZ_HIPCHK(signaler, hipEventRecord(signaler->ev_xctx, signaler->stream));
    YGZ_HIPCHK(waiter, hipStreamWaitEvent(waiter->stream, signaler->ev_xctx, 0));
    return YGZ_OK;
}

// ygz_hip_mark remembers the point the context's stream has reached (everything enqueued so far, e.g. an upload); ygz_hip_wait_mark
// orders what `waiter` enqueues from now on behind the last mark of `signaler` -- but not behind what signaler enqueued after it.  The
// offline run chains the uploads of its lanes this way: copies queue up first-in-first-out at full PCIe rate instead of sharing it,
// so the kernels of chunk k start while chunk k + 1 is still crossing the link.
int ygz_hip_mark(ygz_hip_ctx *ctx)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx) return YGZ_E_INVALID;
    if (!ctx->ev_mark) YGZ_HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_mark, hipEventDisableTiming));
    YGZ_HIPCHK(ctx, hipEventRecord(ctx->ev_mark, ctx->stream));
    return YGZ_OK;
}

int ygz_hip_wait_mark(ygz_hip_ctx *waiter, ygz_hip_ctx *signaler)
{
    if (!waiter || !signaler || waiter->device != signaler->device) return YGZ_E_INVALID;
    if (waiter == signaler || !signaler->ev_mark) return YGZ_OK;           // nothing marked yet
    YgzDeviceGuard dg_(waiter);
    YGZ_HIPCHK(waiter, hipStreamWaitEvent(waiter->stream, signaler->ev_mark, 0));
    return YGZ_OK;
}

// ---- plumbing for host code that drives several contexts and a collective library (ygz_slam_amd/host/ygz_offline.cpp): the context's
// stream as an opaque pointer (what ncclAllGather is enqueued on), device memory for exchange buffers, copies ordered on the stream
int ygz_hip_abi_version(void) { return YGZ_HIP_ABI_VERSION; }

int ygz_hip_get_stream(ygz_hip_ctx *ctx, void **stream)
{
    if (!ctx || !stream) return YGZ_E_INVALID;
    *stream = (void *)ctx->stream;
    return YGZ_OK;
}

int ygz_hip_get_device(const ygz_hip_ctx *ctx, int *device, int *compute_units)
{
    if (!ctx) return YGZ_E_INVALID;
    if (device) *device = ctx->device;
    if (compute_units) *compute_units = ctx->n_cu;
    return YGZ_OK;
}

// hipSetDevice(ctx's device) for the calling thread and left that way: what a collective library expects of the thread that calls it
int ygz_hip_make_current(ygz_hip_ctx *ctx)
{
    if (!ctx) return YGZ_E_INVALID;
    YGZ_HIPCHK(ctx, hipSetDevice(ctx->device));
    return YGZ_OK;
}

int ygz_hip_device_alloc(ygz_hip_ctx *ctx, void **out, size_t bytes)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || !out || bytes == 0) return YGZ_E_INVALID;
    *out = nullptr;
    YGZ_HIPCHK(ctx, hipMalloc(out, bytes));
    YGZ_HIPCHK(ctx, hipMemsetAsync(*out, 0, bytes, ctx->stream));
    return YGZ_OK;
}

int ygz_hip_device_free(ygz_hip_ctx *ctx, void *p)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx) return YGZ_E_INVALID;
    if (!p) return YGZ_OK;
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    YGZ_HIPCHK(ctx, hipFree(p));
    return YGZ_OK;
}

// kind 0: host -> device, 1: device -> host, 2: device -> device; on the context's stream.  wait == 0 needs page-locked host memory that
// stays valid until the next synchronising call.
int ygz_hip_copy(ygz_hip_ctx *ctx, void *dst, const void *src, size_t bytes, int kind, int wait)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || !dst || !src || kind < 0 || kind > 2) return YGZ_E_INVALID;
    if (bytes == 0) return YGZ_OK;
    { int rj = ygz_join(ctx); if (rj != YGZ_OK) return rj; }
    const hipMemcpyKind k = kind == 0 ? hipMemcpyHostToDevice : kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    YGZ_HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, k, ctx->stream));
    if (wait) YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return YGZ_OK;
}

int ygz_hip_set_wait_hook(ygz_hip_ctx *ctx, void (*fn)(void *), void *user)
{
    if (!ctx) return YGZ_E_INVALID;
    ctx->wait_hook = fn; ctx->wait_hook_user = fn ? user : nullptr;
    return YGZ_OK;
}

int ygz_hip_timer_begin(ygz_hip_ctx *ctx)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx) return YGZ_E_INVALID;
    YGZ_HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return YGZ_OK;
}

int ygz_hip_timer_end(ygz_hip_ctx *ctx, float *elapsed_ms)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || !elapsed_ms) return YGZ_E_INVALID;
    { int rj = ygz_join(ctx); if (rj != YGZ_OK) return rj; }
    YGZ_HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    YGZ_HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
    YGZ_HIPCHK(ctx, hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1));
    return YGZ_OK;
}

static const char *const k_kernel_names[KID_COUNT] = {
    "k_bgr2gray", "k_pyr_down", "k_fast_select", "k_compact", "k_describe", "k_hamming_nn", "k_match_finalize", "k_track_load",
    "k_find_direct_projection", "k_align2d", "k_sparse_align", "k_scharr", "k_klt", "k_klt_pad", "k_ba_pose_prep", "k_ba_points", "k_ba_final",
    "k_ba_chi2", "k_pose_only_ba", "k_ba_lm", "k_bow_transform", "k_bow_match", "k_depth_from_triangulation", "k_lmap_match", "k_lmap_aux",
    "k_match_postfilter", "k_track_aux", "k_depth_filter", "k_window" };

int ygz_hip_probe_begin(ygz_hip_ctx *ctx, const char *kernel_name, int max_launches)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || !kernel_name || max_launches < 1 || max_launches > 65536) return YGZ_E_INVALID;
    int id = -1;
    for (int i = 0; i < KID_COUNT; ++i) if (strcmp(kernel_name, k_kernel_names[i]) == 0) id = i;
    if (id < 0) return YGZ_E_INVALID;
    while ((int)ctx->probe_ev.size() < 2 * max_launches) {
        hipEvent_t e;
        YGZ_HIPCHK(ctx, hipEventCreate(&e));
        ctx->probe_ev.push_back(e);
    }
    ctx->probe_used = 0; ctx->probe_id = id;
    return YGZ_OK;
}

int ygz_hip_probe_end(ygz_hip_ctx *ctx, double *total_ms, int *launches)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || !total_ms || !launches) return YGZ_E_INVALID;
    { int rj = ygz_join(ctx); if (rj != YGZ_OK) return rj; }
    for (int i = 0; i < 3; ++i) if (ctx->aux[i]) YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->aux[i]));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    double tot = 0;
    for (int i = 0; i + 1 < ctx->probe_used; i += 2) {
        float ms = 0;
        YGZ_HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->probe_ev[i], ctx->probe_ev[i + 1]));
        tot += ms;
    }
    *total_ms = tot; *launches = ctx->probe_used / 2;
    ctx->probe_id = -1; ctx->probe_used = 0;
    return YGZ_OK;
}

int ygz_hip_level_size(const ygz_hip_ctx *ctx, int level, int *w, int *h)
{
    if (!ctx || level < 0 || level >= ctx->n_levels_alloc) return YGZ_E_INVALID;
    if (w) *w = ctx->lw[level];
    if (h) *h = ctx->lh[level];
    return YGZ_OK;
}

int ygz_hip_upload_bgr(ygz_hip_ctx *ctx, int slot, const uint8_t *bgr, int stride_bytes)
{
    YgzDeviceGuard dg_(ctx);
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || !bgr || slot < 0 || slot >= ctx->prm.max_frames) return YGZ_E_INVALID;
    const int w = ctx->lw[0], h = ctx->lh[0];
    if (stride_bytes < w * 3) return YGZ_E_INVALID;
    if (!ctx->bgr) YGZ_HIPCHK(ctx, hipMalloc((void **)&ctx->bgr, (size_t)ctx->prm.max_frames * w * h * 3 + 64));
    YGZ_HIPCHK(ctx, hipMemcpy2DAsync(ctx->bgr + (size_t)slot * w * h * 3, (size_t)w * 3, bgr, (size_t)stride_bytes,
                                     (size_t)w * 3, (size_t)h, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->pyr_valid[slot] = 0; ctx->pad_levels[slot] = 0;
    return YGZ_OK;
}

int ygz_hip_upload_gray(ygz_hip_ctx *ctx, int slot, const uint8_t *gray, int stride_bytes)
{
    YgzDeviceGuard dg_(ctx);
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || !gray || slot < 0 || slot >= ctx->prm.max_frames) return YGZ_E_INVALID;
    const int w = ctx->lw[0], h = ctx->lh[0];
    if (stride_bytes < w) return YGZ_E_INVALID;
    YGZ_HIPCHK(ctx, hipMemcpy2DAsync(ctx->lvl[0] + (size_t)slot * w * h, (size_t)w, gray, (size_t)stride_bytes,
                                     (size_t)w, (size_t)h, hipMemcpyHostToDevice, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->pyr_valid[slot] = 0; ctx->pad_levels[slot] = 0;
    return YGZ_OK;
}

int ygz_hip_build_pyramid(ygz_hip_ctx *ctx, int slot_begin, int n_slots, int from_bgr)
{
    YgzDeviceGuard dg_(ctx);
    if (!ctx || slot_begin < 0 || n_slots < 1 || slot_begin + n_slots > ctx->prm.max_frames) return YGZ_E_INVALID;
    if (from_bgr && !ctx->bgr) return YGZ_E_STATE;
    { int rj = ygz_join(ctx, 1u << YGZ_AUX_BA); if (rj != YGZ_OK) return rj; }      // a pending BA linearisation reads no image
    if (ctx->klt_prep_pending) { YGZ_HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_prep, 0)); ctx->klt_prep_pending = false; }   // ... but LK working images being built do
    int rc = ygz_launch_gray_pyramid(ctx, slot_begin, n_slots, from_bgr, ctx->n_levels_alloc);
    if (rc != YGZ_OK) return rc;
    for (int s = slot_begin; s < slot_begin + n_slots; ++s) ctx->pyr_valid[s] = 1;
    ctx->klt_prep_valid = false;
    return YGZ_OK;
}

int ygz_hip_download_level(ygz_hip_ctx *ctx, int slot, int level, uint8_t *dst)
{
    YgzDeviceGuard dg_(ctx);
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || !dst || slot < 0 || slot >= ctx->prm.max_frames || level < 0 || level >= ctx->n_levels_alloc) return YGZ_E_INVALID;
    const size_t n = (size_t)ctx->lw[level] * ctx->lh[level];
    YGZ_HIPCHK(ctx, hipMemcpyAsync(dst, ctx->lvl[level] + (size_t)slot * n, n, hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return YGZ_OK;
}

// the tracker's framed copy of a level (KLT_B = 24 pixels of BORDER_REFLECT_101 around it, row pitch (w + 48 + 3) & ~3) as the pyramid
// kernels (or k_klt_pad) left it: dst [h + 48][w + 48].  YGZ_E_STATE when the slot has no current framed copy of that level.
int ygz_hip_download_framed_level(ygz_hip_ctx *ctx, int slot, int level, uint8_t *dst)
{
    YgzDeviceGuard dg_(ctx);
    if (ctx) { int rj_ = ygz_join(ctx); if (rj_ != YGZ_OK) return rj_; }
    if (!ctx || !dst || slot < 0 || slot >= ctx->prm.max_frames || level < 0 || level >= ctx->n_levels_alloc) return YGZ_E_INVALID;
    if (!ctx->klt_pad[level] || ctx->pad_levels[slot] <= level) return YGZ_E_STATE;
    const int w = ctx->lw[level], h = ctx->lh[level], pw = KLT_PW(w), ph = h + 2 * KLT_B;
    YGZ_HIPCHK(ctx, hipMemcpy2DAsync(dst, (size_t)(w + 2 * KLT_B), ctx->klt_pad[level] + (size_t)slot * pw * ph, (size_t)pw, (size_t)(w + 2 * KLT_B), (size_t)ph,
                                     hipMemcpyDeviceToHost, ctx->stream));
    YGZ_HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return YGZ_OK;
}

}  // extern "C"
