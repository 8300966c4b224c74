// Internal declarations shared by the HIP translation units of libygz_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <vector>
#include "../../include/ygz_hip.h"

#define YGZ_KLT_LEVELS 5          // Tracker.cpp:97 maxLevel 4 -> 5 levels
// the tracker's working images: every level inside a reflect-101 frame of KLT_B pixels (klt.hip; written by the pyramid kernels, image.hip)
#define KLT_B      24          // >= window + 1, multiple of 4
#define KLT_PW(w)  (((w) + 2 * KLT_B + 3) & ~3)      // framed row pitch, 4-byte aligned for any level width
#define YGZ_N_SCRATCH  28

struct ygz_hip_ctx {
    ygz_hip_params prm;
    int device = 0;
    int n_cu = 256;                          // compute units of the device (launch-shape decisions)
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int last_hip_error = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    // geometry
    int n_levels_alloc = 0;                 // image levels with storage (>= pyramid_levels)
    int lw[YGZ_MAX_LEVELS] = {0}, lh[YGZ_MAX_LEVELS] = {0};
    int grid_cols = 0, grid_rows = 0, cells = 0;

    // frame store (HBM): level L of slot s at lvl[L] + s*lw[L]*lh[L]
    uint8_t *lvl[YGZ_MAX_LEVELS] = {nullptr};
    uint8_t *bgr = nullptr;                 // [max_frames][h][w][3], allocated on first BGR upload
    std::vector<uint8_t> pyr_valid;         // per slot: pyramid built
    // Scharr derivative levels for KLT (int16 x2 per pixel), allocated on first KLT call
    int16_t *deriv[YGZ_MAX_LEVELS] = {nullptr};
    uint8_t *klt_pad[YGZ_MAX_LEVELS] = {nullptr};       // reflect-101 framed copies of the levels (KLT working images)
    int32_t *klt_slots = nullptr; int n_klt_slots = 0, n_klt_refs = 0;   // distinct slots of the pair table: [0, n_klt_slots) all, then the reference slots
    std::vector<int32_t> klt_slots_host;    // the first n_klt_slots entries of klt_slots
    std::vector<uint8_t> pad_levels;        // per slot: the framed copies of levels [0, pad_levels) match the slot's pyramid (written by the pyramid kernels or by k_klt_pad)

    // extractor state per slot
    uint32_t *cell_first = nullptr;         // [F][cells]  min over candidates of (visit<<1 | isnan)
    unsigned long long *cell_best = nullptr;// [F][cells]  max over non-NaN of (ordered(score)<<32 | ~visit)
    uint8_t  *occupied = nullptr;           // [F][cells]
    double   *kp_px = nullptr;              // [F][cells][2]
    int32_t  *kp_level = nullptr;           // [F][cells]
    float    *kp_score = nullptr, *kp_angle = nullptr;
    uint32_t *kp_desc = nullptr;            // [F][cells][8]
    int32_t  *n_kp = nullptr;               // [F]
    uint8_t  *dbg_score[YGZ_MAX_LEVELS] = {nullptr}, *dbg_nms[YGZ_MAX_LEVELS] = {nullptr};

    // matcher state per pair (capacity max_frames pairs)
    int32_t *pair_q = nullptr, *pair_t = nullptr;   // [F] slot ids
    int32_t *m_tq = nullptr, *m_td = nullptr;       // [F][cells] nearest query of each train row
    unsigned long long *m_key = nullptr;            // [F][cells] (dist<<32 | train) per query
    int32_t *m_idx = nullptr, *m_dist = nullptr, *m_dist2 = nullptr;   // [F][cells] final per query
    int n_pairs = 0;
    uint8_t *m_good = nullptr; int32_t *m_good_n = nullptr; double *m_min_dis = nullptr;   // M3 post-filter per pair (postfilter.hip)
    bool pf_valid = false;

    // resident tracking state: one "track set" per pair (capacity max_frames pairs), filled either from the
    // keypoints of the pair's reference slot (device-to-device) or from host arrays (single-pair APIs)
    bool trk_alloc = false;
    int32_t *trk_n = nullptr;                // [F]
    double  *trk_px = nullptr;               // [F][cells][2]  reference pixels (level 0)
    int32_t *trk_level = nullptr;            // [F][cells]
    double  *trk_depth = nullptr;            // [F][cells]
    uint8_t *trk_has_mp = nullptr;           // [F][cells]
    double  *pair_T = nullptr;               // [F][2][7]  (T_ref, T_cur) of each pair
    double  *kp_depth = nullptr;             // [F][cells] per-slot keypoint depth (Feature::_depth)
    uint8_t *kp_has_mp = nullptr;            // [F][cells] Feature::_mappoint != nullptr
    float   *klt_pts = nullptr;              // [F][cells][2] in/out
    float   *klt_err = nullptr;              // [F][cells]
    uint8_t *klt_status = nullptr;           // [F][cells]
    double  *fdp_px = nullptr;               // [F][cells][2] in/out
    int32_t *fdp_level = nullptr;            // [F][cells]
    uint8_t *fdp_ok = nullptr;               // [F][cells]
    double  *sa_out = nullptr;               // [F][16]: pose 7, n_meas, iters per level
    uint8_t *fdp_cand = nullptr;             // [F][cells] candidate in view (LocalMapping::FindCandidates); 1 unless ygz_hip_track_adopt_pose cleared it
    double  *po_pw = nullptr;                // [F][cells][3] map point of each reference feature (world), pose-only stage
    double  *po_pose = nullptr;              // [F][6] [t; log(so3)] in/out
    double  *po_T = nullptr;                 // [F][7] the same pose as quaternion + translation
    double  *po_depth = nullptr;             // [F][cells]
    uint8_t *po_bad = nullptr;               // [F][cells] 1: not a feature of the current frame, or outlier
    int32_t *po_cnt = nullptr;               // [F][2] inliers, rounds
    uint8_t *sa_work = nullptr;              // [F][sa_work_stride]
    size_t   sa_work_stride = 0;
    int      deriv_slots = 0;                // slots covered by the Scharr buffers
    bool     klt_prep_valid = false;         // the LK working images of the current pair table / pyramids are already built

    // optional stage overlap: independent resident stages run on side streams (forked from / joined to `stream`)
    int overlap = 0;
    hipStream_t aux[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    bool aux_pending[3] = {false, false, false};

    // per-kernel HIP-event probe (bench.py roofline leg): events around every launch of ONE chosen kernel
    int probe_id = -1, probe_used = 0;
    std::vector<hipEvent_t> probe_ev;

    // growable scratch buffers
    void  *scratch[YGZ_N_SCRATCH] = {nullptr};
    size_t scratch_bytes[YGZ_N_SCRATCH] = {0};
    // page-locked host mirrors of scratch buffers (ygz_scratch_mirror): an entry point that moves many small arrays packs them into the
    // mirror at the device offsets and crosses PCIe with ONE copy per direction (the single-frame class-surface calls: a pageable
    // hipMemcpyAsync per array cost 36 staged copies per frame)
    void  *scratch_host[YGZ_N_SCRATCH] = {nullptr};
    size_t scratch_host_bytes[YGZ_N_SCRATCH] = {0};

    // resident BA windows
    struct Vocab;
    Vocab *vocab = nullptr;                  // DBoW3 vocabulary tree + per-keypoint BoW (bow.hip)
    struct BaWindow;
    std::vector<BaWindow*> ba;
    void *ba_table = nullptr;                // device array of per-window descriptors (entries of device-built windows are patched by k_win_edges)
    void *ba_table_host = nullptr;           // host mirror (std::vector<BaDev>*), entries re-uploaded one by one when their window changed
    bool  ba_table_dirty = true;             // some window's entry must be (re)uploaded
    int   ba_max_K = 0, ba_max_P = 0;

    // page-locked staging arena for the small host tables an asynchronous entry point hands to the copy engine (pair tables, poses,
    // window descriptors): a slice stays valid until the next ygz_hip_synchronize of this context, so no entry point has to wait for the
    // stream just because its host arguments are temporaries
    uint8_t *stage = nullptr; size_t stage_cap = 0, stage_used = 0;
    void (*wait_hook)(void *) = nullptr; void *wait_hook_user = nullptr;   // ygz_hip_set_wait_hook: called once between launch and wait of the next single-frame sparse alignment
    int lmap_async_n = 0, lmap_async_k = 0;                   // candidates / keyframes of the pending ygz_hip_find_direct_projection_mp_begin run (0: none)
    int lds_per_block = 0;                                    // hipDeviceAttributeMaxSharedMemoryPerBlock, asked once
    // per-slot depth images (RGB-D style input of the offline run: Feature::_depth of the keypoints is sampled from them on the device)
    void *depth_img = nullptr; int depth_w = 0, depth_h = 0, depth_kind = 0; double depth_scale = 1.0;
    // keyframe store + relative-pose store of the offline run (window.hip)
    struct KfStore;
    KfStore *kfs = nullptr;
    hipEvent_t ev_xctx = nullptr;            // ygz_hip_stream_wait: recorded on this context's stream, waited for by another's
    hipEvent_t ev_mark = nullptr;            // ygz_hip_mark / ygz_hip_wait_mark
    hipEvent_t ev_prep = nullptr;            // end of the LK working images built ahead on a side stream (ygz_hip_track_klt_prepare)
    bool klt_prep_pending = false;           // ... and nobody has waited for it yet
    int  ba_last_path = 0;                   // ygz_hip_ba_last_path
    bool lm_spread = false;                  // resident-LM teams spread over the XCDs instead of one XCD each (ygz_hip_ba_set_team_placement)
    int  lm_light_barrier = -1;              // the same-XCD barrier without the L2 write-back: -1 not tested yet, 0 failed its self-test on this device (full barrier), 1 passed
    int  lm_team_budget = 0;                 // workgroups a resident-LM launch may hold (0: half of the CUs), ygz_hip_ba_set_team_budget
    bool match_aux_reads_track = false;      // a direct projection (reads the track sets) is pending on the matcher's side stream
    bool describe_aside = false;             // ygz_hip_detect leaves the descriptor kernel on the matcher's side stream (YGZ_DESCRIBE_ASIDE=1)
    bool sa_attr_set = false;                // the dynamic-LDS opt-in of k_sparse_align was made on this context's device
    bool lm_attr_set = false;                // the dynamic-LDS opt-in of k_ba_lm_team was made on this context's device
    double *sa_lin = nullptr;             // != nullptr for the launch: [32] chi2 sum, n_meas, H (21, upper triangle), Jres (6) of the LAST linearisation (ygz_hip_sparse_align_residuals)
    double *sa_out_host = nullptr;        // for the launch: page-locked copy of sa_out that the kernel writes itself (the single-frame call: no copy back)
    bool sa_rel = false;                  // the pose in sa_out is T_cur_from_ref itself (no product with the reference pose on the way in or out)
    int sa_n_hint = 0;                    // features of the ONE problem of a single-frame call (0: unknown, the counts are on the device): sizes the LDS tiers
    int  klt_prep_levels = 0;                // levels covered by the LK working images while klt_prep_valid
    // instruction-issue priority (s_setprio 0..3) the latency- / memory-bound kernels raise their wavefronts to, so that they keep
    // their pace when a VALU-bound kernel of another stream shares their SIMDs (bit 0: sparse alignment, bit 1: direct projection,
    // bit 2: BA linearisation, bit 3: LM / ceres / pose-only loops); YGZ_WAVE_PRIO=<mask> overrides
    int  wave_prio_mask = 0;
};

#define YGZ_HIPCHK(ctx, call)                                            \
    do { hipError_t e_ = (call);                                         \
         if (e_ != hipSuccess) { (ctx)->last_hip_error = (int)e_; return YGZ_E_HIP; } } while (0)

static inline int ygz_div_up(int a, int b) { return (a + b - 1) / b; }

// Every entry point that may allocate or launch makes the context's device current for its duration and restores the
// caller's device afterwards (a process may hold contexts on several GPUs, and torch may have switched device since create).
struct YgzDeviceGuard {
    int prev = -1; bool switched = false;
    explicit YgzDeviceGuard(const ygz_hip_ctx *c)
    {
        if (!c) return;
        if (hipGetDevice(&prev) == hipSuccess && prev != c->device) switched = (hipSetDevice(c->device) == hipSuccess);
    }
    ~YgzDeviceGuard() { if (switched) (void)hipSetDevice(prev); }
    YgzDeviceGuard(const YgzDeviceGuard &) = delete;
    YgzDeviceGuard &operator=(const YgzDeviceGuard &) = delete;
};

// kernel ids for the probe
enum { KID_BGR2GRAY = 0, KID_PYR_DOWN, KID_FAST_SELECT, KID_COMPACT, KID_DESCRIBE, KID_HAMMING_NN, KID_MATCH_FINALIZE,
       KID_TRACK_LOAD, KID_FDP, KID_ALIGN2D, KID_SPARSE_ALIGN, KID_SCHARR, KID_KLT, KID_KLT_PAD, KID_BA_POSE_PREP, KID_BA_POINTS,
       KID_BA_POSES, KID_BA_CHI2, KID_POSE_ONLY, KID_BA_LM, KID_BOW_TRANSFORM, KID_BOW_MATCH, KID_DEPTH_TRI, KID_LMAP_MATCH, KID_LMAP_AUX, KID_MATCH_POSTFILTER, KID_TRACK_AUX, KID_DEPTH_FILTER, KID_WINDOW, KID_COUNT };

#define YGZ_LAUNCH(ctx, kid, kern, grid, block, ...)                                                         \
    do { const bool pr_ = (ctx)->probe_id == (kid) && (ctx)->probe_used + 2 <= (int)(ctx)->probe_ev.size();    \
         if (pr_) (void)hipEventRecord((ctx)->probe_ev[(ctx)->probe_used], (ctx)->stream);                     \
         hipLaunchKernelGGL(kern, grid, block, 0, (ctx)->stream, __VA_ARGS__);                                 \
         if (pr_) { (void)hipEventRecord((ctx)->probe_ev[(ctx)->probe_used + 1], (ctx)->stream); (ctx)->probe_used += 2; } } while (0)

// the same with dynamic LDS bytes
#define YGZ_LAUNCH_DYN(ctx, kid, kern, grid, block, dyn, ...)                                                \
    do { const bool pr_ = (ctx)->probe_id == (kid) && (ctx)->probe_used + 2 <= (int)(ctx)->probe_ev.size();    \
         if (pr_) (void)hipEventRecord((ctx)->probe_ev[(ctx)->probe_used], (ctx)->stream);                     \
         hipLaunchKernelGGL(kern, grid, block, dyn, (ctx)->stream, __VA_ARGS__);                               \
         if (pr_) { (void)hipEventRecord((ctx)->probe_ev[(ctx)->probe_used + 1], (ctx)->stream); (ctx)->probe_used += 2; } } while (0)

// scratch ids
enum { SCR_MATCH_Q = 0, SCR_MATCH_T, SCR_ALIGN_IN, SCR_ALIGN_OUT, SCR_SA_IN, SCR_SA_OUT, SCR_SA_WORK,
       SCR_KLT_PTS, SCR_KLT_OUT, SCR_BA_0, SCR_BOW, SCR_LMAP, SCR_WIN, SCR_GEN_0 = 16 };

int ygz_scratch(ygz_hip_ctx *ctx, int id, size_t bytes, void **out);
// page-locked host memory of (at least) the capacity of scratch buffer `id` (call after ygz_scratch); valid until the scratch grows
int ygz_scratch_mirror(ygz_hip_ctx *ctx, int id, void **host);
// `bytes` of page-locked host memory that stay valid until the next ygz_hip_synchronize (nullptr: allocation failed)
void *ygz_stage(ygz_hip_ctx *ctx, size_t bytes);
// Several small arrays of a single-frame call in ONE transfer: the host fills slices of a page-locked block (ygz_pack_add returns the slice of a
// device destination), ygz_pack_upload copies the block to a device staging area and ONE kernel scatters the slices; ygz_pack_fetch is the mirror
// image for results (one gather kernel, one copy back; the caller waits and reads the slices).  A pageable or even page-locked hipMemcpyAsync per
// array is a blit kernel of ~4 us on the stream each: the nine of a single-frame sparse alignment were 36 us in front of a 300 us kernel.
#define YGZ_PACK_MAX 12
struct YgzPackSegs { const uint8_t *src[YGZ_PACK_MAX]; uint8_t *dst[YGZ_PACK_MAX]; uint32_t bytes[YGZ_PACK_MAX]; int n; };
struct YgzPack { uint8_t *host = nullptr, *dev = nullptr; size_t used = 0, cap = 0; YgzPackSegs segs; };
int   ygz_pack_begin(ygz_hip_ctx *ctx, YgzPack *pk, size_t capacity_bytes, int scratch_id);
void *ygz_pack_add(YgzPack *pk, const void *device_ptr, size_t bytes);        // the host slice of that device array (nullptr: full)
int   ygz_pack_upload(ygz_hip_ctx *ctx, YgzPack *pk);                          // host slices -> their device arrays (asynchronous)
int   ygz_pack_fetch(ygz_hip_ctx *ctx, YgzPack *pk);                           // device arrays -> host slices (asynchronous: synchronise before reading)
bool  ygz_zero_copy();                                                         // small transfers by kernels that read / write the page-locked staging memory (default; YGZ_ZERO_COPY=0: the copy engine)
int   ygz_kcopy(ygz_hip_ctx *ctx, void *dst, const void *src, size_t bytes, int kind);   // device <-> page-locked host block on the stream: a copy kernel up to 1 MB (no copy engine), hipMemcpyAsync beyond; kind = hipMemcpyKind
// the brute-force matcher over descriptor sets desc + s * set_stride (u32 units), sizes set_count[s], pairs (pair_q[p], pair_t[p]): device
// arrays; results in ctx->m_idx / m_dist [n_pairs][cells] (hamming.hip)
int ygz_run_match(ygz_hip_ctx *ctx, const uint32_t *desc, size_t set_stride, const int32_t *set_count, const int32_t *pair_q,
                  const int32_t *pair_t, int n_pairs, int max_rows, int cross_check, bool want_second);
void ygz_kf_store_free(ygz_hip_ctx *ctx);   // window.hip
int ygz_join(ygz_hip_ctx *ctx, unsigned skip_mask = 0);   // main stream waits for every pending side-stream stage (bit i of
                                                          // skip_mask: leave side stream i pending -- for entry points that do not
                                                          // touch what that stage reads or writes)
#define YGZ_AUX_SPARSE 0
#define YGZ_AUX_BA     1
#define YGZ_AUX_MATCH  2

// RAII: run the enclosed launches on side stream `idx` (sparse-align 0, BA 1, matcher 2) when overlap is enabled.
// The side stream first waits for everything already enqueued on the main stream (fork), and the main stream waits
// for it at the next ygz_join (any entry point that reads or overwrites shared state, and ygz_hip_synchronize).
struct YgzAuxScope {
    ygz_hip_ctx *c; int idx; hipStream_t saved; bool active;
    YgzAuxScope(ygz_hip_ctx *ctx, int i) : c(ctx), idx(i), saved(ctx->stream), active(false)
    {
        if (!ctx->overlap || !ctx->aux[i]) return;
        if (hipEventRecord(ctx->ev_fork, ctx->stream) != hipSuccess) return;
        if (hipStreamWaitEvent(ctx->aux[i], ctx->ev_fork, 0) != hipSuccess) return;
        ctx->stream = ctx->aux[i]; active = true;
    }
    ~YgzAuxScope()
    {
        if (!active) return;
        (void)hipEventRecord(c->ev_join[idx], c->aux[idx]);
        c->stream = saved; c->aux_pending[idx] = true;
    }
};
int ygz_ensure_levels(ygz_hip_ctx *ctx, int n_levels);      // allocates image levels up to n_levels

// launchers implemented in the kernel translation units
int ygz_launch_gray_pyramid(ygz_hip_ctx *ctx, int slot_begin, int n_slots, int from_bgr, int up_to_level);
int ygz_launch_detect(ygz_hip_ctx *ctx, int slot_begin, int n_slots);
int ygz_launch_describe(ygz_hip_ctx *ctx, int slot_begin, int n_slots);
int ygz_track_ensure(ygz_hip_ctx *ctx);                      // allocates the resident tracking state
struct YgzPack;
int ygz_track_set_pairs(ygz_hip_ctx *ctx, const int32_t *cur_slot, const int32_t *ref_slot, const double *T_cur,
                        const double *T_ref, int n_pairs, YgzPack *pk = nullptr);      // pk: the tables join the caller's packed upload instead of being copied one by one
int ygz_launch_klt(ygz_hip_ctx *ctx, int n_pairs, const ygz_klt_params *prm);
int ygz_klt_prepare_early(ygz_hip_ctx *ctx);
int ygz_launch_fdp(ygz_hip_ctx *ctx, int n_pairs);
int ygz_launch_sparse_align(ygz_hip_ctx *ctx, int n_pairs, int max_level, int min_level, int n_iter);
// ba::OptimizeCurrentPoseOnly on device arrays (pose_only.hip): rows of frame f = [off[f], off[f+1]) or, when cnt != nullptr,
// [f * stride, f * stride + cnt[f]); use (nullable) masks rows that are not features of the frame
struct YgzPoDev {
    const int32_t *off, *cnt; int stride; const uint8_t *use;
    const double *px, *pw; double *poses; uint8_t *bad; double *depth; int32_t *inliers, *rounds;
    double *T_out;            // nullable: [n_frames][7] SE3(SO3::exp(pose.tail<3>()), pose.head<3>()) of the final pose (BA.cpp:254)
};
int ygz_launch_pose_only(ygz_hip_ctx *ctx, int n_frames, const YgzPoDev &d);
int ygz_pf_ensure(ygz_hip_ctx *ctx);
int ygz_launch_match_postfilter(ygz_hip_ctx *ctx, const int32_t *set_count, const int32_t *pair_q, int n_pairs, double lo, double hi, double factor);
// LocalMapping::FindCandidates + ProjectMapPoints for the windows of the keyframe store (align.hip, called by window.hip): for pair p =
// (window pair_w[p], keyframe j = pair_j[p]) and every selected anchor feature s of that window, the map point (the feature's pixel and
// depth in the anchor's camera) is projected with the chained pose Tj[w][j], tested like FindCandidates and refined with
// FindDirectProjection from the anchor's image into the keyframe's; images live in the store's rows.
struct YgzWinProject {
    const uint8_t *rows; size_t row_bytes, off_px, off_depth, off_level, off_img[YGZ_MAX_LEVELS];
    int n_levels, w[YGZ_MAX_LEVELS], h[YGZ_MAX_LEVELS];
    const int32_t *pair_w, *pair_j, *kf_index, *counts;     // [n_pairs], [n_pairs], [n_win][Kcap], the store's per-row counts
    int n_pairs, Kcap, Pcap, set_base;
    const double *Tj;                                        // [n_win][Kcap][7]
    double *obs_px; uint8_t *obs_ok; size_t stride;          // [n_pairs][stride][2], [n_pairs][stride]
};
int ygz_launch_win_project(ygz_hip_ctx *ctx, const YgzWinProject &P);
bool ygz_ba_window_has_dup(const ygz_hip_ctx *ctx, int window);    // an uploaded BA window repeats a (point, free pose) pair (ba.hip)
int ygz_ba_fetch_result(ygz_hip_ctx *ctx, int window, const void *d_stats, ygz_ba_stats *stats, double *poses, double *points, double *chi2_edge);   // ba.hip: statistics + state + per-edge chi2 in one transfer

// ---------------------------------------------------------------------------------------------
// device helpers
#ifdef __HIPCC__
__device__ __forceinline__ int ygz_lane() { return (int)(threadIdx.x & 63); }
// wave-uniform: raise this wavefront's issue priority for the rest of the kernel (on != 0)
__device__ __forceinline__ void ygz_raise_prio(int on) { if (on) __builtin_amdgcn_s_setprio(3); }

// order-preserving map float -> uint32 (for atomicMax on non-NaN floats)
__device__ __forceinline__ uint32_t ygz_f2ord(float f)
{
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// XCD-aware block mapping for 2-D grids (x = work inside one frame / frame pair, y = frame or pair index).
// MI355X dispatches workgroup L to XCD L % 8 and every XCD has a private 4 MiB L2.  With the natural mapping the 8
// XCDs all stream the same frame at the same time (8 copies in 8 L2s); here frame/pair y is pinned to XCD y % 8, so
// each L2 holds only its own frames and a frame's pixels are fetched from HBM once.  Launch with
// gridDim.y = round_up(n_outer, 8).  Pure performance: any placement gives the same results.
__device__ __forceinline__ bool ygz_xcd_remap(int n_outer, int &bx, int &outer)
{
    const int nbx = (int)gridDim.x;
    const int L = (int)blockIdx.x + nbx * (int)blockIdx.y;
    const int xcd = L & 7, j = L >> 3;
    const int ol = j / nbx;
    bx = j - ol * nbx;
    outer = ol * 8 + xcd;
    return outer < n_outer;
}
// same for 3-D grids: (x, y) = tile inside a frame, z = frame / pair index; launch with gridDim.z = round_up(n_outer, 8)
__device__ __forceinline__ bool ygz_xcd_remap3(int n_outer, int &bx, int &by, int &outer)
{
    const int nx = (int)gridDim.x, nb = nx * (int)gridDim.y;
    const int L = (int)blockIdx.x + nx * ((int)blockIdx.y + (int)gridDim.y * (int)blockIdx.z);
    const int xcd = L & 7, j = L >> 3;
    const int ol = j / nb, r = j - ol * nb;
    by = r / nx; bx = r - by * nx;
    outer = ol * 8 + xcd;
    return outer < n_outer;
}
static inline int ygz_round_up8(int n) { return (n + 7) & ~7; }

// Wave64 float sum on the DPP data path (row-local adds + 4 readlanes) instead of 6 rounds of ds_bpermute: the
// cross-lane latency drops from ~6 x LDS round trips to a few dependent VALU ops.  Fixed order:
// ((q ^ 1) , (q ^ 2)) inside quads, half-row mirror, row mirror, then rows 0..3 left to right.  Result is uniform.
__device__ __forceinline__ float ygz_wave_sum_f(float v)
{
    v = __fadd_rn(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false)));   // quad_perm [1,0,3,2]
    v = __fadd_rn(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false)));   // quad_perm [2,3,0,1]
    v = __fadd_rn(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false)));  // row_half_mirror
    v = __fadd_rn(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false)));  // row_mirror
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return __fadd_rn(__fadd_rn(__fadd_rn(r0, r1), r2), r3);
}
__device__ __forceinline__ int ygz_wave_sum_i(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
// lane i receives lane i - n of its row of 16 (0 when there is none): one DPP-modified VALU op instead of an LDS round trip
#define YGZ_DPP_SHR(v, n) __builtin_amdgcn_update_dpp(0, (v), 0x110 + (n), 0xF, 0xF, false)
// inclusive prefix sum over the 64 lanes (order of the additions is irrelevant to its users: approximate prefixes only)
__device__ __forceinline__ float ygz_wave_scan_f(float v)
{
#define YGZ_SCAN_STEP_(ctrl, rmask) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), (rmask), 0xF, false))
    YGZ_SCAN_STEP_(0x111, 0xF); YGZ_SCAN_STEP_(0x112, 0xF); YGZ_SCAN_STEP_(0x114, 0xF); YGZ_SCAN_STEP_(0x118, 0xF);   // row_shr 1, 2, 4, 8
    YGZ_SCAN_STEP_(0x142, 0xA);                                                                                 // row_bcast:15 -> rows 1, 3
    YGZ_SCAN_STEP_(0x143, 0xC);                                                                                 // row_bcast:31 -> rows 2, 3
#undef YGZ_SCAN_STEP_
    return v;
}
// FP64 sum over the 64 lanes, same fixed order as ygz_wave_sum_f: two DPP moves (the halves) + one v_add_f64 per step instead
// of two LDS-crossbar permutes; result is uniform.
__device__ __forceinline__ double ygz_wave_sum_d(double v)
{
#define YGZ_DSTEP_(ctrl) { const int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(v), (ctrl), 0xF, 0xF, false);            \
                           const int hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(v), (ctrl), 0xF, 0xF, false);            \
                           v += __hiloint2double(hi_, lo_); }
    YGZ_DSTEP_(0xB1) YGZ_DSTEP_(0x4E) YGZ_DSTEP_(0x141) YGZ_DSTEP_(0x140)
#undef YGZ_DSTEP_
    const int l = __double2loint(v), h = __double2hiint(v);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(h, 0), __builtin_amdgcn_readlane(l, 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(h, 16), __builtin_amdgcn_readlane(l, 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(h, 32), __builtin_amdgcn_readlane(l, 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(h, 48), __builtin_amdgcn_readlane(l, 48));
    return ((r0 + r1) + r2) + r3;
}
// N FP64 sums over the 64 lanes at once, stage by stage over ALL values: N independent chains per stage instead of N sums one after the other, each
// a chain of dependent FP64 adds with exec-mask changes in between.  The additions and their order are those of ygz_wave_sum_d (inside the rows of
// 16, then ((r0 + r1) + r2) + r3, the last three through row broadcasts into rows 1, 2, 3; disabled rows add -0.0): bit-identical totals, which
// lanes 48..63 hold on return.
template <int N>
__device__ __forceinline__ void ygz_wave_sums_d(double (&v)[N])
{
#define YGZ_SUMS_STAGE_(ctrl, rmask)                                                                                                \
    _Pragma("unroll") for (int k_ = 0; k_ < N; ++k_) {                                                                              \
        const int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(v[k_]), (ctrl), (rmask), 0xF, false);                          \
        const int hi_ = __builtin_amdgcn_update_dpp((int)0x80000000, __double2hiint(v[k_]), (ctrl), (rmask), 0xF, false);            \
        v[k_] += __hiloint2double(hi_, lo_); }
    YGZ_SUMS_STAGE_(0xB1, 0xF) YGZ_SUMS_STAGE_(0x4E, 0xF) YGZ_SUMS_STAGE_(0x141, 0xF) YGZ_SUMS_STAGE_(0x140, 0xF)
    YGZ_SUMS_STAGE_(0x142, 0x2) YGZ_SUMS_STAGE_(0x143, 0x4) YGZ_SUMS_STAGE_(0x142, 0x8)
#undef YGZ_SUMS_STAGE_
}
// correctly rounded float sqrt: the native v_sqrt_f32 path is 1 ulp; sqrt in double then one rounding is
// exact for float inputs (53 >= 2*24+2 bits)
__device__ __forceinline__ float ygz_sqrtf_cr(float x) { return (float)sqrt((double)x); }

// 8 (or 5) consecutive bytes from an arbitrary byte address: ONE unaligned global_load_dwordx2 (a wave-wide byte gather costs the address
// unit as much as a dword load, so patch windows are fetched row-wise; the memory pipeline handles the misalignment).  Rounds 1-3 fetched
// two or three aligned dwords and shifted them together with v_alignbyte: five VALU instructions per row that the VALU-bound kernels
// could not afford (-DYGZ_ALIGNED_LOADS keeps that form for A/B).  Touches exactly 8 bytes from p (every image buffer has 64 bytes of slack).
typedef const __attribute__((address_space(1))) uint32_t *ygz_gptr32;
typedef uint16_t __attribute__((aligned(1))) ygz_u16u;          // 2 adjacent bytes at any address: one (unaligned) global_load_ushort
typedef uint32_t __attribute__((aligned(1))) ygz_u32u;          // 4 adjacent bytes at any address: one (unaligned) global_load_dword
typedef const __attribute__((address_space(1))) ygz_u32u *ygz_gptr32u;
typedef uint32_t ygz_u32x2 __attribute__((ext_vector_type(2)));
typedef ygz_u32x2 __attribute__((aligned(1))) ygz_u32x2u;       // 8 adjacent bytes at any address: one (unaligned) global_load_dwordx2
typedef const __attribute__((address_space(1))) ygz_u32x2u *ygz_gptr2u;
__device__ __forceinline__ void ygz_load8(const uint8_t *p, uint32_t &lo, uint32_t &hi)
{
#ifdef YGZ_ALIGNED_LOADS
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    ygz_gptr32 q = (ygz_gptr32)(a & ~(uintptr_t)3);
    const uint32_t sh = (uint32_t)(a & 3);
    const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
    lo = __builtin_amdgcn_alignbyte(d1, d0, sh);
    hi = __builtin_amdgcn_alignbyte(d2, d1, sh);
#else
    const ygz_u32x2 v = *(ygz_gptr2u)reinterpret_cast<uintptr_t>(p);
    lo = v.x; hi = v.y;
#endif
}
__device__ __forceinline__ void ygz_load5(const uint8_t *p, uint32_t &lo, uint32_t &hi)      // bytes 0..4 valid
{
#ifdef YGZ_ALIGNED_LOADS
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    ygz_gptr32 q = (ygz_gptr32)(a & ~(uintptr_t)3);
    const uint32_t sh = (uint32_t)(a & 3);
    const uint32_t d0 = q[0], d1 = q[1];
    lo = __builtin_amdgcn_alignbyte(d1, d0, sh);
    hi = d1 >> (8 * sh);
#else
    const ygz_u32x2 v = *(ygz_gptr2u)reinterpret_cast<uintptr_t>(p);
    lo = v.x; hi = v.y;
#endif
}
#define YGZ_BYTE(lo, hi, k) ((int)((((k) < 4) ? ((lo) >> (8 * ((k) & 3))) : ((hi) >> (8 * ((k) & 3)))) & 255u))
__device__ __forceinline__ float ygz_ord2f(uint32_t o)
{
    uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    return __uint_as_float(u);
}
#endif
