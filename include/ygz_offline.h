/*
 * ygz_offline.h -- C ABI of the batched offline run (BASELINE.json configs[4]) in libygz_host.so: one sequence of N frames sharded
 * over the GPUs of a node, one process (rank) per GPU, host code in C++, collectives through RCCL directly.
 *
 * What it replaces in the reference (paths relative to the reference tree): the per-frame loop of the callers,
 *   test/test_vo_track.cpp:100-113 -> VisualOdometry::AddFrame (src/Module/VisualOdometry.cpp:38-107: TrackRefFrame, TrackLocalMap,
 *   OptimizeCurrentPoseOnly per frame) and LocalMapping::LocalBA (src/Module/LocalMapping.cpp:149-208,301-336 -> ba::LocalBAG2O,
 *   src/Algorithm/BA.cpp:386-543),
 * restructured for throughput: frames are tracked in chunks (all pairs of a chunk per launch, every pair solved from T_ref = identity so
 * that a shard needs no pose of its neighbour), keyframes = every kf_stride-th frame, BA windows = window_kfs consecutive keyframes in
 * the gauge of their anchor, built and optimised on the device while later chunks run.  The sharded run equals the unsharded one bit
 * for bit.  Ranks exchange exactly twice per run: the relative poses T_rel of their frames (one all-gather, 56 bytes per frame) and the
 * refined window states = keyframe poses + map points (one all-gather, 48 KB per window: north_star's "broadcast of map points");
 * keyframe rows cross ranks point to point only for windows that straddle a shard boundary.
 *
 * The driver owns: the shard and its chunk plan, `lanes` tracking contexts (ygz_hip_ctx) that take the chunks in turn plus one context
 * for the keyframe store and the BA windows, page-locked result rows per chunk, window readiness, the resident-LM launches and their
 * retry, the exchanges.  Python (bench.py, tests/) only renders frames into page-locked memory, calls ygz_offline_run and reads the
 * result arrays.
 *
 * Conventions as ygz_hip.h: int status codes (YGZ_OK or YGZ_E_*), nothing throws across the boundary, poses as 7 doubles
 * (qx,qy,qz,qw,tx,ty,tz).  Not thread-safe per handle.
 */
#ifndef YGZ_OFFLINE_H_
#define YGZ_OFFLINE_H_
#include "ygz_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

#define YGZ_OFFLINE_SUMMARY_FIELDS 32       /* per tracked pair: the row of ygz_hip_track_get_summary */
#define YGZ_OFFLINE_STATE_TAIL     12       /* per window behind poses and points: K, P, E, iterations, trials, chi2_0, chi2, lambda, edges tested, outliers, chi2, chi2 of inliers */
#define YGZ_OFFLINE_RCCL_ID_BYTES  128      /* sizeof(ncclUniqueId) */

typedef struct ygz_offline ygz_offline;

typedef struct {
    int width, height, levels;        /* frame size (1280 x 720 in configs[4]), pyramid levels (Basic/Frame.h:23: 3) */
    int n_frames;                     /* frames of the WHOLE sequence */
    int rank, world, device;          /* this process: rank of world, on HIP device `device` */
    int chunk;                        /* frames per chunk (128) */
    int kf_stride, window_kfs;        /* keyframe = every kf_stride-th frame (8); window = window_kfs consecutive keyframes (8) */
    int max_points, ba_iterations;    /* map points per window (2000); optimize(20) (BA.cpp:501-502) */
    int lanes;                        /* tracking contexts that take the chunks in turn (3; 4 pays with gray frames) */
    int lm_group;                     /* windows per resident-LM launch; 0: half of this rank's windows (at most 8) on a shard of > 256 frames, else all */
    int obs_mode;                     /* 1: observations = LocalMapping::ProjectMapPoints (projection + FindDirectProjection); 0: good Hamming matches */
    int ba_rounds;                    /* 1; 2: the chi2 > outlier_chi2 edges are switched off and the window is optimised once more */
    double outlier_chi2;              /* 5.991 (BA.cpp:503-515) */
    int frame_channels;               /* 3: BGR frames [h][w][3]; 1: gray frames [h][w] (cv::cvtColor done by the caller) */
    int depth_w, depth_h, depth_kind; /* depth image per frame (ygz_hip_upload_depth_batch: kind 0 f32 m, 1 u16 * depth_scale, 2 f64 m) */
    double depth_scale;
    int pipeline_ba;                  /* 1: windows are built and optimised as soon as their last keyframe is tracked; 0: after the tracking */
    int defer_gaps;                   /* keyframe-free gaps behind the last windows processed at the very end (chunk plan); -1: 13 on a shard of >= 768 frames, every gap on a shard of <= 160 frames, else 0 */
    int ramp, kf_tail;                /* chunk plan: short chunks at both ends (1); the frames behind the shard's last keyframe as the last chunk (1) */
    int stage_overlap;                /* ygz_hip_set_overlap on the lanes (0) */
    int bg_team_budget;               /* workgroups a resident-LM launch may hold while tracking chunks follow (0: library default) */
    int bg_team_spread;               /* 1 (default): such a launch spreads its teams over the XCDs (ygz_hip_ba_set_team_placement); the last launch of a run, which nothing follows, is compact */
} ygz_offline_params;

/* The two exchange primitives, on HOST memory.  NULL hook + world > 1: RCCL on device buffers (the product path).  A hook replaces RCCL
 * where it cannot run -- two ranks on ONE device in the tests (RCCL refuses that) carry the same calls over gloo.  Both return 0 on success.
 *   all_gather: every rank contributes `bytes` at send; recv [world][bytes] in rank order.
 *   send_recv : rank src sends `bytes` at buf to rank dst; called by both with the same arguments (only src and dst call it). */
typedef struct {
    void *user;
    int (*all_gather)(void *user, const void *send, void *recv, size_t bytes);
    int (*send_recv)(void *user, void *buf, size_t bytes, int src, int dst);
} ygz_offline_exchange;

/* what a run leaves (arrays owned by the handle, valid until the next run / destroy) */
typedef struct {
    int n_frames, first_frame, n_own;            /* the shard: frames [first_frame, first_frame + n_own) */
    const double  *T_rel;                         /* [n_frames][7] pose of frame f in the frame of f - 1 (row 0 = identity), every rank's */
    const double  *trajectory;                    /* [n_frames][7] T[0] = identity, T[f] = T_rel[f] * T[f - 1] (VisualOdometry.cpp:66) */
    const double  *summary;                       /* [n_frames][32] per-pair summary rows of the frames this rank tracked (others 0) */
    const int32_t *n_kp;                          /* [n_frames] keypoints per frame (own frames; others 0) */
    int n_windows, state_doubles;                 /* every window of the sequence; row = [poses 6 K | points 3 max_points | 12 tail] */
    const double  *window_state;                  /* [n_windows][state_doubles], every owner's rows after the exchange */
    const int32_t *window_owner;                  /* [n_windows] */
    const int32_t *window_kfs;                    /* [n_windows][window_kfs] frames (unused entries -1) */
    int n_chunks, lm_launches, lm_retries, n_degenerate;
    double ms_track, ms_gather, ms_ba_tail, ms_exchange;      /* host clock of the last run: tracking, T_rel gather + rows, LM tail behind the tracking, state exchange + download */
    int backend;                                  /* 0: single rank, 1: RCCL, 2: host hook */
} ygz_offline_results;

void ygz_offline_default_params(ygz_offline_params *p);
/* contiguous shard of `rank`: frames [first, first + count), halo = 1 when the predecessor of `first` is extracted here too */
int  ygz_offline_shard(int n_frames, int rank, int world, int *first, int *count, int *halo);
/* the chunks of the shard in processing order (host logic only, no device): ranges [n][3] = (chunk index, first frame, end frame);
 * a chunk may consist of several ranges.  *n = number of ranges (also when it exceeds capacity: YGZ_E_CAPACITY) */
int  ygz_offline_plan(const ygz_offline_params *p, int32_t *ranges, int capacity, int *n);
/* the same for an explicit frame range [first, last) and window list (win_first_last [n_windows][2] = anchor frame, last keyframe);
 * kf_stride > 0: the frames behind the range's last keyframe form the last chunk; defer as ygz_offline_params::defer_gaps (>= 0) */
int  ygz_offline_plan_range(int first, int last, int chunk, int ramp, int kf_stride, const int32_t *win_first_last, int n_windows, int defer,
                            int32_t *ranges, int capacity, int *n);
/* The exchange both collectives of a run are built on, in its host form: rank r contributes counts[r] rows of row_bytes at `local`; `full`
 * receives the concatenation in rank order (sum(counts) rows) through ONE hook->all_gather of blocks padded to max(counts) rows.  The
 * driver uses it for T_rel (counts = frames per shard, 56-byte rows) and for the window states (counts = windows per owner); with RCCL the same
 * padded blocks travel through ncclAllGather on device buffers.  No device needed: the CPU tests run it with world size 2 over gloo. */
int  ygz_offline_ragged_all_gather(const ygz_offline_exchange *hook, int rank, int world, const int32_t *counts, size_t row_bytes, const void *local, void *full);
/* rank 0 calls this once and hands the 128 bytes to every rank (any bootstrap channel: MPI, a file, torch's store) */
int  ygz_offline_rccl_unique_id(void *id128);
/* world > 1: either rccl_id (from ygz_offline_rccl_unique_id; collective: every rank must call) or hook; world == 1: both NULL */
int  ygz_offline_create(ygz_offline **out, const ygz_offline_params *p, const void *rccl_id, const ygz_offline_exchange *hook);
void ygz_offline_destroy(ygz_offline *o);
const char *ygz_offline_last_error(const ygz_offline *o);

/* The frames this rank needs are [first - halo, first + count): frames / depth point at frame `first_in_buffer` of PAGE-LOCKED arrays
 * [n][h][w][channels] u8 and [n][depth_h][depth_w] (ygz_hip_pinned_alloc); they must stay valid until the run returns. */
int  ygz_offline_run(ygz_offline *o, const uint8_t *frames, const void *depth, int first_in_buffer);
/* the same in steps (what ygz_offline_run calls): the tracking of every chunk (+ the windows it completes), the T_rel all-gather and the
 * keyframe rows of straddling windows, the remaining windows + the state exchange */
int  ygz_offline_track(ygz_offline *o, const uint8_t *frames, const void *depth, int first_in_buffer);
int  ygz_offline_gather(ygz_offline *o);
int  ygz_offline_ba_round(ygz_offline *o);
int  ygz_offline_get_results(const ygz_offline *o, ygz_offline_results *r);

/* ---- for tests and profiling ---- */
/* called after a chunk's results are in, before its lane is reused: lane = the tracking context that holds the chunk (read anything
 * through ygz_hip.h), frames [n_frames] = sequence index of every slot, pairs [n_pairs][2] = (cur, ref) frame of every pair */
typedef void (*ygz_offline_chunk_fn)(void *user, ygz_hip_ctx *lane, int chunk, int n_frames, const int32_t *frames, int n_pairs, const int32_t *pairs);
int  ygz_offline_set_chunk_callback(ygz_offline *o, ygz_offline_chunk_fn fn, void *user);
int  ygz_offline_contexts(ygz_offline *o, ygz_hip_ctx **ba, ygz_hip_ctx **lanes, int capacity, int *n_lanes);
/* windows this rank owns (indices into the sequence's windows, ascending); local slot k of the BA context = owned[k] */
int  ygz_offline_owned_windows(const ygz_offline *o, int32_t *owned, int capacity, int *n);
/* build (and with optimize != 0 solve) owned windows [slot0, slot0 + n) now, behind everything the lanes have enqueued */
int  ygz_offline_build_windows(ygz_offline *o, int slot0, int n, int optimize);
/* the retry of a resident-LM team that timed out at a barrier: the windows are rebuilt and solved by ONE workgroup each */
int  ygz_offline_retry_windows(ygz_offline *o, const int32_t *slots, int n);

#ifdef __cplusplus
}
#endif
#endif /* YGZ_OFFLINE_H_ */
