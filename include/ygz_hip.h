/*
 * ygz_hip.h -- C ABI of the MI355X (gfx950) implementation of ygz-slam's per-frame hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  The C++
 * class surfaces (include/ygz/...: ygz::Frame, FeatureDetector, Matcher, Tracker, ba::)
 * sit on top of it; INTEGRATION.md shows the bindings.  Every entry point cites the
 * reference interface it replaces (paths relative to the reference tree).
 *
 * Conventions
 *  - every function returns int: YGZ_OK (0) or a negative YGZ_E_* code; nothing throws or
 *    aborts (reference error conventions are bool/count returns);
 *  - a context owns all device memory (HBM) and one HIP stream; functions on the same
 *    context are not thread-safe, different contexts are independent (one per GPU/thread);
 *  - compute entry points are asynchronous on the context's stream and operate on data that
 *    is already resident in HBM; the ygz_hip_*_upload / *_download / get_* functions move
 *    host data and synchronise;
 *  - "slot" = index of a frame resident in HBM ([0, max_frames)); kernels are batched over
 *    contiguous slot ranges;
 *  - poses are 7 doubles (qx,qy,qz,qw,tx,ty,tz) = Sophus::SE3 (unit quaternion+translation);
 *    descriptors are 32 bytes per row (8 x u32), byte i bit j = BRIEF test 8i+j.
 */
#ifndef YGZ_HIP_H_
#define YGZ_HIP_H_
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define YGZ_OK              0
#define YGZ_E_INVALID      -1   /* bad argument */
#define YGZ_E_HIP          -2   /* a HIP runtime call failed (see ygz_hip_last_hip_error) */
#define YGZ_E_NO_DEVICE    -3   /* no usable gfx950 device */
#define YGZ_E_CAPACITY     -4   /* slot / keypoint / batch capacity exceeded */
#define YGZ_E_STATE        -5   /* call order violated (e.g. detect before pyramid) */

#define YGZ_MAX_LEVELS      8

/* Bumped whenever an exported function changes its argument list under the same name (C has no mangling: a caller built against an older
 * header would still link).  Bindings compare ygz_hip_abi_version() with the header they were written against at load time
 * (ygz_slam_amd/_lib.py, include/ygz/hip/Runtime.h).  5: ygz_hip_kf_row_bytes / ygz_hip_kf_store_create / ygz_hip_ba_build_windows
 * gained their trailing int (round 4); ygz_hip_get_stream / device_alloc / copy added (round 5).  6: ygz_ceres_options gained trust_region_strategy
 * (in the struct's tail padding: same size); ygz_hip_find_direct_projection_mp (+ _begin / _end), ygz_hip_sparse_align_residuals, ygz_hip_ba_light_barrier added (round 6). */
#define YGZ_HIP_ABI_VERSION 6

typedef struct ygz_hip_ctx ygz_hip_ctx;

typedef struct {
    int   image_width, image_height;  /* level-0 size; config/default.yaml:15-16 (640x480) */
    int   pyramid_levels;             /* Basic/Frame.h:23 (3) */
    int   cell_size;                  /* feature.cell, default.yaml:50 (10) */
    int   fast_threshold;             /* feature.detection_threshold, default.yaml:51 (15) */
    int   nms_tie_suppress;           /* 0: drop a corner iff a neighbour is strictly greater */
    int   max_frames;                 /* frame slots resident in HBM */
    float fx, fy, cx, cy;             /* PinholeCamera float intrinsics, Basic/Camera.h:107 */
    int   debug_maps;                 /* !=0: keep per-pixel FAST score / NMS maps for parity tests */
} ygz_hip_params;

/* keypoints of one frame, structure-of-arrays (mirror of ygz::Feature, Basic/Feature.h:15-36) */
typedef struct {
    double  *px;       /* [n][2] level-0 pixel  (Feature::_pixel) */
    int32_t *level;    /* [n]                   (Feature::_level) */
    float   *score;    /* [n] Shi-Tomasi        (Feature::_score) */
    float   *angle;    /* [n] degrees           (Feature::_angle) */
    uint8_t *desc;     /* [n][32]               (Feature::_desc) */
} ygz_kpt_soa;

/* ---- context ---------------------------------------------------------------------------- */
void ygz_hip_default_params(ygz_hip_params *p);
/* stream: an existing hipStream_t to launch on (e.g. torch's current stream), or NULL to create one */
int  ygz_hip_create(ygz_hip_ctx **out, int device, const ygz_hip_params *prm, void *stream);
void ygz_hip_destroy(ygz_hip_ctx *ctx);
int  ygz_hip_synchronize(ygz_hip_ctx *ctx);
/* the context's stream waits for every stage pending on a side stream (what ygz_hip_synchronize does first), without blocking the host:
 * an event recorded on the stream afterwards marks the end of everything enqueued so far (bench.py's per-step timing) */
int  ygz_hip_join(ygz_hip_ctx *ctx);
/* enable != 0: the resident stages that do not depend on each other -- ygz_hip_track_sparse_align,
 * ygz_hip_ba_linearize_resident, ygz_hip_match_slots_again -- are launched on side HIP streams forked from the context's
 * stream, so they overlap with whatever is enqueued after them (KLT, direct projection); every entry point that reads or
 * overwrites their data, and ygz_hip_synchronize, joins them first.  Off by default. */
int  ygz_hip_set_overlap(ygz_hip_ctx *ctx, int enable);
/* Host work of the caller's own while a single-frame call waits for its kernel: fn(user) is called ONCE by the next ygz_hip_sparse_align (the longest wait
 * of a frame: one Gauss-Newton problem, ~0.3 ms) between its launch and its wait, then the hook is cleared.  fn must not call into this context.
 * fn == NULL clears.  (The class surface gathers the candidates of the frame's speculative FindDirectProjection launch there.) */
int  ygz_hip_set_wait_hook(ygz_hip_ctx *ctx, void (*fn)(void *), void *user);
const char *ygz_hip_error_string(int code);
int  ygz_hip_last_hip_error(const ygz_hip_ctx *ctx);
int  ygz_hip_max_keypoints(const ygz_hip_ctx *ctx);     /* = number of grid cells */
int  ygz_hip_abi_version(void);                         /* YGZ_HIP_ABI_VERSION of the library that is loaded */
/* plumbing for host code that drives several contexts and a collective library (ygz_slam_amd/host/ygz_offline.cpp; no reference
 * counterpart -- the reference is single-GPU): the context's hipStream_t as an opaque pointer (RCCL calls are enqueued on it), its device
 * ordinal / CU count, zero-filled device memory for exchange buffers, and a copy ordered on the stream (kind 0: host -> device, 1: device ->
 * host, 2: device -> device; wait == 0 needs page-locked host memory) */
int  ygz_hip_get_stream(ygz_hip_ctx *ctx, void **stream);
int  ygz_hip_get_device(const ygz_hip_ctx *ctx, int *device, int *compute_units);
int  ygz_hip_make_current(ygz_hip_ctx *ctx);            /* hipSetDevice(the context's device) for the calling thread, left that way */
int  ygz_hip_device_alloc(ygz_hip_ctx *ctx, void **out, size_t bytes);
int  ygz_hip_device_free(ygz_hip_ctx *ctx, void *p);
int  ygz_hip_copy(ygz_hip_ctx *ctx, void *dst, const void *src, size_t bytes, int kind, int wait);
/* HIP-event timing on the context's stream (bench.py roofline leg) */
int  ygz_hip_timer_begin(ygz_hip_ctx *ctx);
int  ygz_hip_timer_end(ygz_hip_ctx *ctx, float *elapsed_ms);   /* synchronises */
/* HIP events around EVERY launch of one named kernel (e.g. "k_klt") until probe_end: total time and launch count */
int  ygz_hip_probe_begin(ygz_hip_ctx *ctx, const char *kernel_name, int max_launches);
int  ygz_hip_probe_end(ygz_hip_ctx *ctx, double *total_ms, int *launches);   /* synchronises */

/* ---- A1: frame store + pyramid -- replaces Frame::InitFrame / CreateImagePyramid
 *      (src/Basic/Frame.cpp:22-40: cv::cvtColor BGR2GRAY + cv::pyrDown per level) -------------- */
int  ygz_hip_upload_bgr(ygz_hip_ctx *ctx, int slot, const uint8_t *bgr, int stride_bytes);
int  ygz_hip_upload_gray(ygz_hip_ctx *ctx, int slot, const uint8_t *gray, int stride_bytes);
/* builds levels 1..pyramid_levels-1 (and level 0 from BGR when from_bgr!=0) for n slots */
int  ygz_hip_build_pyramid(ygz_hip_ctx *ctx, int slot_begin, int n_slots, int from_bgr);
int  ygz_hip_download_level(ygz_hip_ctx *ctx, int slot, int level, uint8_t *dst /* w_l*h_l */);
/* the tracker's working image of a level: the level inside a 24-pixel BORDER_REFLECT_101 frame (what cv::buildOpticalFlowPyramid makes with
 * copyMakeBorder), as the pyramid kernels wrote it once the tracker's buffers exist (first ygz_hip_track_klt / ygz_hip_klt_track).
 * dst [h + 48][w + 48]; YGZ_E_STATE when the slot has no current framed copy of the level.  (Test / debugging aid.) */
int  ygz_hip_download_framed_level(ygz_hip_ctx *ctx, int slot, int level, uint8_t *dst);
int  ygz_hip_level_size(const ygz_hip_ctx *ctx, int level, int *w, int *h);

/* ---- A2-A7: extractor -- replaces FeatureDetector::Detect
 *      (src/Algorithm/FeatureDetector.cpp:345-444: fast_corner_detect_10 + fast_corner_score_10
 *      + fast_nonmax_3x3 per level, per-cell best Shi-Tomasi, IC_Angle, ComputeOrbDescriptor) ---- */
/* occupied: host [n_slots][cells] bytes (SetExistingFeatures, :446-464) or NULL.  Results stay
 * in HBM (keypoint SoA per slot, cell-index order == the order Detect pushes to _features). */
int  ygz_hip_detect(ygz_hip_ctx *ctx, int slot_begin, int n_slots, const uint8_t *occupied);
int  ygz_hip_keypoint_count(ygz_hip_ctx *ctx, int slot, int *n);
int  ygz_hip_get_keypoints(ygz_hip_ctx *ctx, int slot, ygz_kpt_soa *out, int capacity, int *n);
/* FeatureDetector::ComputeAngleAndDescriptor(Frame*) (:580-588): replaces the keypoint list of
 * `slot` by the given pixels/levels and (re)computes angle + descriptor for them */
int  ygz_hip_describe(ygz_hip_ctx *ctx, int slot, const double *px /*[n][2]*/, const int32_t *level, int n);
/* FeatureDetector::ComputeDescriptor(Feature*) (:591-594): descriptor only, with the given angles (degrees) */
int  ygz_hip_describe_given_angle(ygz_hip_ctx *ctx, int slot, const double *px, const int32_t *level, const float *angle, int n);
/* parity/debug (needs debug_maps): per-pixel maps of one level after ygz_hip_detect:
 * score[y*w+x] = 0 (no FAST-10 corner at the threshold) or fast_corner_score_10 + 1;
 * nms[y*w+x]   = 1 iff the corner survives fast_nonmax_3x3 */
int  ygz_hip_get_fast_maps(ygz_hip_ctx *ctx, int slot, int level, uint8_t *score, uint8_t *nms);

/* ---- M1-M3: 256-bit Hamming matcher -- replaces Matcher::DescriptorDistance
 *      (src/Algorithm/Matcher.cpp:30-43) and cv::BFMatcher(NORM_HAMMING, crossCheck).match()
 *      (test/test_orb_match.cpp:86-93) -------------------------------------------------------- */
/* descriptor sets of slots (query_slot[i], train_slot[i]), i < n_pairs, as left by detect/describe.
 * cross_check: 0 plain nearest neighbour, 1 OpenCV cross-check, 2 strict mutual NN.
 * Results per pair stay in HBM; fetch with ygz_hip_get_matches. */
int  ygz_hip_match_slots(ygz_hip_ctx *ctx, const int32_t *query_slot, const int32_t *train_slot,
                         int n_pairs, int cross_check);
/* re-run the matcher on the pair table uploaded by the last ygz_hip_match_slots (no host copies) */
int  ygz_hip_match_slots_again(ygz_hip_ctx *ctx, int cross_check);
int  ygz_hip_get_matches(ygz_hip_ctx *ctx, int pair, int32_t *train_idx /*[nq]*/, int32_t *dist /*[nq]*/,
                         int capacity, int *nq);
/* stand-alone form on host descriptor arrays (uploads, matches, downloads).  dist2 (second-best
 * distance, Matcher::SearchByBoW :242-246 ratio test) may be NULL and needs cross_check==0.
 * nq, nt <= grid cells x max_frames (the result rows of all pairs of the context), YGZ_E_CAPACITY beyond. */
int  ygz_hip_hamming_match(ygz_hip_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt,
                           int cross_check, int32_t *train_idx, int32_t *dist, int32_t *dist2);

/* ---- M3: the "good match" filter of test/test_orb_match.cpp:97-104 on the matcher's output: min_dis = the smallest distance
 *      among the matches, clamped to [min_floor, min_ceil] (20, 50); a match is good iff distance < factor * min_dis (3).
 *      Resident form: every pair of the current pair table in one launch, flags stay in HBM (ygz_hip_get_good_matches). */
int  ygz_hip_match_postfilter(ygz_hip_ctx *ctx, double min_floor, double min_ceil, double factor);
int  ygz_hip_get_good_matches(ygz_hip_ctx *ctx, int pair, uint8_t *good /*[nq]*/, int capacity, int *nq, int *n_good, double *min_dis);
/* the same on host arrays as ygz_hip_hamming_match returned them (train_idx < 0: no match).  nq == 0 (the reference dereferences
 * end() there) is defined as "nothing kept". */
/* M1-M3 over many host descriptor sets in one call (the matching a keyframe window needs: one set against several others):
 * desc[s] = count[s] rows of 32 bytes; pair p matches query set pair_q[p] against train set pair_t[p] with BFMatcher semantics
 * (test/test_orb_match.cpp:86-93) and, when any of good / n_good / min_dis is given, applies the good-match rule of :95-104.
 * train_idx / dist / good: [n_pairs][ygz_hip_max_keypoints()] rows (entries past count[pair_q[p]] are not written). */
int  ygz_hip_match_sets(ygz_hip_ctx *ctx, int n_sets, const uint8_t *const *desc, const int32_t *count, int n_pairs,
                        const int32_t *pair_q, const int32_t *pair_t, int cross_check, int32_t *train_idx, int32_t *dist,
                        uint8_t *good, int32_t *n_good, double *min_dis, double min_floor, double min_ceil, double factor);
int  ygz_hip_match_postfilter_host(ygz_hip_ctx *ctx, const int32_t *train_idx, const int32_t *dist, int nq, double min_floor,
                                   double min_ceil, double factor, uint8_t *good, int *n_good, double *min_dis);
/* ---- M6: Matcher::CheckFrameDescriptors (src/Algorithm/Matcher.cpp:45-84): Hamming distance of n given (index1, index2) feature
 *      pairs of two frames, best_dist = min clamped to [init_low, init_high] (Matcher.h:27-28), keep[i] = dist[i] < ratio * best_dist
 *      (initMatchRatio 3.0).  Slot form: resident descriptors of the two slots; pair form: host descriptors [n][32] row by row.
 *      n == 0 (UB in the reference) returns n_good = 0. */
int  ygz_hip_check_frame_descriptors(ygz_hip_ctx *ctx, int slot1, int slot2, const int32_t *idx1, const int32_t *idx2, int n,
                                     int init_low, int init_high, float ratio, int32_t *dist, uint8_t *keep, int *n_good, int *best_dist);
int  ygz_hip_check_descriptor_pairs(ygz_hip_ctx *ctx, const uint8_t *desc1, const uint8_t *desc2, int n, int init_low, int init_high,
                                    float ratio, int32_t *dist, uint8_t *keep, int *n_good, int *best_dist);

/* ---- L1-L2: patch alignment -- replaces Matcher::FindDirectProjection (Matcher.cpp:356-417:
 *      GetWarpAffineMatrix, GetBestSearchLevel, WarpAffine, cvutils::Align2D CVUtils.cpp:186-318) -- */
typedef struct {
    int    ref_slot, cur_slot;
    double T_ref[7], T_cur[7];      /* Frame::_TCW of the two frames */
} ygz_align_pair;
/* n candidates: px_ref [n][2], depth_ref [n] (>=0; Feature::_depth or z of World2Camera),
 * level_ref [n], px_cur [n][2] in (prediction) / out (refined), search_level [n] out,
 * ok [n] out (the bool FindDirectProjection returns).  Host arrays; synchronises. */
int  ygz_hip_find_direct_projection(ygz_hip_ctx *ctx, const ygz_align_pair *pair,
                                    const double *px_ref, const double *depth_ref, const int32_t *level_ref,
                                    double *px_cur, int32_t *search_level, uint8_t *ok, int n);
/* ---- LocalMapping::FindCandidates + ProjectMapPoints (src/Module/LocalMapping.cpp:47-120) in one call.
 * FindCandidates: each non-bad local map point is projected into the current frame (World2Camera, Camera2Pixel); behind the
 * camera or outside InFrame(px,20) -> in_view = 0.  ProjectMapPoints: every candidate (an observation of a point in one of
 * the K resident local keyframes: pixel and pyramid level of that observation) is refined by the MapPoint overload of
 * Matcher::FindDirectProjection (Matcher.cpp:356-383, depth = z of the point in that keyframe); per point the FIRST success
 * in candidate order is kept (the reference iterates a std::map<Feature*,...>, i.e. heap-address order; the order here is
 * the caller's).  Outputs per point: in_view, px_proj [P][2], match_cand (candidate index or -1), px_match [P][2]
 * (Feature::_pixel of the new feature), match_level (its _level).  Host arrays; synchronises. */
typedef struct {
    int n_points;      const double *pos_world /*[P][3]*/; const uint8_t *point_bad /*[P] or NULL*/;
    int n_keyframes;   const int32_t *kf_slot /*[K]*/;     const double *kf_T /*[K][7] Frame::_TCW*/;
    int n_candidates;  const int32_t *cand_point, *cand_kf, *cand_level /*[C]*/; const double *cand_px_ref /*[C][2]*/;
} ygz_local_map;
int  ygz_hip_track_local_map(ygz_hip_ctx *ctx, int cur_slot, const double T_cur[7], const ygz_local_map *m,
                             uint8_t *in_view, double *px_proj, int32_t *match_cand, double *px_match, int32_t *match_level,
                             int32_t *n_matched);
/* ---- the call the reference's UNCHANGED caller makes: bool Matcher::FindDirectProjection(Frame *ref, Frame *curr, MapPoint *mp, Vector2d &px_curr,
 * int &search_level) (src/Algorithm/Matcher.cpp:356-383), once per candidate from LocalMapping::ProjectMapPoints (src/Module/LocalMapping.cpp:88-118).
 * n independent candidates over K resident reference keyframes in ONE launch, every candidate evaluated (no first-success rule): candidate i =
 * map point pos_world[i] seen in keyframe cand_kf[i] at px_ref[i] / level_ref[i] (the Feature mp->_obs[ref->_keyframe_id]); depth =
 * World2Camera(pos_world, ref->_TCW)[2], sign not tested (as the reference).  px_in [n][2]: the caller's predictions; NULL: the prediction is
 * LocalMapping::FindCandidates' projection Camera2Pixel(World2Camera(pos_world, T_cur)) (LocalMapping.cpp:58-59), written to px_proj [n][2] with
 * in_view [n] = !(z < 0) && InFrame(px, 20); candidates out of view are not evaluated (ok = 0, px_cur undefined).  ok [n] = the bool returned,
 * px_cur [n][2] = refined pixel (level 0), search_level [n].  Host arrays; synchronises.  The class surface memoises one such launch per
 * current frame behind the per-candidate method (ygz_host.cpp: FdpMemo), bit-identical to n = 1 calls. */
int  ygz_hip_find_direct_projection_mp(ygz_hip_ctx *ctx, int cur_slot, const double T_cur[7], int n_keyframes, const int32_t *kf_slot,
                                       const double *kf_T /*[K][7]*/, int n, const int32_t *cand_kf, const double *pos_world /*[n][3]*/,
                                       const double *px_ref, const int32_t *level_ref, const double *px_in /*or NULL*/,
                                       uint8_t *in_view /*may be NULL with px_in*/, double *px_proj /*may be NULL with px_in*/,
                                       uint8_t *ok, double *px_cur, int32_t *search_level);
/* The same launch in two halves: _begin (the px_in = NULL form) queues uploads, kernels and the copy back and does not wait; _end (same n) waits and
 * hands out what ygz_hip_find_direct_projection_mp would have returned.  One run pending per context (a second _begin replaces it); YGZ_E_STATE from
 * _end when none is (or n differs); other calls on the context in between queue behind it.  The class surface queues its speculative launch at the
 * end of Matcher::SparseImageAlignment (src/Algorithm/Matcher.cpp:16-31) and collects at the first per-candidate call of the frame. */
int  ygz_hip_find_direct_projection_mp_begin(ygz_hip_ctx *ctx, int cur_slot, const double T_cur[7], int n_keyframes, const int32_t *kf_slot,
                                             const double *kf_T, int n, const int32_t *cand_kf, const double *pos_world, const double *px_ref,
                                             const int32_t *level_ref);
int  ygz_hip_find_direct_projection_mp_end(ygz_hip_ctx *ctx, int n, uint8_t *in_view, double *px_proj, uint8_t *ok, double *px_cur,
                                           int32_t *search_level);
/* bare cvutils::Align2D on host-provided patches against level `level` of `cur_slot`:
 * pwb [n][100], patch [n][64], uv [n][2] in/out (level pixels), ok [n], chi2 [n] (may be NULL) */
int  ygz_hip_align2d(ygz_hip_ctx *ctx, int cur_slot, int level, const uint8_t *pwb, const uint8_t *patch,
                     double *uv, uint8_t *ok, float *chi2, int n, int n_iter);

/* ---- L3: sparse image alignment -- replaces SparseImgAlign::run
 *      (src/Algorithm/SparseImageAlign.cpp:21-50 + NLLSSolver::optimizeGaussNewton,
 *      include/ygz/Algorithm/NLSSolver_impl.hpp:15-89; what Matcher::SparseImageAlignment calls) --- */
/* features of the reference frame: px [n][2], depth [n], has_mappoint [n].  T_cur in/out.
 * max_level/min_level/n_iter as SparseImgAlign ctor (Matcher.cpp:18: 2,0,30).  n_meas_out =
 * the size_t run() returns.  The whole Gauss-Newton loop runs on the GPU. */
int  ygz_hip_sparse_align(ygz_hip_ctx *ctx, int ref_slot, const double T_ref[7], int cur_slot, double T_cur[7],
                          const double *px, const double *depth, const uint8_t *has_mappoint, int n,
                          int max_level, int min_level, int n_iter, int *n_meas_out, int *iters_out /*[levels] or NULL*/);

/* One NLLSSolver::computeResiduals(model, linearize_system = true) of SparseImgAlign (src/Algorithm/SparseImageAlign.cpp:124-223, with
 * precomputeReferencePatches :59-122 for `level`) at a model the caller holds -- the step a solver other than the resident Gauss-Newton loop is
 * built from; SparseImgAlign(..., LevenbergMarquardt, ...) of the class surface (include/ygz/Algorithm/NLSSolver_impl.hpp:91-212) drives its
 * trials with it.  T_cur_from_ref = the solver's model (SparseImageAlign.cpp:37).  chi2_sum: the reference's float running sum of res^2 over the
 * features / pixels in order (exact); n_meas: measurements (16 per feature used); H [36] row-major symmetric and Jres [6] (either may be NULL):
 * H_ / Jres_ after the call.  Features as ygz_hip_sparse_align.  Host arrays; synchronises. */
int  ygz_hip_sparse_align_residuals(ygz_hip_ctx *ctx, int ref_slot, int cur_slot, const double T_cur_from_ref[7],
                                    const double *px, const double *depth, const uint8_t *has_mappoint, int n, int level,
                                    double *chi2_sum, int *n_meas, double *H, double *Jres);

/* ---- L4: pyramidal LK -- replaces cv::calcOpticalFlowPyrLK as called by Tracker::TrackKLT
 *      (src/Algorithm/Tracker.cpp:92-98) ------------------------------------------------------- */
typedef struct {
    int win, max_level, max_iter;     /* Tracker.h:25-27, Tracker.cpp:97 (21, 4, 30) */
    double eps, min_eig_threshold;    /* 0.001, 1e-4 */
    int use_initial_flow;             /* OPTFLOW_USE_INITIAL_FLOW */
} ygz_klt_params;
void ygz_hip_default_klt_params(ygz_klt_params *p);
/* level-0 images of prev_slot/cur_slot; prev_pts [n][2], next_pts [n][2] in/out, status [n], err [n]; any n (served in pieces of grid-cell size) */
int  ygz_hip_klt_track(ygz_hip_ctx *ctx, int prev_slot, int cur_slot, const float *prev_pts, float *next_pts,
                       int n, const ygz_klt_params *prm, uint8_t *status, float *err);

/* the same plus Tracker::TrackKLT's survivor rule (Tracker.cpp:100-112) evaluated on the device: keep[i] = status[i] != 0 and
 * InFrame(next_pts[i], border) (20 in the reference); n_keep = number of survivors */
int  ygz_hip_klt_track_filtered(ygz_hip_ctx *ctx, int prev_slot, int cur_slot, const float *prev_pts, float *next_pts, int n,
                                const ygz_klt_params *prm, int border, uint8_t *status, float *err, uint8_t *keep, int *n_keep);

/* ---- resident batched tracking: the same L1-L4 kernels over MANY frame pairs per launch, inputs taken from
 *      the keypoints the extractor left in HBM, no host round trip between stages.  This is the per-frame path
 *      of VisualOdometry::AddFrame (src/Module/VisualOdometry.cpp:38-107: Tracker::Track -> TrackRefFrame ->
 *      TrackLocalMap) restructured for throughput; pair i = (cur_slot[i], ref_slot[i]). ------------------------ */
/* Feature::_depth / Feature::_mappoint of the keypoints of `slot` (keypoint order of ygz_hip_get_keypoints) */
int  ygz_hip_set_keypoint_depths(ygz_hip_ctx *ctx, int slot, const double *depth, const uint8_t *has_mappoint, int n);
/* uploads the pair table + poses (T_* [n_pairs][7]) and loads the reference keypoints of every pair into its
 * track set on the device.  predict != 0: the direct-projection start pixel is the projection of the feature with
 * (T_cur, T_ref) (LocalMapping::FindCandidates, LocalMapping.cpp:47-80), else the reference pixel itself.
 * KLT starts from the reference pixels (Tracker::SetReference); sparse alignment starts from T_ref (Matcher.cpp:471). */
int  ygz_hip_track_begin(ygz_hip_ctx *ctx, const int32_t *cur_slot, const int32_t *ref_slot, const double *T_cur,
                         const double *T_ref, int n_pairs, int predict);
int  ygz_hip_track_reload(ygz_hip_ctx *ctx, int predict);       /* device-side reload only (next step, same pairs) */
int  ygz_hip_track_klt(ygz_hip_ctx *ctx, const ygz_klt_params *prm);
/* optional: the working images of calcOpticalFlowPyrLK (buildOpticalFlowPyramid's framed levels + Scharr derivatives, OpenCV
   lkpyramid.cpp; called from src/Algorithm/Tracker.cpp:97) for the current pair table, built ahead on a side stream -- call after
   ygz_hip_build_pyramid; ygz_hip_track_klt (default parameters) then skips them.  Same results with or without. */
int  ygz_hip_track_klt_prepare(ygz_hip_ctx *ctx);
int  ygz_hip_track_direct(ygz_hip_ctx *ctx);
int  ygz_hip_track_sparse_align(ygz_hip_ctx *ctx, int max_level, int min_level, int n_iter);
/* VisualOdometry::TrackRefFrame -> TrackLocalMap hand-over (src/Module/VisualOdometry.cpp:281-302, src/Module/LocalMapping.cpp:47-80)
 * on the device for every pair: the pose the sparse alignment left becomes the pair's current pose and every reference feature's
 * map point (Pixel2Camera(px, depth) taken to the world with T_ref) is re-projected with it -- the start pixel of the direct
 * projection; features without depth, behind the camera or outside InFrame(px, 20) stop being candidates (ok = 0 afterwards). */
int  ygz_hip_track_adopt_pose(ygz_hip_ctx *ctx);
/* LocalMapping::OptimizeCurrent -> ba::OptimizeCurrentPoseOnly (src/Module/LocalMapping.cpp:122-127, src/Algorithm/BA.cpp:188-264) for
 * every pair, on the features ProjectMapPoints created (the direct-projection successes: pixel = refined pixel, map point = the
 * reference feature's), starting from the pair's current pose.  Results stay in HBM. */
int  ygz_hip_track_pose_only(ygz_hip_ctx *ctx);
/* pose [6] = [t; log(so3)], T [7] = SE3(SO3::exp(.), t) (BA.cpp:254); bad/depth [n] per reference feature (bad = 1 also for
 * features that were not projected); any output may be NULL */
int  ygz_hip_track_get_pose_only(ygz_hip_ctx *ctx, int pair, double pose[6], double T[7], int *inliers, int *rounds, uint8_t *bad,
                                 double *depth, int capacity, int *n);
int  ygz_hip_track_get_klt(ygz_hip_ctx *ctx, int pair, float *pts, uint8_t *status, float *err, int capacity, int *n);
int  ygz_hip_track_get_direct(ygz_hip_ctx *ctx, int pair, double *px, int32_t *level, uint8_t *ok, int capacity, int *n);
int  ygz_hip_track_get_pose(ygz_hip_ctx *ctx, int pair, double T[7], int *n_meas, int *iters /*[levels] or NULL*/);

/* ---- bulk traffic for pipelines that stream frames through the context (offline run, bench.py --stream).  wait == 0: the
 *      copy is only enqueued on the context's stream -- host buffers must then be page-locked (ygz_hip_pinned_alloc) and stay
 *      valid until the next synchronising call; wait != 0: synchronises.  Layouts are the resident ones: [n_slots][cells][...]. */
int  ygz_hip_pinned_alloc(void **out, size_t bytes);
int  ygz_hip_pinned_free(void *p);
/* Frame::_color of n consecutive slots in one copy: bgr [n_slots][h][w][3], contiguous (src/Basic/Frame.cpp:22-30 input) */
int  ygz_hip_upload_bgr_batch(ygz_hip_ctx *ctx, int slot_begin, int n_slots, const uint8_t *bgr, int wait);
/* the same for frames that are already gray (level 0 of the pyramid): gray [n_slots][h][w] */
int  ygz_hip_upload_gray_batch(ygz_hip_ctx *ctx, int slot_begin, int n_slots, const uint8_t *gray, int wait);
/* Feature::_pixel of every keypoint of n slots: px [n_slots][cells][2], count [n_slots] */
int  ygz_hip_get_keypoint_pixels_batch(ygz_hip_ctx *ctx, int slot_begin, int n_slots, double *px, int32_t *count, int wait);
/* all keypoint fields of n slots (members of *out may be NULL): arrays [n_slots][cells][...] */
int  ygz_hip_get_keypoints_batch(ygz_hip_ctx *ctx, int slot_begin, int n_slots, ygz_kpt_soa *out, int32_t *count, int wait);
/* Feature::_depth / _mappoint != nullptr of n slots: depth [n_slots][cells], has_mappoint [n_slots][cells] */
int  ygz_hip_set_keypoint_depths_batch(ygz_hip_ctx *ctx, int slot_begin, int n_slots, const double *depth, const uint8_t *has_mappoint,
                                       int wait);
/* n_kp of n consecutive slots (what Frame::_features.size() would be), e.g. beside ygz_hip_track_get_summary */
int  ygz_hip_get_keypoint_counts(ygz_hip_ctx *ctx, int slot_begin, int n_slots, int32_t *count, int wait);
/* RGB-D style input: a depth image per slot, dw x dh samples covering the level-0 frame (dw <= width, dh <= height; kind 0: float32
 * metres, kind 1: uint16 with depth = value * scale, TUM RGB-D: 1 / 5000, kind 2: float64 metres); depth [n_slots][dh][dw].  ygz_hip_keypoint_depths_from_image
 * then sets Feature::_depth of every keypoint of the slots to the sample at ((int)x * dw / width, (int)y * dh / height) and
 * _mappoint != nullptr to depth > 0, on the device -- what ygz_hip_set_keypoint_depths[_batch] do from host arrays.  (In the reference
 * a feature's depth is the z of its map point, src/Module/LocalMapping.cpp:100-111; the offline run's depth image stands in for the map.) */
int  ygz_hip_upload_depth_batch(ygz_hip_ctx *ctx, int slot_begin, int n_slots, const void *depth, int dw, int dh, int kind, double scale,
                                int wait);
int  ygz_hip_keypoint_depths_from_image(ygz_hip_ctx *ctx, int slot_begin, int n_slots);
/* Feature::_depth / _mappoint != nullptr of the keypoints of a slot as they stand on the device (either output may be NULL) */
int  ygz_hip_get_keypoint_depths(ygz_hip_ctx *ctx, int slot, double *depth, uint8_t *has_mappoint, int capacity, int *n);
/* per pair of the resident pair table 32 doubles, reduced on the device: [0..6] pose after sparse alignment, [7] its n_meas / 16,
 * [8..13] pose after pose-only BA [t; log so3], [14] inliers, [15] rounds, [16] cross-checked matches, [17] good matches (M3),
 * [18] min_dis, [19] KLT tracks with status 1, [20] direct-projection successes, [21] reference features, [22] query keypoints, [23] 0,
 * [24..30] the pose-only pose as (qx,qy,qz,qw,tx,ty,tz), [31] 0 */
int  ygz_hip_track_get_summary(ygz_hip_ctx *ctx, double *out /*[capacity_pairs][32]*/, int capacity_pairs, int *n_pairs, int wait);

/* ---- B1-B5: local-BA edge stack -- replaces the per-iteration work g2o does for
 *      EdgeSophusSE3ProjectXYZ (include/ygz/G2oTypes.h:84-132: computeError, linearizeOplus) plus
 *      BaseBinaryEdge::constructQuadraticForm with RobustKernelHuber, as driven by
 *      ba::LocalBAG2O (src/Algorithm/BA.cpp:386-543, optimize() at :501-502) ----------------------- */
typedef struct {
    int n_poses, n_points, n_edges;
    const double  *poses;        /* [n_poses][6]  g2o vertex order [omega; t] (G2oTypes.h:88) */
    const uint8_t *pose_fixed;   /* [n_poses] */
    const double  *points;       /* [n_points][3] */
    const int32_t *edge_pose;    /* [n_edges] */
    const int32_t *edge_point;   /* [n_edges] */
    const double  *obs;          /* [n_edges][2] pixels */
    double fx, fy, cx, cy;       /* G2oTypes.h:60-66 */
    double huber_delta;          /* BA.cpp:451 (5.991); <= 0: no robust kernel */
    int    formulation;          /* 0: pixel residual (G2oTypes.h); 1: normalised plane, pose order
                                    [t; omega] (legacy include/ygz/g2o_types.h:33-86, src/optimizer.cpp);
                                    2: the ceres functors (include/ygz/Ceres/CeresReprojectionError.h:33-69, ...PoseOnly.h:27-58,
                                    ...PointOnly.h:45-77): pose = [t; angle-axis] with additive update, obs in normalised
                                    coordinates, Jacobians = what AutoDiffCostFunction<...,2,6,3> returns */
    /* optional, any formulation (NULL = absent).  They express the ceres problems of src/Algorithm/BA.cpp on one edge
     * list: a PointOnly functor is an edge to a pose with pose_fixed, a PoseOnly functor an edge to a point with
     * point_fixed, SetEnable(false) is edge_enable = 0, ceres::HuberLoss(a) is edge_huber = a. */
    const uint8_t *point_fixed;  /* [n_points] */
    const double  *edge_huber;   /* [n_edges] per-edge Huber width (<= 0: none); NULL: huber_delta for every edge */
    const uint8_t *edge_enable;  /* [n_edges] */
} ygz_ba_problem;
/* Outputs (host, any may be NULL): Hpp [n_poses][36], bp [n_poses][6], Hll [n_points][9],
 * bl [n_points][3], Hpl [n_edges][18] (6x3 = Jpose^T w Jpoint), err [n_edges][2], chi2_edge [n_edges],
 * chi2 = sum of robustified chi2.  Uploads the problem, runs the kernels, downloads. */
int  ygz_hip_ba_linearize(ygz_hip_ctx *ctx, const ygz_ba_problem *pb, double *Hpp, double *bp, double *Hll,
                          double *bl, double *Hpl, double *err, double *chi2_edge, double *chi2);
/* An edge list may hold the same (point, pose) pair more than once (two features of one frame that share a map point, as
 * ba::OptimizeCurrent can produce): every edge contributes to chi2, Hll, bl, Hpp, bp and has its own Hpl block, as ceres / g2o count
 * every residual block.  Only ygz_hip_ba_optimize_resident refuses such a window (YGZ_E_INVALID); ygz_hip_ba_optimize then runs
 * the reduced system on the host. */
/* resident form: upload structure once, then re-linearise for new states without host copies */
int  ygz_hip_ba_upload(ygz_hip_ctx *ctx, int window, const ygz_ba_problem *pb);
int  ygz_hip_ba_set_state(ygz_hip_ctx *ctx, int window, const double *poses, const double *points);
/* same, state already in HBM (device pointers, e.g. the buffer an RCCL broadcast filled); asynchronous */
int  ygz_hip_ba_set_state_device(ygz_hip_ctx *ctx, int window, const double *d_poses, const double *d_points);
int  ygz_hip_ba_linearize_resident(ygz_hip_ctx *ctx, int window_begin, int n_windows);
int  ygz_hip_ba_download(ygz_hip_ctx *ctx, int window, double *Hpp, double *bp, double *Hll, double *bl,
                         double *Hpl, double *err, double *chi2_edge, double *chi2);
/* number of enabled edges whose point lay behind the camera (p_z < 0) in the last linearisation -- the condition on which
 * CeresReprojectionErrorPoseOnly::operator() reports failure (CeresReprojectionErrorPoseOnly.h:48-51) */
int  ygz_hip_ba_behind_camera(ygz_hip_ctx *ctx, int window, int *n_behind);
int  ygz_hip_ba_set_enable(ygz_hip_ctx *ctx, int window, const uint8_t *edge_enable);

/* ---- B4: the LM loop of ba::LocalBAG2O (src/Algorithm/BA.cpp:390-395,501-502: g2o OptimizationAlgorithmLevenberg +
 *      BlockSolver_6_3 with marginalised points).  Linearisations run on the GPU, the reduced pose system on the host.
 *      poses_io [n_poses][6] / points_io [n_points][3] are updated in place (pb->poses / pb->points are ignored). */
typedef struct {
    int    iterations, lm_trials;
    double chi2_initial, chi2_final, lambda_final;
} ygz_ba_stats;
int  ygz_hip_ba_optimize(ygz_hip_ctx *ctx, const ygz_ba_problem *pb, double *poses_io, double *points_io,
                         int max_iterations, ygz_ba_stats *stats);
/* The same plus the inlier pass that follows optimize() in ba::LocalBAG2O (src/Algorithm/BA.cpp:503-515): chi2_edge [n_edges] = every edge's chi2 at
 * the optimised state (what ygz_hip_ba_linearize would return for it), from one more linearisation of the window that is still resident --
 * no second upload of the graph.  chi2_edge == NULL: exactly ygz_hip_ba_optimize. */
int  ygz_hip_ba_optimize_chi2(ygz_hip_ctx *ctx, const ygz_ba_problem *pb, double *poses_io, double *points_io, int max_iterations,
                              ygz_ba_stats *stats, double *chi2_edge);
/* Which loop the last ygz_hip_ba_optimize / ygz_hip_ba_solve_ceres of the context ran.  Both route to the resident kernels
 * (ygz_hip_ba_optimize_resident / ygz_hip_ba_solve_ceres_resident) when the window has at most 20 free poses (14 for the ceres form) and no repeated (point, pose)
 * edge; otherwise the linearisations run on the GPU and the reduced system on the host -- the same results, about ten times slower.
 * Returns YGZ_BA_PATH_RESIDENT, or YGZ_BA_PATH_HOST_LOOP | the reason bits; 0 before the first call. */
enum { YGZ_BA_PATH_RESIDENT = 1, YGZ_BA_PATH_HOST_LOOP = 2, YGZ_BA_WHY_FREE_POSES = 16, YGZ_BA_WHY_REPEATED_EDGES = 32, YGZ_BA_WHY_FORCED = 64 };
int  ygz_hip_ba_last_path(const ygz_hip_ctx *ctx);
/* The resident LM's teams synchronise through a barrier in HBM; members that share an XCD take a light form of it (no L2 write-back).  The light
 * form is self-tested once per context before the first team launch (a message-passing litmus through the barrier itself): 1 = passed and in use,
 * 0 = failed on this device / partition mode (every barrier takes the full form), -1 = no team launch yet.  YGZ_LM_XCD_BARRIER=0 / 1 overrides. */
int  ygz_hip_ba_light_barrier(const ygz_hip_ctx *ctx);
/* The same loop entirely on the GPU for uploaded windows window_begin .. +n_windows-1 (formulation 0, at most 20 free
 * poses per window): one workgroup per window runs linearisation, Schur complement, Cholesky, back-substitution, update and
 * the lambda policy in HBM/LDS without a host round trip, all windows concurrently.  The windows' states are updated in
 * place (ygz_hip_ba_get_state reads them back); stats [n_windows] may be NULL (then the call is asynchronous). */
int  ygz_hip_ba_optimize_resident(ygz_hip_ctx *ctx, int window_begin, int n_windows, int max_iterations,
                                  ygz_ba_stats *stats);
/* A launch of few windows gives every window a TEAM of workgroups (up to 32; results are bit-identical for every team size) as long
 * as windows x team <= budget workgroups; each of them owns a CU for the whole launch.  Default (0): half of the device's CUs.  A
 * pipeline that runs the LM beside other kernels lowers it for those launches (a member that waits for a CU to drain keeps the
 * members already placed spinning) and restores it for the launch nothing else runs beside.  Clamped to [1, CUs]. */
int  ygz_hip_ba_set_team_budget(ygz_hip_ctx *ctx, int workgroups);
/* Where the members of a team run.  0 (default): one XCD per window -- its 32 CUs for the whole launch, the team barrier without an L2
 * write-back (fastest alone); 1: four CUs of every XCD -- the launch leaves seven eighths of every XCD to the kernels of other streams (a
 * pipeline that runs the LM BESIDE other work wants this: every kernel has workgroups on every XCD and waits for the one a team owns).
 * Results are bit-identical either way. */
int  ygz_hip_ba_set_team_placement(ygz_hip_ctx *ctx, int spread);
int  ygz_hip_ba_get_state(ygz_hip_ctx *ctx, int window, double *poses, double *points);
/* statistics of the last ygz_hip_ba_optimize_resident run on each window of the range (it may have been asynchronous) and the graph
 * sizes dims [n][4] = poses, points, edges, free poses; either output may be NULL.  YGZ_E_HIP: a team of workgroups timed out at a
 * barrier (its members were not resident together); YGZ_E_STATE: no resident run yet.  Synchronises. */
int  ygz_hip_ba_get_stats(ygz_hip_ctx *ctx, int window_begin, int n_windows, ygz_ba_stats *stats, int32_t *dims);

/* ---- the keyframe side of a batched run kept in HBM: what LocalMapping::LocalBA gathers on the host before ba::LocalBAG2O
 *      (src/Module/LocalMapping.cpp:149-208: the window's keyframes, their map points and observations; src/Algorithm/BA.cpp:397-470:
 *      vertices and edges) is assembled on the device from a store of keyframe rows, so a BA round moves no keypoint table and no graph
 *      across PCIe.  Contexts of one device share the store: tracking contexts copy their keyframes into it device to device. ---- */
/* order everything enqueued on `waiter` from now on behind everything enqueued on `signaler` so far (same device; no host wait) */
int  ygz_hip_stream_wait(ygz_hip_ctx *waiter, ygz_hip_ctx *signaler);
/* finer: ygz_hip_mark remembers what the context has enqueued so far (e.g. up to an upload); ygz_hip_wait_mark orders what `waiter`
 * enqueues from now on behind signaler's last mark only.  Uploads chained this way cross PCIe first-in-first-out at the full rate. */
int  ygz_hip_mark(ygz_hip_ctx *ctx);
int  ygz_hip_wait_mark(ygz_hip_ctx *waiter, ygz_hip_ctx *signaler);
/* host helper: T_out[0] = identity, T_out[i] = T_rel[i] * T_out[i - 1] (Sophus SE3 product; poses as 7 doubles): the trajectory of a
 * batched run from its per-pair relative poses (VisualOdometry.cpp:66 chains the same product frame by frame).  No device work. */
int  ygz_hip_se3_chain(const double *T_rel, int n, double *T_out);
/* bytes of one keyframe row for this context's grid: pixels f64 [cells][2] | depth f64 [cells] | level i32 [cells] | descriptors
 * [cells][32] | count i32 | (with_images != 0) the keyframe's pyramid, levels 0 .. pyramid_levels - 1 -- each part 64-byte aligned.
 * Rows are fixed-size so that the rows of other ranks arrive by ONE all-gather on the store's memory. */
size_t ygz_hip_kf_row_bytes(const ygz_hip_ctx *ctx, int with_images);
/* a store of n_keyframes rows (+ max_windows rows of work space behind them), the relative pose T_rel of n_frames frames (pose of
 * frame f in the frame of f - 1; (qx,qy,qz,qw,tx,ty,tz)).  rows_mem: device memory of the caller (>= (n_keyframes + max_windows) *
 * row bytes, e.g. a tensor its collectives can address) or NULL to let the library allocate.  with_images != 0: ygz_hip_kf_store_put also
 * copies the keyframe's pyramid into its row (what ygz_hip_ba_build_windows needs for obs_mode 1). */
int  ygz_hip_kf_store_create(ygz_hip_ctx *ctx, int n_keyframes, int n_frames, int max_windows, void *rows_mem, size_t rows_mem_bytes, int with_images);
int  ygz_hip_kf_store_info(ygz_hip_ctx *ctx, void **rows, size_t *row_bytes, void **trel, int *n_keyframes, int *n_frames);
/* row kf_index[i] <- the keypoints (Feature::_pixel, _level, _desc, _depth) of slot src_slot[i] of `src`; enqueued on src's stream */
int  ygz_hip_kf_store_put(ygz_hip_ctx *store, ygz_hip_ctx *src, int n, const int32_t *src_slot, const int32_t *kf_index);
/* T_rel[first_frame + i] <- the pose ygz_hip_track_pose_only left for pair first_pair + i of `src` (on src's stream); the host form
 * for frames tracked elsewhere (asynchronous on the store's stream) */
int  ygz_hip_kf_store_put_trel(ygz_hip_ctx *store, ygz_hip_ctx *src, int first_pair, int n_pairs, int first_frame);
int  ygz_hip_kf_store_set_trel(ygz_hip_ctx *ctx, int first_frame, int n, const double *T_rel);
/* call after a collective wrote rows into the store's memory (re-reads the rows' counts) */
int  ygz_hip_kf_store_refresh(ygz_hip_ctx *ctx);
/* BA windows whose graph the device builds: capacity K keyframes (pose 0 constant, BA.cpp:404) x max_points map points */
int  ygz_hip_ba_reserve_windows(ygz_hip_ctx *ctx, int window_begin, int n_windows, int K, int max_points, double huber_delta);
/* window i = the n_kfs[i] keyframes in store rows kf_index[i][0..K) = frames kf_frame[i][.] of the sequence (ascending; entry 0 is the
 * anchor).  Map points: the anchor's features with depth (first max_points in keypoint order), Pixel2Camera in the anchor's camera
 * (Camera.h:56-62); observations: the anchor's pixel plus, in every other keyframe of the window,
 *   obs_mode 0: the good cross-checked Hamming match of the anchor's descriptor (test/test_orb_match.cpp:86-104);
 *   obs_mode 1: what LocalMapping::ProjectMapPoints leaves there (src/Module/LocalMapping.cpp:47-120): the map point projected with the
 *               chained pose, kept if in front of the camera and InFrame(px, 20) (FindCandidates), refined by Matcher::FindDirectProjection
 *               (Matcher.cpp:356-383) from the anchor's image into the keyframe's; needs a store created with images;
 * points seen by fewer than two keyframes are dropped; vertex j = log of
 * T_rel(kf_frame[j]) * ... * T_rel(anchor + 1) as [omega; upsilon] (G2oTypes.h:88), the anchor at the identity.  Asynchronous;
 * ygz_hip_ba_optimize_resident / ygz_hip_ba_linearize_resident run on the result, ygz_hip_ba_get_stats returns the sizes. */
int  ygz_hip_ba_build_windows(ygz_hip_ctx *ctx, int window_begin, int n_windows, const int32_t *kf_index, const int32_t *kf_frame,
                              const int32_t *n_kfs, int obs_mode);
/* the inlier test after optimize() (src/Algorithm/BA.cpp:503-515): every enabled edge's chi2 (identity information, no robust kernel) at
 * the windows' current state against chi2_threshold (5.991); asynchronous, behind the resident LM of the same windows.  disable != 0:
 * outlier edges are switched off for later runs (the effect of Feature::_bad = true on the next LocalBAG2O, BA.cpp:436).
 * ygz_hip_ba_get_outlier_stats: out [n_windows][4] = edges tested, outliers, chi2 of the tested edges, chi2 of the inliers (-1: not
 * computed since the last resident LM); synchronises. */
int  ygz_hip_ba_mark_outliers(ygz_hip_ctx *ctx, int window_begin, int n_windows, double chi2_threshold, int disable);
int  ygz_hip_ba_get_outlier_stats(ygz_hip_ctx *ctx, int window_begin, int n_windows, double *out);
/* one row per window: [poses 6 K | points 3 max_points | K, P, E, iterations, lm_trials, chi2_initial, chi2_final, lambda_final | edges
 * tested, outliers, chi2 of the tested edges, chi2 of the inliers (ygz_hip_ba_mark_outliers)], unused entries 0, rows row_doubles
 * (>= 6 K + 3 max_points + 12) apart; dst in device memory (dst_on_device != 0: e.g. the buffer of the map exchange) or host memory */
int  ygz_hip_ba_pack_states(ygz_hip_ctx *ctx, int window_begin, int n_windows, double *dst, size_t row_doubles, int dst_on_device, int wait);


/* ---- B6/B7: ceres::Solve as the reference configures it (src/Algorithm/BA.cpp:219-226,372-375: default options =
 *      trust-region Levenberg-Marquardt, Jacobi scaling; :58-62 TwoViewBACeres: DOGLEG) around the GPU linearisation of a formulation-2 problem.
 *      ba::LocalBA, OptimizeCurrent, OptimizeCurrentPointOnly and TwoViewBACeres are this call on different edge lists. */
typedef struct {
    int    max_num_iterations;                       /* 50 */
    double function_tolerance, gradient_tolerance, parameter_tolerance;           /* 1e-6, 1e-10, 1e-8 */
    double initial_trust_region_radius, max_trust_region_radius, min_trust_region_radius;   /* 1e4, 1e16, 1e-32 */
    double min_relative_decrease, min_lm_diagonal, max_lm_diagonal;               /* 1e-3, 1e-6, 1e32 */
    int    jacobi_scaling, max_num_consecutive_invalid_steps;                     /* 1, 5 */
    int    fail_behind_camera;                       /* 1: evaluation fails when an enabled edge has p_z < 0 (PoseOnly) */
    int    trust_region_strategy;                    /* 0 LEVENBERG_MARQUARDT (default), 1 DOGLEG (TRADITIONAL_DOGLEG: ba::TwoViewBACeres, BA.cpp:59-60);
                                                        DOGLEG runs the loop on the host around the GPU linearisations (round 6) */
} ygz_ceres_options;
enum { YGZ_CERES_LEVENBERG_MARQUARDT = 0, YGZ_CERES_DOGLEG = 1 };
enum { YGZ_CERES_FUNCTION_TOLERANCE = 0, YGZ_CERES_GRADIENT_TOLERANCE, YGZ_CERES_PARAMETER_TOLERANCE, YGZ_CERES_MIN_RADIUS,
       YGZ_CERES_NO_CONVERGENCE, YGZ_CERES_FAILURE };
typedef struct {
    int    iterations, successful_steps, unsuccessful_steps, termination;
    double initial_cost, final_cost, final_radius;
} ygz_ceres_summary;
void ygz_hip_ceres_default_options(ygz_ceres_options *opt);
int  ygz_hip_ba_solve_ceres(ygz_hip_ctx *ctx, const ygz_ba_problem *pb, double *poses_io, double *points_io,
                            const ygz_ceres_options *opt, ygz_ceres_summary *summary);
/* The same loop entirely on the GPU for uploaded formulation-2 windows window_begin .. +n_windows-1 (at most 14 free and 16 poses per
 * window, no repeated (point, pose) pair): one workgroup per window runs every trust-region iteration (scaled Schur complement,
 * Cholesky, step validity, candidate cost, radius policy) in HBM / LDS, all windows concurrently; the windows' states are updated
 * in place (ygz_hip_ba_get_state).  summaries [n_windows] may be NULL (then the call is asynchronous).  ygz_hip_ba_solve_ceres
 * uses it when the window qualifies. */
int  ygz_hip_ba_solve_ceres_resident(ygz_hip_ctx *ctx, int window_begin, int n_windows, const ygz_ceres_options *opt,
                                     ygz_ceres_summary *summaries);

/* ---- B7: ba::OptimizeCurrentPoseOnly (src/Algorithm/BA.cpp:188-264; the per-frame call of
 *      LocalMapping::OptimizeCurrent, src/Module/LocalMapping.cpp:126) for a batch of frames: one workgroup per frame runs
 *      the four solve / re-classify rounds entirely on the GPU.  Frame f owns features frame_off[f] .. frame_off[f+1]-1:
 *      px [n][2] pixels (Feature::_pixel), pw [n][3] world points (MapPoint::_pos_world).  poses_io [n_frames][6] =
 *      [t; log(so3)] of _TCW in and out; bad [n] (Feature::_bad), depth [n] (Feature::_depth, written for inliers only,
 *      otherwise left), inliers / rounds [n_frames] (may be NULL).  Intrinsics are the context's camera. */
int  ygz_hip_optimize_pose_only(ygz_hip_ctx *ctx, int n_frames, const int32_t *frame_off, const double *px,
                                const double *pw, double *poses_io, uint8_t *bad, double *depth, int32_t *inliers,
                                int32_t *rounds);

/* ---- cvutils::DepthFromTriangulation (include/ygz/Algorithm/CVUtils.h:18-38) for n ray pairs: the step after
 *      SearchForTriangulation in LocalMapping::CreateNewMapPoints (src/Module/LocalMapping.cpp:430-452).  f_ref / f_cur [n][3]
 *      normalised rays, T_search_ref = (qx,qy,qz,qw,tx,ty,tz); ok[i] = 0 when det(A^T A) < determinant_th (depths untouched). */
int  ygz_hip_depth_from_triangulation(ygz_hip_ctx *ctx, const double T_search_ref[7], const double *f_ref, const double *f_cur,
                                      int n, double determinant_th, double *depth1, double *depth2, uint8_t *ok);

/* ---- the triangulation loop of LocalMapping::CreateNewMapPoints (src/Module/LocalMapping.cpp:416-495, the branch where neither
 *      feature has a map point yet), for n matched pairs (SearchForTriangulation's output) of keyframes in slot1 (current keyframe,
 *      pose T1) and slot2 (neighbour, T2): parallax test, DepthFromTriangulation, Matcher::FindDirectProjection of feature 1 into
 *      frame 2 with the triangulated depth, second triangulation with the refined pixel, reprojection test (5.991 px), map point.
 *      px1 [n][2] / level1 [n]: Feature::_pixel / _level in frame 1; px2 [n][2] in/out: fea2->_pixel (overwritten once the direct
 *      projection succeeded, as the reference does); code [n]: 0 map point created, 1 parallel rays, 2 / 4 first / second
 *      triangulation rejected, 3 direct projection failed, 5 reprojection error; depth1 / depth2 / pos_world [n][3] for code 0. */
int  ygz_hip_create_map_points(ygz_hip_ctx *ctx, int slot1, const double T1[7], int slot2, const double T2[7], int n, const double *px1,
                               const int32_t *level1, double *px2, int32_t *code, double *depth1, double *depth2, double *pos_world,
                               int32_t *search_level, int *n_created);

/* ---- the depth filter of the legacy tree: DepthFilter::UpdateSeeds / UpdateSeed / ComputeTau (src/optimizer.cpp:537-735) with
 *      utils::FindEpipolarMatchDirect (src/utils.cpp:330-661: epipolar ZMSSD search + the legacy Align2D + DepthFromTriangulation),
 *      for n seeds against ONE new frame (cur_slot, T_cur).  Seed i belongs to reference frame seed_ref[i] (ref_slot / T_refs
 *      [n_refs]) whose id for the age test is seed_frame_id[i]; kp [n][2] = cv::KeyPoint::pt, octave [n]; a, b, mu, sigma2 [n] are
 *      updated in place (Seed's Beta / Gaussian parameters, float), z_range [n].  state [n]: 0 updated and kept; 1 behind the camera,
 *      2 outside the frame, 3 no epipolar match (kept unchanged); 4 erased, older than max_n_kfs (5); 5 erased, converged
 *      (sqrt(sigma2) < z_range / convergence_sigma2_thresh (100)): pos_world [n][3] = the new map point; 6 erased, NaN.
 *      z [n] = matched depth, matched_px [n][2].  Needs >= 3 pyramid levels. */
int  ygz_hip_depth_filter_update(ygz_hip_ctx *ctx, int cur_slot, const double T_cur[7], int n_refs, const int32_t *ref_slot,
                                 const double *T_refs, int batch_counter, int max_n_kfs, double convergence_sigma2_thresh, int n,
                                 const float *kp, const int32_t *octave, const int32_t *seed_ref, const uint64_t *seed_frame_id,
                                 float *a, float *b, float *mu, const float *z_range, float *sigma2, int32_t *state, double *z,
                                 double *matched_px, double *pos_world, int *n_updated);

/* ---- M4 / M5: BoW-guided matching -- replaces Frame::ComputeBoW (src/Basic/Frame.cpp:190-201 ->
 *      DBoW3::Vocabulary::transform, thirdparty/DBoW3/src/Vocabulary.cpp:706-835), Matcher::SearchByBoW
 *      (src/Algorithm/Matcher.cpp:196-292) and Matcher::SearchForTriangulation (:86-193, epipolar test :338-354).
 *      The vocabulary is the file ORBVocabulary::loadFromBinaryFile reads (test/test_orb_match.cpp:72-74): header
 *      {u32 nb_nodes, u32 size_node, i32 k, i32 L, i32 scoring, i32 weighting}, then per node {i32 parent, u8 desc[32],
 *      f32 weight, u8 is_leaf}; exactly nb_nodes records are read. */
int  ygz_hip_vocab_load(ygz_hip_ctx *ctx, const void *blob, size_t bytes);
int  ygz_hip_vocab_info(ygz_hip_ctx *ctx, int *k, int *L, int *n_nodes, int *n_words);
/* ComputeBoW over the resident keypoints of a slot range: word id, weight, FeatureVector node (levelsup levels above the leaf;
 * -1 when the word is stopped, i.e. weight <= 0) per keypoint, kept in HBM */
int  ygz_hip_compute_bow(ygz_hip_ctx *ctx, int slot_begin, int n_slots, int levelsup);
int  ygz_hip_get_bow(ygz_hip_ctx *ctx, int slot, int32_t *word, double *weight, int32_t *node, int capacity, int *n);
/* the same for host descriptors [n][32] */
int  ygz_hip_bow_transform(ygz_hip_ctx *ctx, const uint8_t *desc, int n, int levelsup, int32_t *word, double *weight,
                           int32_t *node);
/* mode 0: SearchByBoW (best < th_low and best < knn_ratio * second best); mode 1: SearchForTriangulation (E12 row-major,
 * per pair in the slot form).  match12: index in frame 2 or -1 per feature of frame 1 ([n_pairs][max_keypoints] in the slot
 * form); count(s): number of matches.  Slot form: all pairs in one launch, BoW from ygz_hip_compute_bow. */
int  ygz_hip_search_by_bow_slots(ygz_hip_ctx *ctx, int mode, int n_pairs, const int32_t *slot1, const int32_t *slot2,
                                 const double *E12, int th_low, float knn_ratio, double epipolar_dsqr, int32_t *match12,
                                 int32_t *counts);
int  ygz_hip_search_by_bow(ygz_hip_ctx *ctx, int mode, const uint8_t *desc1, const int32_t *node1, const double *px1, int n1,
                           const uint8_t *desc2, const int32_t *node2, const double *px2, int n2, const double *E12,
                           int th_low, float knn_ratio, double epipolar_dsqr, int32_t *match12, int *count);

/* Matcher::Options::checkOrientation (Matcher.h:24; Matcher.cpp:247-256, 271-289, ComputeThreeMaxima :293-336) on the result of a mode-0
 * search: the 30-bin histogram of rot = angle1 - angle2 over the matches, its three maxima, and kept = the count SearchByBoW returns with
 * the option on (matches outside the three fullest bins are subtracted; the reference does not remove them from the map -- its TODO at
 * :284 -- so match12 is left alone).  SearchForTriangulation only fills the histogram (:157-165) and never reads it: nothing to call.
 * Host form: one pair, Feature::_angle as doubles; slot form: the pairs of ygz_hip_search_by_bow_slots, the resident keypoints' angles.
 * hist ([30] per pair) and maxima ([3] per pair, -1 = none) may be NULL. */
int  ygz_hip_bow_orientation(ygz_hip_ctx *ctx, const double *angle1, int n1, const double *angle2, int n2, const int32_t *match12,
                             int *kept, int32_t *hist, int32_t *maxima);
int  ygz_hip_bow_orientation_slots(ygz_hip_ctx *ctx, int n_pairs, const int32_t *slot1, const int32_t *slot2, const int32_t *match12,
                                   int32_t *kept, int32_t *hist, int32_t *maxima);

#ifdef __cplusplus
}
#endif
#endif /* YGZ_HIP_H_ */
