/*
 * ygz_oracle.h -- CPU ORACLE for the ygz-slam per-frame hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a scalar, single-threaded, dependency-free C
 * restatement of the reference arithmetic (file:line citations on every function,
 * paths relative to the reference tree).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it, and only as the checker / the timed CPU
 * baseline.  The product path (ygz_slam_amd/, include/ygz_hip.h) never links it.
 *
 * PARITY STATUS: "parity unpinned" for the pieces whose arithmetic lives in
 * libraries that are absent from the reference tree and from this image
 * (uzh-rpg/fast, OpenCV >= 3.1, g2o, ceres, Eigen): the reference holds no golden
 * vectors or asserting tests (all of test/ only prints), and the reference cannot
 * be compiled here (every translation unit includes Eigen/OpenCV/g2o/ceres
 * headers).  Those pieces are frozen specifications of the published algorithms,
 * each marked [frozen spec] below.  Pieces that restate code that IS in the
 * reference tree (Matcher, CVUtils, SparseImageAlign, G2oTypes, Sophus) follow it
 * line by line and are pinned by the closed-form fixtures of test/test_local_ba.cpp.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off; no FMA contraction so float
 * results do not depend on the compiler).
 */
#ifndef YGZ_ORACLE_H_
#define YGZ_ORACLE_H_
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define YO_MAX_LEVELS 8

/* ---- image pyramid ------------------------------------------------------------ */
typedef struct {
    int levels;
    int w[YO_MAX_LEVELS], h[YO_MAX_LEVELS];
    uint8_t *img[YO_MAX_LEVELS];      /* continuous rows, stride == w (cv::Mat default) */
} yo_pyramid;

/* cv::cvtColor(CV_BGR2GRAY) 8u [frozen spec, OpenCV 3.1 RGB2Gray<uchar>]; Frame.cpp:27 */
void yo_bgr2gray(const uint8_t *bgr, int w, int h, int stride, uint8_t *gray);
/* cv::pyrDown 8u, 5x5 binomial, BORDER_REFLECT_101 [frozen spec]; Frame.cpp:38 */
void yo_pyr_down(const uint8_t *src, int w, int h, uint8_t *dst);
/* Frame::CreateImagePyramid (Frame.cpp:32-40): allocates levels 1.. with malloc */
void yo_pyramid_build(yo_pyramid *p, const uint8_t *gray, int w, int h, int levels);
void yo_pyramid_free(yo_pyramid *p);

/* ---- FAST-10 [frozen spec of uzh-rpg/fast]; call sites FeatureDetector.cpp:366-381 -- */
/* raster-order corner list over x in [3,w-3), y in [3,h-3); returns count (<= max) */
int  yo_fast10_detect(const uint8_t *img, int w, int h, int stride, int thr,
                      int16_t *xy, int max);
/* score = largest threshold at which the pixel is still a FAST-10 corner (bisection) */
void yo_fast10_score(const uint8_t *img, int stride, const int16_t *xy, int n, int thr,
                     int *scores);
/* closed form of the same score for one pixel (used to cross-check the bisection) */
int  yo_fast10_score_closed_form(const uint8_t *p, int stride);
/* 3x3 non-max suppression on a raster-ordered list.  tie_suppress=0: a corner is
 * dropped iff a neighbour has a strictly greater score (SURVEY 8c frozen choice);
 * tie_suppress=1: dropped on >= (the "strict maximum" variant).  returns count. */
int  yo_fast_nonmax_3x3(const int16_t *xy, const int *scores, int n, int tie_suppress,
                        int *nm_idx);

/* ---- extractor (FeatureDetector.cpp) ---------------------------------------------- */
typedef struct {
    int image_width, image_height;   /* FeatureDetector.h:51 (640x480) */
    int cell_size;                   /* default.yaml:50 (10) */
    double detection_threshold;      /* default.yaml:51 (15.0) */
    int pyramid_levels;              /* Frame.h:23 (3) */
    int nms_tie_suppress;            /* see yo_fast_nonmax_3x3 */
} yo_detect_params;

typedef struct {
    double px, py;       /* level-0 pixel (Feature::_pixel) */
    int    level;        /* Feature::_level */
    float  score;        /* Shi-Tomasi (Feature::_score holds this float) */
    float  angle;        /* degrees [0,360) (Feature::_angle holds this float) */
    uint8_t desc[32];    /* Feature::_desc */
} yo_keypoint;

void  yo_detect_params_default(yo_detect_params *p);
/* FeatureDetector::ShiTomasiScore FeatureDetector.cpp:467-507 */
float yo_shi_tomasi(const uint8_t *img, int w, int h, int stride, int u, int v);
/* cv::fastAtan2 [frozen spec, OpenCV 3.x polynomial]; FeatureDetector.cpp:536 */
float yo_fast_atan2(float y, float x);
/* FeatureDetector::IC_Angle FeatureDetector.cpp:509-537 (pt already divided by 2^level) */
float yo_ic_angle(const uint8_t *img, int w, int h, double ptx, double pty);
/* FeatureDetector::ComputeOrbDescriptor FeatureDetector.cpp:539-578 */
void  yo_orb_descriptor(const uint8_t *img, int w, int h, double px, double py, int level,
                        float angle_deg, uint8_t desc[32]);
/* FeatureDetector::Detect FeatureDetector.cpp:345-444.  occupied: grid_rows*grid_cols
 * bytes (non-zero = cell held by an old feature, SetExistingFeatures :446-464) or NULL.
 * out must hold grid_rows*grid_cols entries.  Returns the number of new features, in
 * cell-index order (the order they are pushed to frame->_features, :429-441). */
int   yo_detect(const yo_pyramid *pyr, const yo_detect_params *prm, const uint8_t *occupied,
                yo_keypoint *out);
/* per-level NMS corner dump for tests: returns count, fills xy (level coords) + FAST score */
int   yo_detect_level_corners(const uint8_t *img, int w, int h, int thr, int tie_suppress,
                              int16_t *xy, int *scores, int max);
/* FeatureDetector::ComputeAngleAndDescriptor :580-588 on given (px,py,level) */
void  yo_describe(const yo_pyramid *pyr, yo_keypoint *kps, int n);

/* ---- 256-bit Hamming matcher ---------------------------------------------------- */
/* Matcher::DescriptorDistance Matcher.cpp:30-43 (SWAR popcount on 8 x u32) */
int  yo_descriptor_distance(const uint8_t *a, const uint8_t *b);
/* nearest train row for every query row: first minimum on ties; also second-best
 * distance (to a different index; INT32_MAX if nt<2).  idx=-1 when nt==0. */
void yo_hamming_nn(const uint8_t *q, int nq, const uint8_t *t, int nt,
                   int32_t *idx, int32_t *dist, int32_t *dist2);
/* cv::BFMatcher(NORM_HAMMING, crossCheck).match(q, t) [frozen spec of OpenCV
 * batchDistance]; test/test_orb_match.cpp:86-93.  cross_check: 0 = plain NN,
 * 1 = OpenCV cross-check (for every train row its nearest query; each query keeps
 * the closest train row that voted for it), 2 = strict mutual nearest neighbours.
 * Outputs per query: train idx or -1, distance.  Returns number of matches. */
int  yo_bf_match(const uint8_t *q, int nq, const uint8_t *t, int nt, int cross_check,
                 int32_t *train_idx, int32_t *dist);
/* test_orb_match.cpp:97-104 post filter: min_dis clamped to [20,50], keep d < 3*min_dis.
 * keep[i] set for kept matches; returns count */
int  yo_good_match_filter(const int32_t *train_idx, const int32_t *dist, int nq, uint8_t *keep);
/* Matcher::CheckFrameDescriptors (src/Algorithm/Matcher.cpp:45-84) */
int  yo_check_frame_descriptors(const uint8_t *desc1, const uint8_t *desc2, const int32_t *idx1, const int32_t *idx2, int n,
                                int init_low, int init_high, float ratio, int32_t *dist, uint8_t *keep, int *best_out);

/* ---- SE3 / SO3 (thirdparty/Sophus/sophus/{so3,se3}.cpp) --------------------------- */
typedef struct { double q[4]; /* x,y,z,w */ double t[3]; } yo_se3;
void yo_so3_exp(const double w[3], double q[4], double *theta);         /* so3.cpp:178-202 */
void yo_so3_log(const double q[4], double w[3], double *theta);         /* so3.cpp:127-169 */
void yo_se3_identity(yo_se3 *T);
void yo_se3_exp(const double v[6], yo_se3 *T);      /* se3.cpp:170-196  v=[upsilon;omega] */
void yo_se3_log(const yo_se3 *T, double v[6]);      /* se3.cpp:198-220 */
void yo_se3_mul(const yo_se3 *A, const yo_se3 *B, yo_se3 *C);  /* se3.cpp:59-66 */
void yo_se3_inv(const yo_se3 *A, yo_se3 *B);        /* se3.cpp:77-84 */
void yo_se3_act(const yo_se3 *T, const double p[3], double out[3]);  /* se3.cpp:92-96 */
void yo_quat_to_R(const double q[4], double R[9]);  /* Eigen toRotationMatrix, row-major */

/* ---- camera (Basic/Camera.h): float intrinsics promoted to double in use ------------ */
typedef struct { float fx, fy, cx, cy; } yo_camera;
void yo_camera_default(yo_camera *c);               /* default.yaml:32-35 */

/* ---- patch alignment (CVUtils.cpp, Matcher.cpp) ------------------------------------- */
/* cvutils::Align2D CVUtils.cpp:186-318.  returns 1 on (converged && chi2<20000). */
int  yo_align2d(const uint8_t *cur, int w, int h, int stride,
                const uint8_t *ref_patch_with_border /*10x10*/, const uint8_t *ref_patch /*8x8*/,
                int n_iter, double *u, double *v, float *chi2_out, int *iters_out);
/* Matcher::GetWarpAffineMatrix Matcher.cpp:420-436; A row-major 2x2 */
void yo_warp_affine_matrix(const yo_camera *cam, const yo_se3 *T_ref_w,
                           const double px_ref[2], const double pt_ref[3], int level,
                           const yo_se3 *TCR, double A[4]);
int  yo_best_search_level(const double A[4], int max_level);     /* Matcher.h:123-134 */
/* Matcher::WarpAffine Matcher.cpp:438-466 */
void yo_warp_affine(const double A_cur_ref[4], const uint8_t *img_ref, int w, int h,
                    const double px_ref[2], int level_ref, int search_level,
                    int half_patch_size, uint8_t *patch);
/* Matcher::FindDirectProjection (Feature* overload) Matcher.cpp:385-417.
 * px_cur in/out (level-0 pixels).  returns 1/0. */
int  yo_find_direct_projection(const yo_camera *cam, const yo_pyramid *ref, const yo_se3 *T_ref,
                               const yo_pyramid *cur, const yo_se3 *T_cur,
                               const double px_ref[2], double depth_ref, int level_ref,
                               double px_cur[2], int *search_level);

int  yo_find_direct_projection_n(const yo_camera *cam, const yo_pyramid *ref, const yo_se3 *T_ref,
                                 const yo_pyramid *cur, const yo_se3 *T_cur, int n,
                                 const double *px_ref, const double *depth_ref, const int32_t *level_ref,
                                 double *px_cur, int32_t *search_level, uint8_t *ok);
/* MapPoint overload, Matcher.cpp:356-383 (depth = z of the point in the reference keyframe, no sign test) */
int  yo_find_direct_projection_mp(const yo_camera *cam, const yo_pyramid *ref, const yo_se3 *T_ref,
                                  const yo_pyramid *cur, const yo_se3 *T_cur, const double pos_world[3],
                                  const double px_ref[2], int level_ref, double px_cur[2], int *search_level);
/* LocalMapping::FindCandidates + ProjectMapPoints, LocalMapping.cpp:47-120 (candidates in caller order); see align.c */
/* LocalMapping::CreateNewMapPoints, triangulation loop (src/Module/LocalMapping.cpp:416-495): oracle/mapping.c */
int  yo_create_map_points(const yo_camera *cam, const yo_pyramid *pyr1, const yo_se3 *T1, const yo_pyramid *pyr2, const yo_se3 *T2, int n,
                          const double *px1, const int32_t *level1, double *px2, int32_t *code, double *depth1, double *depth2,
                          double *pos_world, int32_t *search_level);
/* legacy DepthFilter::UpdateSeeds (src/optimizer.cpp:537-735) + utils::FindEpipolarMatchDirect (src/utils.cpp:330-661): oracle/mapping.c */
int  yo_depth_filter_update(const yo_camera *cam, const yo_pyramid *refs, const yo_se3 *T_refs, const int32_t *frame_idx,
                            const uint64_t *frame_id, int batch_counter, int max_n_kfs, double convergence_sigma2_thresh,
                            const yo_pyramid *cur, const yo_se3 *T_cur, int n, const float *kp, const int32_t *octave,
                            float *a, float *b, float *mu, const float *z_range, float *sigma2,
                            int32_t *state, double *z_out, double *matched_px, double *pos_world);
int  yo_track_candidates(const yo_camera *cam, const yo_se3 *T_ref, const yo_se3 *T_cur, const double *px_ref, const double *depth,
                         int n, int w, int h, double *pos_world, double *px_pred, uint8_t *cand);
int  yo_track_local_map(const yo_camera *cam, const yo_pyramid *kf_pyr, const yo_se3 *kf_T, int K,
                        const yo_pyramid *cur, const yo_se3 *T_cur,
                        const double *pos_world, const uint8_t *point_bad, int P,
                        const int32_t *cand_point, const int32_t *cand_kf, const double *cand_px_ref, const int32_t *cand_level, int C,
                        uint8_t *in_view, double *px_proj, int32_t *match_cand, double *px_match, int32_t *match_level);

/* cvutils::DepthFromTriangulation CVUtils.h:18-38 */
int  yo_depth_from_triangulation(const yo_se3 *T_search_ref, const double f_ref[3], const double f_cur[3],
                                 double determinant_th, double *depth1, double *depth2);

/* ---- sparse image alignment (SparseImageAlign.cpp, NLSSolver_impl.hpp) --------------- */
typedef struct {
    int n_iter_total;        /* GN iterations executed over all levels */
    int n_meas_last;         /* n_meas_ of the last computeResiduals call */
    double chi2_last;
    int iters_per_level[YO_MAX_LEVELS];
} yo_sparse_align_stats;
/* SparseImgAlign::run SparseImageAlign.cpp:21-50 with GaussNewton (Matcher.cpp:18:
 * max_level 2, min_level 0, n_iter 30).  has_mappoint[i]!=0 <=> Feature::_mappoint.
 * T_cur_w in/out.  returns n_meas_/16. */
size_t yo_sparse_align(const yo_camera *cam, const yo_pyramid *ref, const yo_se3 *T_ref_w,
                       const yo_pyramid *cur, yo_se3 *T_cur_w,
                       const double *px /*[n][2]*/, const double *depth, const uint8_t *has_mappoint,
                       int n, int max_level, int min_level, int n_iter,
                       yo_sparse_align_stats *stats);
/* the same with method_ = LevenbergMarquardt (NLLSSolver::optimizeLevenbergMarquardt, NLSSolver_impl.hpp:91-212); stats->n_iter_total counts
 * trials, iters_per_level the value of iter_ at which each level's loop ended */
size_t yo_sparse_align_lm(const yo_camera *cam, const yo_pyramid *ref, const yo_se3 *T_ref_w,
                          const yo_pyramid *cur, yo_se3 *T_cur_w,
                          const double *px /*[n][2]*/, const double *depth, const uint8_t *has_mappoint,
                          int n, int max_level, int min_level, int n_iter,
                          yo_sparse_align_stats *stats);
/* one linearisation (computeResiduals with linearize_system=true) for kernel parity:
 * H 36 (row-major, symmetric), Jres 6, returns chi2/n_meas; visible in/out as the
 * reference keeps it across levels. */
double yo_sparse_align_linearize(const yo_camera *cam, const yo_pyramid *ref,
                                 const yo_pyramid *cur, const yo_se3 *T_cur_ref,
                                 const double *px, const double *depth, const uint8_t *has_mappoint,
                                 int n, int level, uint8_t *visible, double H[36], double Jres[6],
                                 int *n_meas);
/* 6x6 LDLT solve used by SparseImgAlign::solve (:225-231) [frozen spec of Eigen ldlt] */
int  yo_ldlt6_solve(const double H[36], const double b[6], double x[6]);

/* ---- pyramidal LK [frozen spec of cv::calcOpticalFlowPyrLK]; Tracker.cpp:92-98 -------- */
typedef struct {
    int win;          /* Tracker.h:25 (21) */
    int max_level;    /* Tracker.cpp:97 (4) */
    int max_iter;     /* Tracker.h:26 (30) */
    double eps;       /* Tracker.h:27 (0.001) */
    double min_eig_threshold;  /* OpenCV default 1e-4 */
    int use_initial_flow;      /* Tracker.cpp:97 OPTFLOW_USE_INITIAL_FLOW */
} yo_klt_params;
void yo_klt_params_default(yo_klt_params *p);
void yo_klt_track(const uint8_t *prev, const uint8_t *next, int w, int h,
                  const float *prev_pts /*[n][2]*/, float *next_pts /*[n][2] in/out*/, int n,
                  const yo_klt_params *prm, uint8_t *status, float *err);

/* ---- local BA edge stack (G2oTypes.h, g2o_types.h, BA.cpp) ---------------------------- */
typedef struct {
    int n_poses, n_points, n_edges;
    const double *poses;        /* [n_poses][6]  vertex estimate order [omega(3); t(3)] G2oTypes.h:88 */
    const uint8_t *pose_fixed;  /* [n_poses] */
    const double *points;       /* [n_points][3] */
    const int32_t *edge_pose;   /* [n_edges] */
    const int32_t *edge_point;  /* [n_edges] */
    const double *obs;          /* [n_edges][2] pixels */
    double fx, fy, cx, cy;      /* copied from float intrinsics G2oTypes.h:60-66 */
    double huber_delta;         /* BA.cpp:451 (5.991); <=0 disables the kernel */
} yo_ba_problem;
/* EdgeSophusSE3ProjectXYZ::computeError G2oTypes.h:84-91 */
void yo_ba_edge_error(const double pose[6], const double pt[3], const double obs[2],
                      double fx, double fy, double cx, double cy, double err[2]);
/* EdgeSophusSE3ProjectXYZ::linearizeOplus G2oTypes.h:93-132: Jp 2x3, Jx 2x6 row-major */
void yo_ba_edge_jacobians(const double pose[6], const double pt[3],
                          double fx, double fy, double Jp[6], double Jx[12]);
/* legacy normalised-plane edge include/ygz/g2o_types.h:45-86 (pose order [t; omega]) */
void yo_ba_edge_error_norm(const double pose_tw[6], const double pt[3], const double obs_n[2],
                           double err[2]);
void yo_ba_edge_jacobians_norm(const double pose_tw[6], const double pt[3], double Jp[6], double Jx[12]);
/* g2o constructQuadraticForm + RobustKernelHuber over all edges [frozen spec of g2o]:
 * Hpp [n_poses][36], bp [n_poses][6], Hll [n_points][9], bl [n_points][3],
 * Hpl [n_edges][18] (6x3 row-major = Jx^T w Jp), err [n_edges][2], chi2_edge [n_edges]
 * (raw e^T e), returns sum of robustified chi2 (rho[0]). Fixed poses get no Hpp/bp/Hpl. */
double yo_ba_linearize(const yo_ba_problem *pb, double *Hpp, double *bp, double *Hll, double *bl,
                       double *Hpl, double *err, double *chi2_edge);
/* VertexSE3Sophus::oplusImpl G2oTypes.h:38-45 */
void yo_ba_pose_oplus(double pose[6], const double upd[6]);

/* ---- BoW-guided matching (M4, M5): oracle/bow.c ------------------------------------------------------------ */
typedef struct {
    int k, L, scoring, weighting;     /* header of the DBoW3 binary vocabulary */
    int n_nodes, n_words;             /* nodes include the root (id 0) */
    int32_t *parent;                  /* [n_nodes] */
    uint8_t *desc;                    /* [n_nodes][32] */
    double  *weight;                  /* [n_nodes] */
    int32_t *word_id;                 /* [n_nodes] -1 for inner nodes */
    int32_t *child_off, *child;       /* CSR of the children, in push_back order */
} yo_vocab;
int  yo_vocab_parse(const void *blob, size_t bytes, yo_vocab *v);
void yo_vocab_free(yo_vocab *v);
void yo_bow_transform_one(const yo_vocab *v, const uint8_t *d, int levelsup, int32_t *word, double *weight, int32_t *nid);
int  yo_bow_transform(const yo_vocab *v, const uint8_t *desc, int n, int levelsup, int32_t *word, double *weight, int32_t *node,
                      int32_t *bow_word, double *bow_value);
int  yo_search_by_bow(const uint8_t *desc1, const int32_t *node1, int n1, const uint8_t *desc2, const int32_t *node2, int n2,
                      int th_low, float knn_ratio, int32_t *match12);
void  yo_set_exp_libm(int on);               /* test hook: DepthFilter::UpdateSeed with glibc's expf instead of include/ygz_exp.h */
float yo_expf_shared(float x);               /* (float)ygz_exp_nonpos((double)x): the exponential the oracle and the device share */
void  yo_update_seed(float x, float tau2, float *a, float *b, float *mu, float z_range, float *sigma2);
int  yo_bow_orientation(const double *angle1, const double *angle2, const int32_t *match12, int n1, int32_t *hist, int32_t *ind);
int  yo_search_for_triangulation(const yo_camera *cam, const uint8_t *desc1, const int32_t *node1, const double *px1, int n1,
                                 const uint8_t *desc2, const int32_t *node2, const double *px2, int n2,
                                 const double E12[9], int th_low, double epipolar_dsqr, int32_t *match12);

/* ---- ceres-side rows (B3, B6, B7) and the two NLLS drivers: oracle/ceres_ba.c ---------------------------- */
typedef struct {
    int n_poses, n_points, n_edges;
    double *poses;              /* [n_poses][6] = [t; angle-axis] (BA.cpp:96-99); yo_ceres_solve updates it */
    const uint8_t *pose_fixed;  /* constant pose = CeresReprojectionErrorPointOnly's embedded _TCW; NULL: none */
    double *points;             /* [n_points][3] */
    const uint8_t *point_fixed; /* constant point = CeresReprojectionErrorPoseOnly's embedded _pt_world; NULL: none */
    const int32_t *edge_pose, *edge_point;
    const double *obs_n;        /* [n_edges][2] normalised image coordinates (Camera::Pixel2Camera2D) */
    const double *edge_huber;   /* [n_edges] ceres::HuberLoss(a); <= 0 or NULL: no loss function */
    const uint8_t *edge_enable; /* SetEnable; NULL: all enabled */
    int fail_behind_camera;     /* PoseOnly functor: evaluation fails when p_z < 0 */
} yo_ceres_problem;
void yo_ceres_rotate_point(const double aa[3], const double p[3], double out[3]);
void yo_ceres_edge(const double pose[6], const double pt[3], const double obs_n[2],
                   double r[2], double Jpose[12], double Jpt[6], double *p_z);
int  yo_ceres_linearize(const yo_ceres_problem *pb, const double *poses, const double *points, double *cost,
                        double *Hpp, double *bp, double *Hll, double *bl, double *Hpl,
                        double *Jx_out, double *Jp_out, double *rc_out);
int  yo_ba_schur_solve(int K, int P, int E, const int32_t *edge_pose, const int32_t *edge_point,
                       const uint8_t *pose_free, const uint8_t *point_free,
                       const double *Hpp, const double *Hll, const double *Hpl, const double *bp, const double *bl,
                       const double *dp, const double *dl, double *xp, double *xl);
typedef struct {
    int max_num_iterations;
    double function_tolerance, gradient_tolerance, parameter_tolerance;
    double initial_trust_region_radius, max_trust_region_radius, min_trust_region_radius;
    double min_relative_decrease, min_lm_diagonal, max_lm_diagonal;
    int jacobi_scaling, max_num_consecutive_invalid_steps;
    int trust_region_strategy;  /* 0 LEVENBERG_MARQUARDT (ceres' default), 1 DOGLEG (TRADITIONAL_DOGLEG: what ba::TwoViewBACeres asks for, BA.cpp:60) */
} yo_ceres_options;
enum { YO_CERES_LEVENBERG_MARQUARDT = 0, YO_CERES_DOGLEG = 1 };
enum { YO_CERES_FUNCTION_TOLERANCE = 0, YO_CERES_GRADIENT_TOLERANCE, YO_CERES_PARAMETER_TOLERANCE, YO_CERES_MIN_RADIUS,
       YO_CERES_NO_CONVERGENCE, YO_CERES_FAILURE };
typedef struct {
    int iterations, successful_steps, unsuccessful_steps, termination;
    double initial_cost, final_cost, final_radius;
} yo_ceres_summary;
void yo_ceres_default_options(yo_ceres_options *o);
int  yo_ceres_solve(yo_ceres_problem *pb, const yo_ceres_options *opt, yo_ceres_summary *sum);
typedef struct { int iterations, lm_trials; double chi2_initial, chi2_final, lambda_final; } yo_lm_stats;
int  yo_g2o_lm(const yo_ba_problem *pb, double *poses, double *points, int max_iterations, yo_lm_stats *stats);
/* ba::TwoViewBACeres (BA.cpp:11-89), ba::OptimizeCurrent (:91-186), ba::OptimizeCurrentPointOnly (:266-322) on plain arrays */
int  yo_two_view_ba_ceres(const yo_camera *cam, const yo_se3 *ref, yo_se3 *curr, int n, const double *px_ref, const double *px_curr,
                          uint8_t *inlier, double *pts_ref, yo_ceres_summary *sum);
int  yo_optimize_current(const yo_camera *cam, yo_se3 *T_cur, int n, const double *px, const int32_t *feat_point, int P, double *points,
                         int K, const yo_se3 *kf_T, const int32_t *obs_off, const int32_t *obs_kf, const double *obs_px,
                         uint8_t *bad, double *depth, yo_ceres_summary *sum);
int  yo_optimize_current_point_only(const yo_camera *cam, const yo_se3 *T_cur, int n, const double *px, const int32_t *feat_point,
                                    const uint8_t *feat_bad, int P, double *points, int K, const yo_se3 *kf_T, const int32_t *obs_off,
                                    const int32_t *obs_kf, const double *obs_px, yo_ceres_summary *sum);
int  yo_optimize_current_pose_only(const yo_camera *cam, double pose_io[6], int n, const double *px, const double *pw,
                                   uint8_t *bad_out, double *depth_out, int *rounds_run);

#ifdef __cplusplus
}
#endif
#endif
