// ygz::ba -- local bundle adjustment with the surface of include/ygz/Algorithm/BA.h:23-66.
// LocalBAG2O reproduces ba::LocalBAG2O (src/Algorithm/BA.cpp:386-543): graph build, Levenberg-Marquardt with
// Schur complement (g2o BlockSolver_6_3 restated), 20 iterations, Huber delta 5.991, chi2 > 5.991 -> Feature::_bad.
// The whole loop -- linearisations (residuals, Jacobians, Hpp/Hll/Hpl/b blocks, robust chi2), Schur complement, Cholesky of the
// reduced 6K x 6K system, update and lambda policy -- runs on the GPU (ygz_hip_ba_optimize -> k_ba_lm_team) for windows of up to
// 14 free poses without repeated (point, pose) edges; other windows keep the linearisations on the GPU and solve the reduced system
// on the host.
// The ceres-based entry points (BA.cpp:11-384) are the same edge stack in the ceres parametrisation (pose = [t; angle-axis],
// normalised observations, additive update; C ABI formulation 2) under ceres' default trust-region Levenberg-Marquardt
// (ygz_hip_ba_solve_ceres); OptimizeCurrentPoseOnly -- the per-frame call of LocalMapping -- runs all four rounds in one
// kernel (ygz_hip_optimize_pose_only).  TwoViewBACeres asks ceres for DOGLEG (BA.cpp:59); it is run with the LM strategy.
#ifndef YGZ_BA_H_
#define YGZ_BA_H_
#include "ygz/Basic.h"
namespace ygz {
namespace ba {
struct LocalBAStats { int iterations = 0, lm_trials = 0, outliers = 0; double chi2_initial = 0, chi2_final = 0; };
void LocalBAG2O(std::set<Frame *> &local_keyframes, std::set<MapPoint *> &local_map_points);
void LocalBAG2O(std::set<Frame *> &local_keyframes, std::set<MapPoint *> &local_map_points, LocalBAStats *stats);
// BA.h:23-60 of the reference
void TwoViewBACeres(const SE3 &ref, SE3 &curr, const vector<Vector2d> px_ref, const vector<Vector2d> px_curr,
                    vector<bool> &inlier, vector<Vector3d> &pts_ref);
void OptimizeCurrent(Frame *current);
void OptimizeCurrentPoseOnly(Frame *current);
void OptimizeCurrentPointOnly(Frame *current);
void LocalBA(std::set<Frame *> &local_keyframes, std::set<MapPoint *> &local_map_points);
// batched form of OptimizeCurrentPoseOnly: one launch for all frames (throughput mode, not in the reference)
void OptimizeCurrentPoseOnlyBatch(const vector<Frame *> &frames);
}
}
#endif
