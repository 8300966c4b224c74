// ygz::ba -- local bundle adjustment with the surface of include/ygz/Algorithm/BA.h:23-66.
// LocalBAG2O reproduces ba::LocalBAG2O (src/Algorithm/BA.cpp:386-543): graph build, Levenberg-Marquardt with
// Schur complement (g2o BlockSolver_6_3 restated), 20 iterations, Huber delta 5.991, chi2 > 5.991 -> Feature::_bad.
// Every linearisation (residuals, Jacobians, Hpp/Hll/Hpl/b blocks, robust chi2) runs on the GPU; the reduced
// 6K x 6K system is solved on the host.  The ceres-based variants (TwoViewBACeres, OptimizeCurrent*, LocalBA) are
// not part of this build (ceres autodiff/trust-region internals are out of the hot-path scope).
#ifndef YGZ_BA_H_
#define YGZ_BA_H_
#include "ygz/Basic.h"
namespace ygz {
namespace ba {
struct LocalBAStats { int iterations = 0, lm_trials = 0, outliers = 0; double chi2_initial = 0, chi2_final = 0; };
void LocalBAG2O(std::set<Frame *> &local_keyframes, std::set<MapPoint *> &local_map_points);
void LocalBAG2O(std::set<Frame *> &local_keyframes, std::set<MapPoint *> &local_map_points, LocalBAStats *stats);
}
}
#endif
