// ygz::cvutils -- the live entry points of include/ygz/Algorithm/CVUtils.h (Align2D :166-172 and the helpers the
// modules call).  Align2D runs on the GPU against the HBM copy of `cur_img`, which must be a pyramid level of a
// frame that went through Frame::InitFrame().
#ifndef YGZ_CVUTILS_H_
#define YGZ_CVUTILS_H_
#include "ygz/Basic/Common.h"
namespace ygz {
namespace cvutils {
bool Align2D(const cv::Mat &cur_img, uint8_t *ref_patch_with_border, uint8_t *ref_patch, const int n_iter,
             Vector2d &cur_px_estimate, bool no_simd = false);
// many patches against the same level image in one launch; pwb [n][100], px [n] in/out
int Align2DBatch(const cv::Mat &cur_img, const uint8_t *ref_patches_with_border, int n, const int n_iter,
                 vector<Vector2d> &cur_px_estimate, vector<bool> &ok);
// CVUtils.h:18-38
bool DepthFromTriangulation(const SE3 &T_search_ref, const Vector3d &f_ref, const Vector3d &f_cur, double &depth1,
                            double &depth2, const double &determinant_th = 1e-5);
}
}
#endif
