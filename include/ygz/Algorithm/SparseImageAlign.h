// ygz::SparseImgAlign -- same constructor and run() as include/ygz/Algorithm/SparseImageAlign.h:10-58; the whole
// Gauss-Newton loop runs on the GPU (ygz_hip_sparse_align).
#ifndef YGZ_SPARSE_IMAGE_ALIGN_
#define YGZ_SPARSE_IMAGE_ALIGN_
#include "ygz/Basic.h"
namespace ygz {
class SparseImgAlign {
public:
    enum Method { GaussNewton, LevenbergMarquardt };
    SparseImgAlign(int n_levels, int min_level, int n_iter, Method method, bool display, bool verbose);
    size_t run(Frame *ref_frame, Frame *cur_frame);
    int iterations(int level) const { return level >= 0 && level < 8 ? iters_[level] : 0; }
private:
    int max_level_, min_level_, n_iter_;
    int iters_[8] = { 0 };
};
}
#endif
