// ygz::SparseImgAlign -- same constructor and run() as include/ygz/Algorithm/SparseImageAlign.h:10-58.  Method GaussNewton (what the reference's
// only caller asks for, Matcher.cpp:18): the whole loop runs on the GPU (ygz_hip_sparse_align).  Method LevenbergMarquardt
// (NLLSSolver::optimizeLevenbergMarquardt, NLSSolver_impl.hpp:91-212; round 6): the solver's bookkeeping runs on the host, every
// computeResiduals(model, ...) of a trial is one launch (ygz_hip_sparse_align_residuals) -- correct and ~30 x slower than the resident loop; nobody in
// the reference calls it.
#ifndef YGZ_SPARSE_IMAGE_ALIGN_
#define YGZ_SPARSE_IMAGE_ALIGN_
#include "ygz/Basic.h"
namespace ygz {
class SparseImgAlign {
public:
    enum Method { GaussNewton, LevenbergMarquardt };
    SparseImgAlign(int n_levels, int min_level, int n_iter, Method method, bool display, bool verbose);
    size_t run(Frame *ref_frame, Frame *cur_frame);
    int iterations(int level) const { return level >= 0 && level < 8 ? iters_[level] : 0; }     // iterations a level took (LM: the value of iter_ at which its loop ended)
    int trials() const { return trials_; }                                                         // LM: computeResiduals trials of the last run, all levels
private:
    size_t run_lm(Frame *ref_frame, Frame *cur_frame);
    int max_level_, min_level_, n_iter_;
    Method method_ = GaussNewton;
    int iters_[8] = { 0 };
    int trials_ = 0;
};
}
#endif
