// ygz::hip::Runtime -- process-wide owner of the ygz_hip context behind the class surfaces: creates the context
// from ygz::Config on first use (image size, pyramid levels, grid cell, FAST threshold, intrinsics), maps Frames to
// HBM slots (least-recently-used eviction, transparent re-upload) and remembers which host pyramid level mirrors
// which (frame, level) so that cvutils::Align2D(const cv::Mat&, ...) can find the HBM copy of its image argument.
// Environment: YGZ_HIP_DEVICE (default 0), YGZ_HIP_MAX_FRAMES (default 64).
#ifndef YGZ_HIP_RUNTIME_H_
#define YGZ_HIP_RUNTIME_H_
#include <cstdint>
struct ygz_hip_ctx;
namespace ygz {
struct Frame;
namespace hip {
class Runtime {
public:
    static Runtime &Get();
    ygz_hip_ctx *ctx();                 // throws std::runtime_error if no usable gfx950 device (no CPU fallback)
    int cells();
    int Resident(Frame *f);             // HBM slot of f (uploads / rebuilds the pyramid if it was evicted)
    void Release(Frame *f);
    void RegisterLevels(Frame *f);
    bool FindLevel(const uint8_t *data, Frame **f, int *level);
private:
    Runtime();
    ~Runtime();
    struct Impl;
    Impl *p_;
};
void check(int rc, const char *what);
}
}
#endif
