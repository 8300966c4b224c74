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
// status of an ABI call, the reference's way (SURVEY 8b "error conventions": bool / count returns and a glog line, no exceptions): YGZ_OK -> true;
// anything else -> LOG(ERROR) with the ABI's message and false, and the surface that called returns its failure value (false / 0 / unchanged
// outputs).  Only a missing device -- there is no CPU path to fall back to -- throws (Runtime::ctx).
bool check(int rc, const char *what);
// Matcher::FindDirectProjection behind per-candidate callers (ygz_host.cpp: FdpMemo): one speculative launch per current frame, answers handed out
// only on bit-equal inputs.  Environment YGZ_FDP_MEMO=0 (or SetFdpSpeculation(false)) makes every call its own n = 1 launch.  When the previous
// current frame was served that way, the launch of the next one is queued at the end of Matcher::SparseImageAlignment (its pose is known there) and
// collected at its first per-candidate call -- the caller's own FindCandidates runs in between; YGZ_FDP_PRELAUNCH=0: launch at the first call.
struct FdpMemoStats { unsigned long long hits = 0, single = 0, launches = 0, speculated = 0; double speculate_ms = 0; };
void SetFdpSpeculation(bool on);
void SetFdpBypass(bool on);             // true: calls take their own n = 1 launch and leave the memo as it is (to compare the two inside one loop)
FdpMemoStats GetFdpMemoStats();
void ResetFdpMemoStats();
}
}
#endif
