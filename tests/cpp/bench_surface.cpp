// The reference-shaped loop, ONE FRAME AT A TIME, through the ygz:: class surfaces -- what the unchanged callers of the reference do
// (test/test_vo_track.cpp:100-113 -> VisualOdometry::AddFrame, src/Module/VisualOdometry.cpp:38-107, state VO_GOOD):
//
//     Frame::InitFrame                                   (src/Basic/Frame.cpp:22-40)
//     _curr->_TCW = _ref->_TCW; TrackRefFrame            (VisualOdometry.cpp:66-67,281-302: Matcher::SparseImageAlignment)
//     TrackLocalMap                                      (LocalMapping.cpp:24-45: FindCandidates + ProjectMapPoints + OptimizeCurrent =
//                                                         ba::OptimizeCurrentPoseOnly)
//     FeatureDetector::Detect(frame, false)              (extraction on every frame -- the metric's "extract"; the reference detects on
//                                                         keyframes only, VisualOdometry.cpp:SetKeyframe)
//     every kf_stride-th frame: keyframe + ba::LocalBAG2O over keyframe 0 + the newest `local_kfs` - 1 keyframes and the map points they observe
//                                                         (LocalMapping.cpp:149-208,301-336)
//
// New map points come from a depth image at the keyframes (the stand-in for the initialiser / triangulation, as in the offline run).
// Built as a shared object: bench.py (`surface` block) and tests/test_gpu_surface.py call ygz_bench_surface through ctypes and run the oracle
// on the same loop for the CPU side.  This file is test / bench infrastructure: the product is libygz_host.so behind the headers it includes.
#include "ygz/Basic.h"
#include "ygz/Algorithm.h"
#include "ygz/hip/Runtime.h"
#include <chrono>
#include <deque>
using namespace ygz;

namespace {
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- TrackLocalMap's first two steps the way the reference's UNCHANGED caller performs them: LocalMapping::FindCandidates and
// LocalMapping::ProjectMapPoints (src/Module/LocalMapping.cpp:47-120), with reference-named methods only -- one Matcher::FindDirectProjection
// (MapPoint overload) per candidate.  Same containers as the caller: candidates in a std::map keyed by Feature* (visited in heap-address order),
// the matched map points in a std::set.
struct UnchangedCaller {
    Matcher *_matcher;
    std::set<Frame *> *_local_keyframes;
    std::set<MapPoint *> *_local_map_points;
    bool _verify = false;                                 // every FindDirectProjection call repeated as its own n = 1 launch and compared bit for bit
    long _calls = 0, _mismatches = 0;
    bool _clock = false;                                  // (diagnostic run: host clock around FindCandidates and around every FindDirectProjection call)
    double _t_find = 0, _t_calls = 0;

    std::map<Feature *, Vector2d> FindCandidates(Frame *current)
    {
        std::map<Feature *, Vector2d> candidates;
        for (MapPoint *map_point : *_local_map_points) {
            if (map_point->_bad) continue;
            const Vector3d pt_curr = current->_camera->World2Camera(map_point->_pos_world, current->_TCW);
            const Vector2d px_curr = current->_camera->Camera2Pixel(pt_curr);
            if (pt_curr[2] < 0 || !current->InFrame(px_curr, 20)) { map_point->_track_in_view = false; continue; }
            map_point->_cnt_visible++;
            for (auto &obs_pair : map_point->_obs)
                if (_local_keyframes->find(obs_pair.second->_frame) != _local_keyframes->end()) candidates[obs_pair.second] = px_curr;
        }
        return candidates;
    }
    int ProjectMapPoints(Frame *current, std::map<Feature *, Vector2d> &candidates)
    {
        std::set<MapPoint *> matched_mps;
        for (auto &candidate : candidates) {
            MapPoint *mp = candidate.first->_mappoint;
            if (matched_mps.find(mp) != matched_mps.end()) continue;
            int level = 0;
            const Vector2d px_in = candidate.second;
            const double tc_ = _clock ? now_ms() : 0;
            const bool ret = _matcher->FindDirectProjection(candidate.first->_frame, current, mp, candidate.second, level);
            if (_clock) _t_calls += now_ms() - tc_;
            ++_calls;
            if (_verify) {
                Vector2d px2 = px_in; int level2 = 0;
                hip::SetFdpBypass(true);
                const bool ret2 = _matcher->FindDirectProjection(candidate.first->_frame, current, mp, px2, level2);
                hip::SetFdpBypass(false);
                if (ret2 != ret || level2 != level || memcmp(px2.data(), candidate.second.data(), 16) != 0) ++_mismatches;
            }
            if (!ret) continue;
            matched_mps.insert(mp);
            Feature *feature = new Feature(candidate.second, level, candidate.first->_score);
            feature->_frame = current;
            feature->_mappoint = mp;
            current->_features.push_back(feature);
        }
        return (int)matched_mps.size();
    }
    int TrackLocalMap(Frame *current)                     // LocalMapping.cpp:24-33 (the pose-only BA that follows is the loop's next stage)
    {
        const double t0_ = _clock ? now_ms() : 0;
        std::map<Feature *, Vector2d> candidates = FindCandidates(current);
        if (_clock) _t_find += now_ms() - t0_;
        return ProjectMapPoints(current, candidates);
    }
};
}

extern "C" {

// bgr [n][h][w][3]; kf_depth [ceil(n / kf_stride)][h][w] float metres (depth image of frames 0, kf_stride, 2 kf_stride ...).
// Outputs: ms [n] host clock per frame (whole iteration, keyframe work included); T_out [n][7] the pose each frame ended with; counts [n][4] =
// map points used by the sparse alignment of this frame (features of the reference frame with a map point), features ProjectMapPoints created,
// features left after the pose-only inlier test, features after Detect; ba [n_kf][4] = iterations, map points, final chi2, ms of every LocalBAG2O.
// Returns 0, or 1 when an exception crossed a surface (message on stderr).
// stage_ms [8] (may be NULL): host clock summed over the frames: InitFrame, SparseImageAlignment, ProjectMapPoints, OptimizeCurrentPoseOnly, Detect,
// keyframe bookkeeping, LocalBAG2O, frame deletion.
// caller: 0 = TrackLocalMap through the batch method Matcher::ProjectMapPoints (one launch; a method the reference does not have);
//         1 = through reference-named methods only: FindCandidates + one Matcher::FindDirectProjection per candidate (UnchangedCaller above);
//         2 = as 1 with the speculative launch behind FindDirectProjection switched off (every call its own n = 1 launch);
//         3 = as 1, and every call is repeated as its own n = 1 launch and compared bit for bit (memo [5] counts the differences);
//         4 = as 1 with the host clock around FindCandidates and every FindDirectProjection call (memo [6..8]; the clock reads cost ~0.1 ms per frame).
// memo [9] (may be NULL): FindDirectProjection calls answered from the speculative launch, calls that took an n = 1 launch, speculative launches,
// candidates they evaluated, calls the caller made, calls whose memoised answer differed from the n = 1 launch (caller 3), ms in FindCandidates
// (the caller's own containers), ms inside the FindDirectProjection calls, of which ms in the speculative launches (gather + launch + table).
int ygz_bench_surface2(const uint8_t *bgr, const float *kf_depth, int n, int w, int h, int kf_stride, int local_kfs, int caller, double *ms, double *T_out,
                       int32_t *counts, double *ba, double *stage_ms, double *memo)
{
    double st_[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#define STAGE(k, code) { const double ts_ = now_ms(); code; st_[k] += now_ms() - ts_; }
    try {
        Config::Set("image.width", std::to_string(w)); Config::Set("image.height", std::to_string(h));
        PinholeCamera cam;
        Frame::SetCamera(&cam);
        FeatureDetector detector;
        detector.LoadParams();
        Matcher matcher;
        UnchangedCaller lm = { &matcher, nullptr, nullptr };
        lm._verify = caller == 3; lm._clock = caller == 4;
        hip::SetFdpSpeculation(caller != 2);
        hip::ResetFdpMemoStats();
        Memory::Clean();
        std::deque<Frame *> kf_order;
        std::vector<Frame *> all_kfs;
        std::vector<MapPoint *> all_mps;
        std::set<Frame *> local_keyframes;
        std::set<MapPoint *> local_map_points;
        lm._local_keyframes = &local_keyframes; lm._local_map_points = &local_map_points;
        Frame *ref = nullptr;
        const size_t fb = (size_t)w * h * 3;
        for (int i = 0; i < n; ++i) {
            const double t0 = now_ms();
            Frame *cur = new Frame;
            cur->_id = (unsigned long)i;
            cur->_color = cv::Mat(h, w, CV_8UC3, const_cast<uint8_t *>(bgr + (size_t)i * fb));
            STAGE(0, cur->InitFrame())
            int n_sa = 0, n_proj = 0, n_inl = 0;
            if (ref) {
                cur->_TCW = ref->_TCW;                                             // VisualOdometry.cpp:66
                for (Feature *f : ref->_features) n_sa += f->_mappoint != nullptr;
                STAGE(1, matcher.SparseImageAlignment(ref, cur))                   // TrackRefFrame
                if (caller == 0) STAGE(2, n_proj = matcher.ProjectMapPoints(cur, local_keyframes, local_map_points))      // TrackLocalMap: FindCandidates + ProjectMapPoints
                else STAGE(2, n_proj = lm.TrackLocalMap(cur))
                if (!cur->_features.empty()) STAGE(3, ba::OptimizeCurrentPoseOnly(cur))     // LocalMapping::OptimizeCurrent
                // an outlier of the pose-only BA stops being an observation of its map point (the reference keeps the pointer and a depth of -1,
                // which its next SparseImageAlignment would back-project: Feature.h:23, SparseImageAlign.cpp:78-83)
                for (Feature *f : cur->_features) { if (f->_bad) f->_mappoint = nullptr; else ++n_inl; }
            }
            STAGE(4, detector.Detect(cur, ref == nullptr))                         // new features where no tracked feature sits (SetExistingFeatures)
            const double t_kf = now_ms();
            double t_ba = 0;
            if (i % kf_stride == 0) {
                // SetKeyframe: the frame enters the map; features without a map point get one from the depth image
                Memory::RegisterKeyFrame(cur);
                all_kfs.push_back(cur);
                const float *D = kf_depth + (size_t)(i / kf_stride) * w * h;
                for (Feature *f : cur->_features) {
                    if (f->_mappoint) { if (!f->_bad) f->_mappoint->_obs[cur->_keyframe_id] = f; continue; }
                    const double d = D[(size_t)(int)f->_pixel[1] * w + (int)f->_pixel[0]];
                    if (!(d > 0)) continue;
                    MapPoint *mp = Memory::CreateMapPoint();
                    all_mps.push_back(mp);
                    mp->_pos_world = cam.Pixel2World(f->_pixel, cur->_TCW, d);
                    mp->_obs[cur->_keyframe_id] = f;
                    mp->_first_seen = mp->_last_seen = cur->_keyframe_id;
                    f->_mappoint = mp; f->_depth = d;
                }
                kf_order.push_back(cur);
                // the local map = keyframe 0 + the newest local_kfs - 1 keyframes.  The reference picks local keyframes by covisibility; the camera of the
                // synthetic sequence hovers around its first pose, so keyframe 0 stays covisible -- and Matcher::GetWarpAffineMatrix (Matcher.cpp:424-431,
                // reproduced as written) is only right for a reference keyframe at the origin: with keyframe 0 first in candidate order most map points
                // keep being matched for the whole sequence instead of for its first 3 keyframes
                if ((int)kf_order.size() > local_kfs) kf_order.erase(kf_order.begin() + 1);
                local_keyframes = std::set<Frame *>(kf_order.begin(), kf_order.end());
                local_map_points.clear();
                std::set<MapPoint *> ba_points;                                   // a point seen by one keyframe only constrains nothing: it is tracked, not optimised
                for (Frame *kf : kf_order) for (Feature *f : kf->_features) if (f->_mappoint && !f->_mappoint->_bad && !f->_bad) {
                    local_map_points.insert(f->_mappoint);
                    if (f->_mappoint->_obs.size() >= 2) ba_points.insert(f->_mappoint);
                }
                double *b = ba + 4 * (size_t)(i / kf_stride);
                b[0] = b[1] = b[2] = b[3] = 0;
                if (kf_order.size() >= 2 && !ba_points.empty()) {
                    ba::LocalBAStats st;
                    const double tb = now_ms();
                    ba::LocalBAG2O(local_keyframes, ba_points, &st);
                    b[0] = st.iterations; b[1] = (double)ba_points.size(); b[2] = st.chi2_final; b[3] = now_ms() - tb; t_ba = b[3];
                }
            }
            st_[5] += now_ms() - t_kf - t_ba; st_[6] += t_ba;
            cur->_TCW.to7(T_out + 7 * (size_t)i);
            counts[4 * i] = n_sa; counts[4 * i + 1] = n_proj; counts[4 * i + 2] = n_inl; counts[4 * i + 3] = (int32_t)cur->_features.size();
            STAGE(7, if (ref && !ref->_is_keyframe) delete ref)                    // VisualOdometry.cpp:88-89
            ref = cur;
            ms[i] = now_ms() - t0;
        }
        if (ref && !ref->_is_keyframe) delete ref;
        for (Frame *kf : all_kfs) delete kf;
        for (MapPoint *mp : all_mps) delete mp;
        Memory::Clean();
        Frame::SetCamera(nullptr);
        if (stage_ms) for (int k = 0; k < 8; ++k) stage_ms[k] = st_[k];
        if (memo) { const hip::FdpMemoStats ms_ = hip::GetFdpMemoStats(); memo[0] = (double)ms_.hits; memo[1] = (double)ms_.single; memo[2] = (double)ms_.launches; memo[3] = (double)ms_.speculated;
                    memo[4] = (double)lm._calls; memo[5] = (double)lm._mismatches; memo[6] = lm._t_find; memo[7] = lm._t_calls; memo[8] = ms_.speculate_ms; }
        hip::SetFdpSpeculation(true);
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "ygz_bench_surface: %s\n", e.what());
        hip::SetFdpSpeculation(true);
        return 1;
    }
}

int ygz_bench_surface(const uint8_t *bgr, const float *kf_depth, int n, int w, int h, int kf_stride, int local_kfs, double *ms, double *T_out,
                      int32_t *counts, double *ba, double *stage_ms)
{ return ygz_bench_surface2(bgr, kf_depth, n, w, h, kf_stride, local_kfs, 0, ms, T_out, counts, ba, stage_ms, nullptr); }

}  // extern "C"
