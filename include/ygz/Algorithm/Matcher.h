// ygz::Matcher -- same surface as include/ygz/Algorithm/Matcher.h:15-152.  Brute-force matching is offered as
// BruteForceMatch (what test/test_orb_match.cpp:86-93 does with cv::BFMatcher).  The BoW-guided searches run on the GPU
// against the vocabulary loaded through ORBVocabulary::loadFromBinaryFile (the reference does not ship vocab/ORBvoc.bin).
#ifndef YGZ_MATCHER_H_
#define YGZ_MATCHER_H_
#include "ygz/Basic/Common.h"
namespace ygz {
struct Frame;
struct MapPoint;
struct Feature;
class SparseImgAlign;
struct DMatch { int queryIdx = -1, trainIdx = -1; float distance = 0; };
class Matcher {
public:
    struct Options {
        int th_high = 100, th_low = 50;
        float knnRatio = 0.9f;
        bool checkOrientation = false;
        float initMatchRatio = 3.0f;
        int init_low = 30, init_high = 80;
        double _max_alignment_motion = 0.2;
        double _epipolar_dsqr = 1e-4;
    } _options;
    static const int HISTO_LENGTH = 30;
    Matcher();
    ~Matcher();
    void SetTCR(const SE3 &TCR) { _TCR_esti = TCR; }
    static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b);                      // Matcher.cpp:30-43
    int CheckFrameDescriptors(Frame *frame1, Frame *frame2, list<pair<int, int>> &matches);  // Matcher.cpp:45-84
    int SearchByBoW(Frame *kf1, Frame *kf2, map<int, int> &matches);                          // Matcher.cpp:196-292
    int SearchForTriangulation(Frame *kf1, Frame *kf2, const Matrix3d &E12, vector<pair<int, int>> &matched_points,
                               const bool &onlyStereo = false);                               // Matcher.cpp:86-193
    // cv::BFMatcher(NORM_HAMMING, crossCheck).match(frame1 descriptors, frame2 descriptors) on the GPU
    int BruteForceMatch(Frame *frame1, Frame *frame2, vector<DMatch> &matches, bool cross_check = true);
    bool FindDirectProjection(Frame *ref, Frame *curr, MapPoint *mp, Vector2d &px_curr, int &search_level);   // Matcher.cpp:356-383
    bool FindDirectProjection(Frame *ref, Frame *curr, Feature *fea_ref, Vector2d &px_curr, int &search_level); // Matcher.cpp:385-417
    // the same for many features of `ref` in one launch (what LocalMapping::ProjectMapPoints loops over)
    int FindDirectProjectionBatch(Frame *ref, Frame *curr, const vector<Feature *> &fea_ref, vector<Vector2d> &px_curr,
                                  vector<int> &search_level, vector<bool> &ok);
    // LocalMapping::FindCandidates + ProjectMapPoints (src/Module/LocalMapping.cpp:47-120) in one launch: projects the local map
    // points into `current`, refines every co-visible observation with FindDirectProjection (MapPoint overload) and appends one
    // new Feature per matched map point to current->_features; side effects as the reference (_track_in_view = false when out
    // of view, _cnt_visible++ when in view).  Candidates are visited per map point in _obs (keyframe id) order -- the
    // reference's order is the heap-address order of a std::map<Feature*, ...>.  Returns the number of matched points.
    int ProjectMapPoints(Frame *current, const std::set<Frame *> &local_keyframes, const std::set<MapPoint *> &local_map_points);
    bool SparseImageAlignment(Frame *ref, Frame *current);                                  // Matcher.cpp:468-492
    SE3 GetTCR() const { return _TCR_esti; }
private:
    SparseImgAlign *_align;
    SE3 _TCR_esti;
    // the last SearchForTriangulation of this object (frames and matched feature index pairs): what the Feature overload of FindDirectProjection speculates
    // on when LocalMapping::CreateNewMapPoints calls it once per matched pair (src/Module/LocalMapping.cpp:398-447)
    Frame *_tri_kf1 = nullptr, *_tri_kf2 = nullptr;
    vector<pair<int, int>> _tri_pairs;
};
}
#endif
