// ygz::Frame -- include/ygz/Basic/Frame.h:20-166: the hot-path part, and (round 6) the covisibility members src/Module reads
// (UpdateConnections, GetBestCovisibilityKeyframes, ...: host bookkeeping over std::map, src/Basic/Frame.cpp:73-176).  InitFrame() uploads the image to an HBM slot of the process-wide ygz_hip context and
// builds the pyramid on the GPU.  _pyramid keeps the reference's spelling (_pyramid[L], .size(), .empty()): it is
// a LAZY host mirror -- level L crosses PCIe the first time somebody indexes it (round 5: InitFrame used to
// drag all levels, 400 KB per VGA frame, back to the host whether or not anybody read them).
#ifndef YGZ_FRAME_H_
#define YGZ_FRAME_H_
#include "ygz/Basic/Common.h"
namespace ygz {
class PinholeCamera;
struct MapPoint;
struct Feature;
struct Frame;
namespace hip {
// what `vector<cv::Mat> _pyramid` (Basic/Frame.h:138) is for the callers -- indexable, sized -- with the download deferred to the first
// access of a level.  The levels themselves live in HBM (ygz::hip::Runtime).
class PyramidMirror {
public:
    size_t size() const { return lv_.size(); }
    bool empty() const { return lv_.empty(); }
    void clear() { lv_.clear(); have_.clear(); }
    cv::Mat &operator[](size_t L) { fetch(L); return lv_[L]; }
    const cv::Mat &operator[](size_t L) const { const_cast<PyramidMirror *>(this)->fetch(L); return lv_[L]; }
    bool fetched(size_t L) const { return L < have_.size() && have_[L]; }
    void reset(Frame *owner, size_t levels) { owner_ = owner; lv_.assign(levels, cv::Mat()); have_.assign(levels, 0); }
private:
    void fetch(size_t L);                   // ygz_host.cpp: ygz_hip_download_level of the owner's slot (re-uploading an evicted frame first)
    Frame *owner_ = nullptr;
    std::vector<cv::Mat> lv_;
    std::vector<char> have_;
};
}
struct Frame {
    struct Option { int _pyramid_level = 3; } _option;
    Frame() {}
    Frame(const Frame &) = delete;
    Frame operator=(const Frame &) = delete;
    ~Frame();
    static void SetCamera(PinholeCamera *camera) { _camera = camera; }
    static PinholeCamera *GetCamera() { return _camera; }
    static void SetORBVocabulary(ORBVocabulary *orb_vocab) { _vocab = orb_vocab; }      // Frame.h:105-108
    void ComputeBoW();                                          // src/Basic/Frame.cpp:190-201
    void InitFrame();                                           // src/Basic/Frame.cpp:22-30
    inline Vector3d Pos() const { return _TCW.inverse().translation(); }
    inline bool InFrame(const Vector2d &pixel, const int &boarder = 10) const
    { return pixel[0] >= boarder && pixel[0] < _color.cols - boarder && pixel[1] >= boarder && pixel[1] < _color.rows - boarder; }
    inline bool InFrame(const cv::Point2f &pixel, const int &boarder = 10) const
    { return pixel.x >= boarder && pixel.x < _color.cols - boarder && pixel.y >= boarder && pixel.y < _color.rows - boarder; }
    inline bool InFrame(const Vector2d &pixel, const int &boarder, const int &level) const
    { return pixel[0] / (1 << level) >= boarder && pixel[0] / (1 << level) < _color.cols - boarder
          && pixel[1] / (1 << level) >= boarder && pixel[1] / (1 << level) < _color.rows - boarder; }
    Vector3d GetCamCenter() const { return _TCW.inverse().translation(); }
    bool GetMeanAndMinDepth(double &mean_depth, double &min_depth);      // src/Basic/Frame.cpp:42-72
    vector<Frame *> GetBestCovisibilityKeyframes(const int &N = 10);     // src/Basic/Frame.cpp:73-78: the first N of _cov_keyframes
    bool IsInFrustum(MapPoint *mp, float viewingCosLimit = 0.5);         // src/Basic/Frame.cpp:80-84 (always true in the reference)
    void AddConnection(Frame *kf, const int &weight);                    // src/Basic/Frame.cpp:154-160
    void UpdateConnections();                                            // src/Basic/Frame.cpp:86-152
    void UpdateBestCovisibles();                                         // src/Basic/Frame.cpp:162-176
    cv::Mat GetAllDescriptors();                                // src/Basic/Frame.cpp:178-188
    void CleanAllFeatures();                                    // src/Basic/Frame.cpp:203-210
    unsigned long _id = 0, _keyframe_id = 0;
    double _timestamp = 0;
    SE3    _TCW = SE3();
    bool   _is_keyframe = false;
    vector<Feature *> _features;
    Mat    _color, _depth;                  // _color: CV_8UC3 (BGR) or CV_8UC1
    hip::PyramidMirror _pyramid;            // lazy host mirror of the HBM levels (Basic/Frame.h:138: vector<cv::Mat>)
    static PinholeCamera *_camera;
    static ORBVocabulary *_vocab;
    DBoW3::BowVector _bow_vec;
    DBoW3::FeatureVector _feature_vec;
    Frame *_ref_keyframe = nullptr;
    bool   _bad = false;
    map<Frame *, int> _connected_keyframe_weights;      // keyframes that share map points with this one -> number shared (Frame.h:147)
    vector<Frame *> _cov_keyframes;                     // the same, heaviest first (Frame.h:150-151)
    vector<int> _cov_weights;
    int    _hip_slot = -1;                  // HBM slot of this frame (managed by ygz::hip::Runtime)
    void CreateImagePyramid();              // fetches every level into the host mirror now
};
}
#endif
