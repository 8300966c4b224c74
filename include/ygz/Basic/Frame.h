// ygz::Frame -- the hot-path part of include/ygz/Basic/Frame.h:20-166 (the covisibility graph is out of scope,
// SURVEY 2.1 #1).  InitFrame() uploads the image to an HBM slot of the process-wide ygz_hip context,
// builds the pyramid on the GPU and mirrors the levels into _pyramid for host readers.
#ifndef YGZ_FRAME_H_
#define YGZ_FRAME_H_
#include "ygz/Basic/Common.h"
namespace ygz {
class PinholeCamera;
struct MapPoint;
struct Feature;
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
    cv::Mat GetAllDescriptors();                                // src/Basic/Frame.cpp:178-188
    void CleanAllFeatures();                                    // src/Basic/Frame.cpp:203-210
    unsigned long _id = 0, _keyframe_id = 0;
    double _timestamp = 0;
    SE3    _TCW = SE3();
    bool   _is_keyframe = false;
    vector<Feature *> _features;
    Mat    _color, _depth;                  // _color: CV_8UC3 (BGR) or CV_8UC1
    vector<Mat> _pyramid;                   // host mirror of the HBM levels
    static PinholeCamera *_camera;
    static ORBVocabulary *_vocab;
    DBoW3::BowVector _bow_vec;
    DBoW3::FeatureVector _feature_vec;
    Frame *_ref_keyframe = nullptr;
    bool   _bad = false;
    int    _hip_slot = -1;                  // HBM slot of this frame (managed by ygz::hip::Runtime)
    void CreateImagePyramid();
};
}
#endif
