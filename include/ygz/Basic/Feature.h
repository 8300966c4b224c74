// ygz::Feature -- same fields as include/ygz/Basic/Feature.h:15-36.
#ifndef YGZ_FEATURE_H_
#define YGZ_FEATURE_H_
#include "ygz/Basic/Common.h"
namespace ygz {
struct Frame;
struct MapPoint;
struct Feature {
    Feature(const Vector2d &pixel, const int &level = 0, const double &score = 0) : _pixel(pixel), _level(level), _score(score) {}
    // `new Feature` / `delete` of the callers, served from a fixed-size block pool (Common.h: ygz::pool)
    static void *operator new(size_t n) { return n == sizeof(Feature) ? pool::FixedPool<(sizeof(Feature) + 15) / 16 * 16>::alloc() : ::operator new(n); }
    static void operator delete(void *p, size_t n) { if (!p) return; if (n == sizeof(Feature)) pool::FixedPool<(sizeof(Feature) + 15) / 16 * 16>::release(p); else ::operator delete(p); }
    Vector2d  _pixel = Vector2d(0, 0);
    double    _depth = -1;
    Vector3d  _normal = Vector3d(0, 0, 0);
    int       _level = -1;
    double    _angle = 0;
    Mat       _desc = cv::Mat(1, 32, CV_8UC1);
    Frame    *_frame = nullptr;
    MapPoint *_mappoint = nullptr;
    bool      _bad = false;
    double    _score = 0;
};
}
#endif
