// ygz::PinholeCamera -- same surface as include/ygz/Basic/Camera.h:10-112 (float intrinsics, double maths).
#ifndef YGZ_CAMERA_H_
#define YGZ_CAMERA_H_
#include "ygz/Basic/Common.h"
#include "ygz/Basic/Config.h"
namespace ygz {
class PinholeCamera {
public:
    PinholeCamera()
    {
        _fx = Config::Get<float>("camera.fx"); _fy = Config::Get<float>("camera.fy");
        _cx = Config::Get<float>("camera.cx"); _cy = Config::Get<float>("camera.cy");
        _k1 = Config::Get<float>("camera.k1"); _k2 = Config::Get<float>("camera.k2");
        _p1 = Config::Get<float>("camera.p1"); _p2 = Config::Get<float>("camera.p2");
        _f = (_fx + _fy) / 2;
    }
    inline Vector3d World2Camera(const Vector3d &p_w, const SE3 &T_c_w) { return T_c_w * p_w; }
    inline Vector3d Camera2World(const Vector3d &p_c, const SE3 &T_c_w) { return T_c_w.inverse() * p_c; }
    inline Vector2d Camera2Pixel(const Vector3d &p_c) { return Vector2d(_fx * p_c[0] / p_c[2] + _cx, _fy * p_c[1] / p_c[2] + _cy); }
    inline Vector3d Pixel2Camera(const Vector2d &p_p, double depth = 1) { return Vector3d((p_p[0] - _cx) * depth / _fx, (p_p[1] - _cy) * depth / _fy, depth); }
    inline Vector2d Pixel2Camera2D(const Vector2d &p_p) { return Vector2d((p_p[0] - _cx) / _fx, (p_p[1] - _cy) / _fy); }
    inline Vector3d Pixel2World(const Vector2d &p_p, const SE3 &T_c_w, double depth = 1) { return Camera2World(Pixel2Camera(p_p, depth), T_c_w); }
    Vector2d World2Pixel(const Vector3d &p_w, const SE3 &T_c_w) { return Camera2Pixel(World2Camera(p_w, T_c_w)); }
    inline float fx() const { return _fx; }
    inline float fy() const { return _fy; }
    inline float cx() const { return _cx; }
    inline float cy() const { return _cy; }
    inline float focal() const { return _f; }
protected:
    float _fx, _fy, _cx, _cy, _f;
    float _k1, _k2, _p1, _p2;
};
}
#endif
