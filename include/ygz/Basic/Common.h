// ygz/Basic/Common.h -- the types the ygz-slam class surfaces are written against.
//
// The reference's Common.h (include/ygz/Basic/Common.h:21-72) pulls in Eigen, Sophus, OpenCV, glog, ceres, g2o and
// DBoW3.  None of them is available to this build, and the hot path needs only a sliver of each, so this header
// provides that sliver under the SAME NAMES (Vector2d, Vector3d, Matrix3d, Vector6d, SO3, SE3, cv::Mat, cv::Point2f,
// LOG(...)) so that code shaped like src/Module/*.cpp and test/*.cpp compiles against include/ygz/ unchanged.
// Numerics of SO3/SE3 follow thirdparty/Sophus/sophus/{so3,se3}.cpp operation by operation (se3_dev.h).
#ifndef YGZ_COMMON_INCLUDE_H_
#define YGZ_COMMON_INCLUDE_H_

#include <vector>
#include <array>
#include <list>
#include <memory>
#include <string>
#include <iostream>
#include <sstream>
#include <set>
#include <unordered_map>
#include <map>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cassert>

using namespace std;      // the reference does this in Common.h:17; kept because its callers rely on it

// ------------------------------------------------------------------------------------------ small fixed vectors
namespace ygz_math {
template <int N> struct Vec {
    double d[N];
    Vec() { for (int i = 0; i < N; ++i) d[i] = 0; }
    Vec(double a, double b) { static_assert(N == 2, "Vec2"); d[0] = a; d[1] = b; }
    Vec(double a, double b, double c) { static_assert(N == 3, "Vec3"); d[0] = a; d[1] = b; d[2] = c; }
    double &operator[](int i) { return d[i]; }
    const double &operator[](int i) const { return d[i]; }
    double &operator()(int i) { return d[i]; }
    const double &operator()(int i) const { return d[i]; }
    double &operator()(int i, int) { return d[i]; }
    const double &operator()(int i, int) const { return d[i]; }
    double x() const { return d[0]; }
    double y() const { return d[1]; }
    double z() const { static_assert(N >= 3, "z"); return d[2]; }
    Vec operator+(const Vec &o) const { Vec r; for (int i = 0; i < N; ++i) r.d[i] = d[i] + o.d[i]; return r; }
    Vec operator-(const Vec &o) const { Vec r; for (int i = 0; i < N; ++i) r.d[i] = d[i] - o.d[i]; return r; }
    Vec operator-() const { Vec r; for (int i = 0; i < N; ++i) r.d[i] = -d[i]; return r; }
    Vec operator*(double s) const { Vec r; for (int i = 0; i < N; ++i) r.d[i] = d[i] * s; return r; }
    Vec operator/(double s) const { Vec r; for (int i = 0; i < N; ++i) r.d[i] = d[i] / s; return r; }
    Vec &operator+=(const Vec &o) { for (int i = 0; i < N; ++i) d[i] += o.d[i]; return *this; }
    Vec &operator*=(double s) { for (int i = 0; i < N; ++i) d[i] *= s; return *this; }
    double dot(const Vec &o) const { double s = 0; for (int i = 0; i < N; ++i) s += d[i] * o.d[i]; return s; }
    double squaredNorm() const { return dot(*this); }
    double norm() const { return std::sqrt(squaredNorm()); }
    static Vec Zero() { return Vec(); }
    const Vec &transpose() const { return *this; }             // only ever streamed (LOG << v.transpose()): prints as a row already
    template <int M> Vec<M> head() const { Vec<M> r; for (int i = 0; i < M; ++i) r.d[i] = d[i]; return r; }
    template <int M> Vec<M> tail() const { Vec<M> r; for (int i = 0; i < M; ++i) r.d[i] = d[N - M + i]; return r; }
    const double *data() const { return d; }
    double *data() { return d; }
};
template <int N> inline Vec<N> operator*(double s, const Vec<N> &v) { return v * s; }
template <int N> inline std::ostream &operator<<(std::ostream &os, const Vec<N> &v)
{ for (int i = 0; i < N; ++i) os << v.d[i] << (i + 1 < N ? " " : ""); return os; }

struct Matrix3d {
    double m[9];      // row-major
    Matrix3d() { for (double &v : m) v = 0; }
    double &operator()(int r, int c) { return m[3 * r + c]; }
    const double &operator()(int r, int c) const { return m[3 * r + c]; }
    Vec<3> operator*(const Vec<3> &v) const
    { return Vec<3>(m[0] * v[0] + m[1] * v[1] + m[2] * v[2], m[3] * v[0] + m[4] * v[1] + m[5] * v[2], m[6] * v[0] + m[7] * v[1] + m[8] * v[2]); }
    Matrix3d operator*(const Matrix3d &o) const
    { Matrix3d r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double v = 0; for (int k = 0; k < 3; ++k) v += (*this)(i, k) * o(k, j); r(i, j) = v; } return r; }
    Matrix3d transpose() const { Matrix3d r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r(i, j) = (*this)(j, i); return r; }
    static Matrix3d Identity() { Matrix3d r; r(0, 0) = r(1, 1) = r(2, 2) = 1; return r; }
};
struct Matrix4d {                     // what SE3::matrix() returns; the callers only print it
    double m[16];
    Matrix4d() { for (double &v : m) v = 0; }
    double &operator()(int r, int c) { return m[4 * r + c]; }
    const double &operator()(int r, int c) const { return m[4 * r + c]; }
};
inline std::ostream &operator<<(std::ostream &os, const Matrix3d &M)
{ for (int r = 0; r < 3; ++r) os << M(r, 0) << " " << M(r, 1) << " " << M(r, 2) << (r < 2 ? "\n" : ""); return os; }
inline std::ostream &operator<<(std::ostream &os, const Matrix4d &M)
{ for (int r = 0; r < 4; ++r) os << M(r, 0) << " " << M(r, 1) << " " << M(r, 2) << " " << M(r, 3) << (r < 3 ? "\n" : ""); return os; }
struct Matrix2d {
    double m[4];
    Matrix2d() { for (double &v : m) v = 0; }
    double &operator()(int r, int c) { return m[2 * r + c]; }
    const double &operator()(int r, int c) const { return m[2 * r + c]; }
    double determinant() const { return m[0] * m[3] - m[2] * m[1]; }
};
}  // namespace ygz_math

using Vector2d = ygz_math::Vec<2>;
using Vector3d = ygz_math::Vec<3>;
typedef ygz_math::Vec<6> Vector6d;
using Matrix2d = ygz_math::Matrix2d;
using Matrix3d = ygz_math::Matrix3d;
using Matrix4d = ygz_math::Matrix4d;
namespace Eigen {                     // the spellings src/Module uses with the namespace written out (LocalMapping.cpp:406)
using Vector2d = ygz_math::Vec<2>; using Vector3d = ygz_math::Vec<3>; using Matrix2d = ygz_math::Matrix2d;
using Matrix3d = ygz_math::Matrix3d; using Matrix4d = ygz_math::Matrix4d;
}

// ------------------------------------------------------------------------------------------ Sophus (non-template)
namespace Sophus {
class SO3 {
public:
    SO3() { q_[0] = q_[1] = q_[2] = 0; q_[3] = 1; }
    static SO3 exp(const Vector3d &omega);                      // so3.cpp:171-202
    Vector3d log() const;                                       // so3.cpp:113-169
    SO3 inverse() const;
    SO3 operator*(const SO3 &o) const;
    Vector3d operator*(const Vector3d &p) const;
    Matrix3d matrix() const;
    static Matrix3d hat(const Vector3d &v)                      // so3.cpp:204-211
    { Matrix3d O; O(0, 1) = -v[2]; O(0, 2) = v[1]; O(1, 0) = v[2]; O(1, 2) = -v[0]; O(2, 0) = -v[1]; O(2, 1) = v[0]; return O; }
    const double *quat() const { return q_; }                   // x,y,z,w
    double q_[4];
};
class SE3 {
public:
    SE3() { t_[0] = t_[1] = t_[2] = 0; }
    SE3(const SO3 &so3, const Vector3d &t) : so3_(so3) { t_[0] = t[0]; t_[1] = t[1]; t_[2] = t[2]; }
    static SE3 exp(const Vector6d &upsilon_omega);              // se3.cpp:170-196
    Vector6d log() const;                                       // se3.cpp:198-220
    SE3 inverse() const;                                        // se3.cpp:77-84
    SE3 operator*(const SE3 &o) const;                          // se3.cpp:59-66
    Vector3d operator*(const Vector3d &p) const;                // se3.cpp:92-96
    Vector3d translation() const { return Vector3d(t_[0], t_[1], t_[2]); }
    Matrix3d rotation_matrix() const { return so3_.matrix(); }
    Matrix4d matrix() const                                     // se3.cpp:98-106: [R t; 0 1]
    { Matrix4d M; const Matrix3d R = so3_.matrix(); for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) M(r, c) = R(r, c); M(r, 3) = t_[r]; } M(3, 3) = 1; return M; }
    const SO3 &so3() const { return so3_; }
    void to7(double out[7]) const { for (int i = 0; i < 4; ++i) out[i] = so3_.q_[i]; for (int i = 0; i < 3; ++i) out[4 + i] = t_[i]; }
    static SE3 from7(const double in[7]) { SE3 T; for (int i = 0; i < 4; ++i) T.so3_.q_[i] = in[i]; for (int i = 0; i < 3; ++i) T.t_[i] = in[4 + i]; return T; }
    SO3 so3_;
    double t_[3];
};
std::ostream &operator<<(std::ostream &os, const SE3 &T);
}  // namespace Sophus
using Sophus::SO3;
using Sophus::SE3;

// ------------------------------------------------------------------------------------------ cv (the sliver in use)
#define CV_8U 0
#define CV_8UC1 0
#define CV_8UC3 16
#define CV_32F 5
// ------------------------------------------------------------------------------------------ fixed-size block pool
// A frame of the per-frame loop creates ~1400 Feature objects (+ the 32-byte descriptor block of each) and deletes as many of the frame before: through
// malloc / free that was ~0.2 ms of a 1.5 ms frame on the GPU box (the blocks interleave with long-lived objects, so the allocator's fast bins do not
// serve them).  Blocks of one size from thread-local free lists instead: `new Feature` / `delete` in the callers stay as they are (class-level
// operators), a block freed by another thread joins that thread's list, slabs are never handed back.
namespace ygz { namespace pool {
template <size_t SZ> struct FixedPool {
    static_assert(SZ % 16 == 0 && SZ >= 16, "block size");
    static void *alloc() { void *&h = head(); if (!h) refill(h); void *p = h; h = *reinterpret_cast<void **>(p); return p; }
    static void release(void *p) { void *&h = head(); *reinterpret_cast<void **>(p) = h; h = p; }
private:
    static void *&head() { static thread_local void *h = nullptr; return h; }
    static void refill(void *&h)
    {
        constexpr int N = 256;
        char *slab = static_cast<char *>(::operator new(SZ * N));
        for (int i = N - 1; i >= 0; --i) { void *b = slab + (size_t)i * SZ; *reinterpret_cast<void **>(b) = h; h = b; }
    }
};
template <class T> struct Alloc {                    // std allocator over FixedPool for single objects (std::allocate_shared's combined block)
    using value_type = T;
    Alloc() = default;
    template <class U> Alloc(const Alloc<U> &) {}
    T *allocate(size_t n) { return n == 1 ? static_cast<T *>(FixedPool<(sizeof(T) + 15) / 16 * 16>::alloc()) : static_cast<T *>(::operator new(n * sizeof(T))); }
    void deallocate(T *p, size_t n) { if (n == 1) FixedPool<(sizeof(T) + 15) / 16 * 16>::release(p); else ::operator delete(p); }
    template <class U> bool operator==(const Alloc<U> &) const { return true; }
    template <class U> bool operator!=(const Alloc<U> &) const { return false; }
};
} }  // namespace ygz::pool

namespace cv {
struct Point2f { float x = 0, y = 0; Point2f() {} Point2f(float a, float b) : x(a), y(b) {} };
struct Point { int x = 0, y = 0; };
struct Size { int width = 0, height = 0; Size() {} Size(int w, int h) : width(w), height(h) {} };
// reference-counted dense 2-D array (8-bit or float, 1 or 3 channels)
class Mat {
public:
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int type, void *ext, size_t step_bytes = 0)
        : rows(r), cols(c), data((uint8_t *)ext), type_(type) { step = step_bytes ? step_bytes : (size_t)c * elemSize(); }
    void create(int r, int c, int type)
    {
        rows = r; cols = c; type_ = type; step = (size_t)c * elemSize();
        const size_t bytes = (size_t)r * step;
        if (bytes <= 64) {                                   // a descriptor row (Feature::_desc, 32 bytes): block and control block in ONE allocation
            auto blk = std::allocate_shared<std::array<uint8_t, 128>>(ygz::pool::Alloc<std::array<uint8_t, 128>>());
            buf_ = std::shared_ptr<uint8_t>(blk, blk->data());
        } else buf_ = std::shared_ptr<uint8_t>(new uint8_t[bytes + 64], std::default_delete<uint8_t[]>());
        data = buf_.get();
    }
    size_t elemSize() const { return type_ == CV_8UC3 ? 3 : (type_ == CV_32F ? 4 : 1); }
    int type() const { return type_; }
    int channels() const { return type_ == CV_8UC3 ? 3 : 1; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    Size size() const { return Size(cols, rows); }
    template <typename T> T &at(int r, int c) { return *reinterpret_cast<T *>(data + (size_t)r * step + (size_t)c * sizeof(T)); }
    template <typename T> const T &at(int r, int c) const { return *reinterpret_cast<const T *>(data + (size_t)r * step + (size_t)c * sizeof(T)); }
    template <typename T> T *ptr(int r = 0) { return reinterpret_cast<T *>(data + (size_t)r * step); }
    template <typename T> const T *ptr(int r = 0) const { return reinterpret_cast<const T *>(data + (size_t)r * step); }
    Mat clone() const { Mat m; if (!empty()) { m.create(rows, cols, type_); for (int r = 0; r < rows; ++r) memcpy(m.data + r * m.step, data + r * step, (size_t)cols * elemSize()); } return m; }
    void copyTo(Mat &dst) const { dst = clone(); }
    Mat row(int r) const { Mat m(1, cols, type_, data + (size_t)r * step, step); m.buf_ = buf_; return m; }
    int rows = 0, cols = 0;
    uint8_t *data = nullptr;
    size_t step = 0;
private:
    int type_ = CV_8UC1;
    std::shared_ptr<uint8_t> buf_;
};
}  // namespace cv
using cv::Mat;
typedef unsigned char uchar;
typedef unsigned short ushort;
inline int cvRound(double v) { return (int)lrint(v); }        // OpenCV: round half to even (SSE2 cvtsd2si / lrint)

// ------------------------------------------------------------------------------------------ glog-shaped logging
namespace ygz_log {
struct Sink { bool on; std::ostringstream s; explicit Sink(bool o) : on(o) {} ~Sink() { if (on) std::cerr << s.str(); }
    template <typename T> Sink &operator<<(const T &v) { if (on) s << v; return *this; }
    Sink &operator<<(std::ostream &(*f)(std::ostream &)) { if (on) s << f; return *this; } };
extern int verbosity;      // 0 = errors/warnings only (default), 1 = INFO too
}
#define LOG(sev) ygz_log::Sink(ygz_log::sev##_enabled())
namespace ygz_log { inline bool INFO_enabled() { return verbosity > 0; } inline bool WARNING_enabled() { return true; } inline bool ERROR_enabled() { return true; } }

// ------------------------------------------------------------------------------------------ DBoW3 (the sliver in use)
// BowVector / FeatureVector with the interface src/ reads (thirdparty/DBoW3/src/{BowVector,FeatureVector}.h); the vocabulary
// tree itself lives in HBM (ygz_hip_vocab_load), Vocabulary is the handle the reference's call sites hold.
namespace DBoW3 {
typedef unsigned int WordId;
typedef double WordValue;
typedef unsigned int NodeId;
class BowVector : public std::map<WordId, WordValue> {};
class FeatureVector : public std::map<NodeId, std::vector<unsigned int>> {};
class Vocabulary {
public:
    bool loadFromBinaryFile(const std::string &filename);       // Vocabulary.cpp (loadFromBinaryFile), exactly nb_nodes records
    bool loadFromMemory(const void *blob, size_t bytes);
    bool empty() const { return n_words_ == 0; }
    // Vocabulary::transform(features, BowVector, FeatureVector, levelsup), Vocabulary.cpp:706-774
    void transform(const std::vector<cv::Mat> &features, BowVector &v, FeatureVector &fv, int levelsup) const;
    int k_ = 0, L_ = 0, n_nodes_ = 0, n_words_ = 0;
};
}  // namespace DBoW3
typedef DBoW3::Vocabulary ORBVocabulary;

// Local mapping patch sizes (Common.h:90-91)
const int WarpHalfPatchSize = 4;
const int WarpPatchSize = 8;

#endif  // YGZ_COMMON_INCLUDE_H_
