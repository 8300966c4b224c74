// libygz_host.so -- the ygz:: class surfaces (include/ygz/...) on top of the C ABI of libygz_hip.so.
// Host-side mirror of the reference interfaces for the hot path: same names, argument meaning and return
// conventions as src/Basic/Frame.cpp, src/Algorithm/{FeatureDetector,Matcher,Tracker,SparseImageAlign,CVUtils}.cpp.
// All dense work is delegated to the GPU library; there is no CPU fallback (a missing device makes Runtime throw).
#include "ygz/Basic.h"
#include "ygz/Algorithm.h"
#include "ygz/hip/Runtime.h"
#include "ygz_hip.h"
#include "../csrc/se3_dev.h"
#include "../csrc/ldlt6.h"
#include <fstream>
#include <stdexcept>
#include <cstdlib>
#include <chrono>
#include <limits>

int ygz_log::verbosity = 0;

// ------------------------------------------------------------------------------------------ Sophus shim
namespace Sophus {
static Se3 to_se3(const SE3 &T) { Se3 s; for (int i = 0; i < 4; ++i) s.q[i] = T.so3_.q_[i]; for (int i = 0; i < 3; ++i) s.t[i] = T.t_[i]; return s; }
static SE3 from_se3(const Se3 &s) { SE3 T; for (int i = 0; i < 4; ++i) T.so3_.q_[i] = s.q[i]; for (int i = 0; i < 3; ++i) T.t_[i] = s.t[i]; return T; }
SO3 SO3::exp(const Vector3d &w) { SO3 r; double th; so3_exp_d(w.d, r.q_, &th); return r; }
Vector3d SO3::log() const { Vector3d o; double th; so3_log_d(q_, o.d, &th); return o; }
SO3 SO3::inverse() const { SO3 r; r.q_[0] = -q_[0]; r.q_[1] = -q_[1]; r.q_[2] = -q_[2]; r.q_[3] = q_[3]; quat_normalize_d(r.q_); return r; }
SO3 SO3::operator*(const SO3 &o) const { SO3 r; quat_mul_d(q_, o.q_, r.q_); quat_normalize_d(r.q_); return r; }
Vector3d SO3::operator*(const Vector3d &p) const { Vector3d o; quat_rotate_d(q_, p.d, o.d); return o; }
Matrix3d SO3::matrix() const { Matrix3d R; quat_to_R_d(q_, R.m); return R; }
SE3 SE3::exp(const Vector6d &u) { Se3 s; se3_exp_d(u.d, &s); return from_se3(s); }
Vector6d SE3::log() const { Vector6d o; Se3 s = to_se3(*this); se3_log_d(&s, o.d); return o; }
SE3 SE3::inverse() const { Se3 a = to_se3(*this), b; se3_inv_d(&a, &b); return from_se3(b); }
SE3 SE3::operator*(const SE3 &o) const { Se3 a = to_se3(*this), b = to_se3(o), c; se3_mul_d(&a, &b, &c); return from_se3(c); }
Vector3d SE3::operator*(const Vector3d &p) const { Se3 a = to_se3(*this); Vector3d o; se3_act_d(&a, p.d, o.d); return o; }
std::ostream &operator<<(std::ostream &os, const SE3 &T)
{ os << "q(" << T.so3_.q_[0] << " " << T.so3_.q_[1] << " " << T.so3_.q_[2] << " " << T.so3_.q_[3] << ") t(" << T.t_[0] << " " << T.t_[1] << " " << T.t_[2] << ")"; return os; }
}

namespace ygz {

// ------------------------------------------------------------------------------------------ Config
static std::map<std::string, std::string> &cfg()
{
    static std::map<std::string, std::string> m = {          // config/default.yaml:8-66
        {"image.width", "640"}, {"image.height", "480"}, {"camera.fx", "520.9"}, {"camera.fy", "521.0"},
        {"camera.cx", "325.1"}, {"camera.cy", "249.7"}, {"frame.pyramid", "3"}, {"tracker.min_features", "50"},
        {"init.min_features", "100"}, {"init.min_disparity", "30"}, {"init.min_inliers", "40"}, {"feature.cell", "10"},
        {"feature.detection_threshold", "15.0"}, {"matcher.th_low", "65"}, {"matcher.th_high", "100"},
        {"matcher.init_low", "30"}, {"matcher.init_high", "100"}, {"matcher.knnRatio", "0.7"},
        {"vo.keyframe.min_rot", "0.1"}, {"vo.keyframe.min_trans", "0.1"}, {"vo.keyframe.min_features", "30"},
        {"LocalMapping.local_keyframes", "3"}, {"LocalMapping.local_mappoints", "500"} };
    return m;
}
bool Config::SetParameterFile(const std::string &filename)
{
    std::ifstream f(filename);
    if (!f) { LOG(ERROR) << "parameter file " << filename << " does not exist." << endl; return false; }
    std::string line;
    while (std::getline(f, line)) {
        const size_t h = line.find_first_of("#%");
        if (h != std::string::npos) line = line.substr(0, h);
        const size_t c = line.find(':');
        if (c == std::string::npos) continue;
        auto trim = [](std::string s) { const size_t a = s.find_first_not_of(" \t\r"), b = s.find_last_not_of(" \t\r"); return a == std::string::npos ? std::string() : s.substr(a, b - a + 1); };
        const std::string k = trim(line.substr(0, c)), v = trim(line.substr(c + 1));
        if (!k.empty() && !v.empty()) cfg()[k] = v;
    }
    return true;
}
void Config::Set(const std::string &key, const std::string &value) { cfg()[key] = value; }
std::string Config::Raw(const std::string &key) { auto it = cfg().find(key); return it == cfg().end() ? std::string("0") : it->second; }

// ------------------------------------------------------------------------------------------ Runtime (context + slots)
namespace hip {
struct Runtime::Impl {
    ygz_hip_ctx *ctx = nullptr;
    int max_frames = 0, levels = 0, cells = 0;
    std::vector<Frame *> owner;
    std::vector<unsigned long long> stamp;
    unsigned long long clock = 0;
    std::map<const uint8_t *, std::pair<Frame *, int>> level_of;     // host level data -> (frame, level)
};
Runtime &Runtime::Get() { static Runtime r; return r; }
Runtime::Runtime() : p_(new Impl) {}
Runtime::~Runtime() { if (p_->ctx) ygz_hip_destroy(p_->ctx); delete p_; }
bool check(int rc, const char *what)
{
    if (rc == YGZ_OK) return true;
    LOG(ERROR) << "ygz::hip: " << what << ": " << ygz_hip_error_string(rc) << endl;
    return false;
}
ygz_hip_ctx *Runtime::ctx()
{
    if (!p_->ctx) {
        ygz_hip_params prm;
        ygz_hip_default_params(&prm);
        prm.image_width = Config::Get<int>("image.width"); prm.image_height = Config::Get<int>("image.height");
        prm.pyramid_levels = Config::Get<int>("frame.pyramid");
        prm.cell_size = Config::Get<int>("feature.cell");
        prm.fast_threshold = (int)(short)Config::Get<double>("feature.detection_threshold");   // double -> short at the libfast call (FeatureDetector.cpp:368)
        prm.fx = Config::Get<float>("camera.fx"); prm.fy = Config::Get<float>("camera.fy");
        prm.cx = Config::Get<float>("camera.cx"); prm.cy = Config::Get<float>("camera.cy");
        const char *mf = getenv("YGZ_HIP_MAX_FRAMES");
        prm.max_frames = mf ? atoi(mf) : 64;
        const char *dev = getenv("YGZ_HIP_DEVICE");
        const int rc = ygz_hip_create(&p_->ctx, dev ? atoi(dev) : 0, &prm, nullptr);
        if (rc != YGZ_OK) throw std::runtime_error(std::string("ygz::hip::Runtime: no usable gfx950 device (ygz_hip_create: ") + ygz_hip_error_string(rc) + "); there is no CPU path");
        p_->max_frames = prm.max_frames; p_->levels = prm.pyramid_levels; p_->cells = ygz_hip_max_keypoints(p_->ctx);
        p_->owner.assign(prm.max_frames, nullptr); p_->stamp.assign(prm.max_frames, 0);
    }
    return p_->ctx;
}
int Runtime::cells() { ctx(); return p_->cells; }
void fdp_memo_forget(const Frame *f);
void Runtime::Release(Frame *f)
{
    fdp_memo_forget(f);
    for (auto it = p_->level_of.begin(); it != p_->level_of.end();) { if (it->second.first == f) it = p_->level_of.erase(it); else ++it; }
    if (f->_hip_slot >= 0 && f->_hip_slot < (int)p_->owner.size() && p_->owner[f->_hip_slot] == f) p_->owner[f->_hip_slot] = nullptr;
    f->_hip_slot = -1;
}
void Runtime::RegisterLevels(Frame *f)
{
    for (size_t L = 0; L < f->_pyramid.size(); ++L) if (f->_pyramid.fetched(L)) p_->level_of[f->_pyramid[L].data] = std::make_pair(f, (int)L);
}
bool Runtime::FindLevel(const uint8_t *data, Frame **f, int *level)
{
    auto it = p_->level_of.find(data);
    if (it == p_->level_of.end()) return false;
    *f = it->second.first; *level = it->second.second;
    return true;
}
// HBM slot of a frame; uploads (again) when the frame was evicted.  gray_only: level 0 is taken from _pyramid[0].
int Runtime::Resident(Frame *f)
{
    ygz_hip_ctx *c = ctx();
    if (f->_hip_slot >= 0 && p_->owner[f->_hip_slot] == f) { p_->stamp[f->_hip_slot] = ++p_->clock; return f->_hip_slot; }
    int slot = -1;
    for (int i = 0; i < p_->max_frames; ++i) if (!p_->owner[i]) { slot = i; break; }
    if (slot < 0) {                                                   // evict the least recently used frame
        slot = 0;
        for (int i = 1; i < p_->max_frames; ++i) if (p_->stamp[i] < p_->stamp[slot]) slot = i;
        p_->owner[slot]->_hip_slot = -1;
    }
    p_->owner[slot] = f; p_->stamp[slot] = ++p_->clock; f->_hip_slot = slot;
    bool up = true;
    if (!f->_pyramid.empty() && !f->_color.empty()) {                 // an initialised frame that lost its slot: the image goes up again (Frame.cpp:22-40 on the GPU)
        const int bgr = f->_color.channels() == 3;
        up = check(bgr ? ygz_hip_upload_bgr(c, slot, f->_color.data, (int)f->_color.step) : ygz_hip_upload_gray(c, slot, f->_color.data, (int)f->_color.step), "upload")
             && check(ygz_hip_build_pyramid(c, slot, 1, bgr), "build_pyramid");
    } else if (!f->_pyramid.empty() && f->_pyramid.fetched(0)) {      // _color was released by the caller: level 0 of the host mirror, if somebody fetched it
        up = check(ygz_hip_upload_gray(c, slot, f->_pyramid[0].data, (int)f->_pyramid[0].step), "upload_gray") && check(ygz_hip_build_pyramid(c, slot, 1, 0), "build_pyramid");
    } else if (!f->_pyramid.empty()) {
        LOG(ERROR) << "ygz::hip::Runtime: an evicted frame has neither _color nor a fetched level 0 to be uploaded again (raise YGZ_HIP_MAX_FRAMES)" << endl;
        up = false;
    }
    if (!up) { p_->owner[slot] = nullptr; f->_hip_slot = -1; return -1; }       // the ABI refuses slot -1 (YGZ_E_INVALID): the surface that asked reports failure
    return slot;
}
}  // namespace hip

// ------------------------------------------------------------------------------------------ Frame
PinholeCamera *Frame::_camera = nullptr;
ORBVocabulary *Frame::_vocab = nullptr;
Frame::~Frame() { if (!_features.empty()) CleanAllFeatures(); hip::Runtime::Get().Release(this); }

void hip::PyramidMirror::fetch(size_t L)
{   // the first reader of a level pays for its copy (and nobody else pays for levels nobody reads)
    if (L >= lv_.size()) throw std::out_of_range("Frame::_pyramid: level out of range");      // (vector::operator[] of the reference: undefined)
    if (have_[L] || !owner_) return;
    Runtime &rt = Runtime::Get();
    ygz_hip_ctx *c = rt.ctx();
    // the level counts as fetched only once its pixels are here: a re-upload of an evicted frame (Resident) must never take a level that is still
    // being fetched for the frame image, and a failed download must not leave uninitialised pixels marked valid
    const int slot = rt.Resident(owner_);
    int w = 0, h = 0;
    if (slot < 0 || !check(ygz_hip_level_size(c, (int)L, &w, &h), "level_size")) return;       // lv_[L] stays empty()
    cv::Mat img(h, w, CV_8UC1);
    if (!check(ygz_hip_download_level(c, slot, (int)L, img.data), "download_level")) return;
    lv_[L] = img;
    have_[L] = 1;
    rt.RegisterLevels(owner_);
}

void Frame::InitFrame()
{
    hip::Runtime &rt = hip::Runtime::Get();
    ygz_hip_ctx *c = rt.ctx();
    int w = 0, h = 0;
    ygz_hip_level_size(c, 0, &w, &h);
    rt.Release(this);
    _pyramid.clear();
    if (_color.empty() || _color.cols != w || _color.rows != h) {
        LOG(ERROR) << "Frame::InitFrame: _color (" << _color.cols << " x " << _color.rows << ") does not match image.width/height (" << w << " x " << h << "); no pyramid" << endl;
        return;
    }
    const int slot = rt.Resident(this);                 // no pyramid yet: slot only
    const int bgr = _color.channels() == 3;             // cv::cvtColor(CV_BGR2GRAY) + pyrDown on the GPU (Frame.cpp:27,38)
    if (!hip::check(bgr ? ygz_hip_upload_bgr(c, slot, _color.data, (int)_color.step) : ygz_hip_upload_gray(c, slot, _color.data, (int)_color.step), "upload")
        || !hip::check(ygz_hip_build_pyramid(c, slot, 1, bgr), "build_pyramid")) { rt.Release(this); return; }
    _pyramid.reset(this, (size_t)_option._pyramid_level);      // levels are fetched when somebody indexes them
}

void Frame::CreateImagePyramid()
{   // every level into the host mirror now (callers that walk frame->_pyramid[L] get them one by one anyway)
    if (_pyramid.size() != (size_t)_option._pyramid_level) _pyramid.reset(this, (size_t)_option._pyramid_level);
    for (size_t L = 0; L < _pyramid.size(); ++L) (void)_pyramid[L];
}

Mat Frame::GetAllDescriptors()
{   // Frame.cpp:178-188: one 32-byte row per feature, feature order
    const int n = (int)_features.size();
    Mat rows(n, 32, CV_8U);
    for (int r = 0; r < n; ++r) std::copy_n(_features[r]->_desc.data, 32, rows.ptr<uchar>(r));
    return rows;
}

void Frame::CleanAllFeatures()
{   // Frame.cpp:203-210: the frame owns its features
    for (Feature *f : _features) delete f;
    vector<Feature *>().swap(_features);
}

bool Frame::GetMeanAndMinDepth(double &mean_depth, double &min_depth)
{   // Frame.cpp:42-72: depth statistics of the features that have a good map point in front of the camera
    double sum = 0, lo = 9999;
    int n = 0;
    for (const Feature *f : _features) {
        if (!f->_mappoint || f->_mappoint->_bad) continue;
        const double z = (_TCW * f->_mappoint->_pos_world)[2];
        if (z < 0) continue;
        sum += z; lo = std::min(lo, z); ++n;
    }
    mean_depth = n ? sum / n : 0.0;
    min_depth = n ? lo : 0.0;
    return n > 0;
}


// ---- covisibility graph (src/Basic/Frame.cpp:73-176): which keyframes see the map points this one sees, and how many of them
vector<Frame *> Frame::GetBestCovisibilityKeyframes(const int &N)
{
    if ((int)_cov_keyframes.size() < N) return _cov_keyframes;
    return vector<Frame *>(_cov_keyframes.begin(), _cov_keyframes.begin() + N);
}
bool Frame::IsInFrustum(MapPoint *, float) { return true; }
void Frame::AddConnection(Frame *kf, const int &weight) { _connected_keyframe_weights[kf] = weight; }

namespace {
// (weight, keyframe) heaviest first; equal weights by descending pointer -- what sorting pair<int, Frame*> ascending and reading it backwards gives
void heaviest_first(vector<pair<int, Frame *>> &wk, vector<Frame *> &kfs, vector<int> &ws)
{
    std::sort(wk.begin(), wk.end(), [](const pair<int, Frame *> &a, const pair<int, Frame *> &b) { return b < a; });
    for (const auto &p : wk) { kfs.push_back(p.second); ws.push_back(p.first); }
}
}
void Frame::UpdateConnections()
{
    map<Frame *, int> shared;                         // other keyframe -> map points in common
    for (const Feature *fea : _features) {
        const MapPoint *mp = fea->_mappoint;
        if (!mp || mp->_bad) continue;
        for (const auto &ob : mp->_obs) if (ob.first != _keyframe_id) shared[Memory::GetKeyFrame(ob.first)]++;
    }
    if (shared.empty()) return;
    const int th = 15;                                // a connection needs 15 points in common; failing that, the single best keyframe is kept
    vector<pair<int, Frame *>> wk;
    wk.reserve(shared.size());
    pair<int, Frame *> best(0, nullptr);
    for (const auto &kv : shared) {
        if (kv.second > best.first) best = make_pair(kv.second, kv.first);
        if (kv.second >= th) wk.push_back(make_pair(kv.second, kv.first));
    }
    if (wk.empty()) {
        wk.push_back(best);
        best.second->AddConnection(this, best.first);
    }
    _connected_keyframe_weights = shared;
    _cov_keyframes.clear(); _cov_weights.clear();
    heaviest_first(wk, _cov_keyframes, _cov_weights);
    LOG(INFO) << "convisible keyframes: " << _cov_keyframes.size() << endl;
}
void Frame::UpdateBestCovisibles()
{   // appends (the reference does not clear first, Frame.cpp:162-176)
    vector<pair<int, Frame *>> wk;
    wk.reserve(_connected_keyframe_weights.size());
    for (const auto &kv : _connected_keyframe_weights) wk.push_back(make_pair(kv.second, kv.first));
    heaviest_first(wk, _cov_keyframes, _cov_weights);
}

// ------------------------------------------------------------------------------------------ Memory
static std::map<unsigned long, Frame *> g_keyframes;
static std::map<unsigned long, MapPoint *> g_points;
static unsigned long g_kf_id = 0, g_pt_id = 0;
Frame *Memory::RegisterKeyFrame(Frame *frame, bool overwrite)
{
    if (!overwrite || g_keyframes.find(frame->_keyframe_id) == g_keyframes.end()) frame->_keyframe_id = g_kf_id++;
    frame->_is_keyframe = true;
    g_keyframes[frame->_keyframe_id] = frame;
    return frame;
}
MapPoint *Memory::RegisterMapPoint(MapPoint *mp) { mp->_id = g_pt_id++; g_points[mp->_id] = mp; return mp; }
MapPoint *Memory::CreateMapPoint() { return RegisterMapPoint(new MapPoint); }      // Memory.cpp:45-52
Frame *Memory::GetKeyFrame(const unsigned long &id) { auto it = g_keyframes.find(id); return it == g_keyframes.end() ? nullptr : it->second; }
MapPoint *Memory::GetMapPoint(const unsigned long &id) { auto it = g_points.find(id); return it == g_points.end() ? nullptr : it->second; }
void Memory::Clean() { g_keyframes.clear(); g_points.clear(); g_kf_id = g_pt_id = 0; }

// ------------------------------------------------------------------------------------------ FeatureDetector
FeatureDetector::FeatureDetector()
{
    _option._grid_rows = (int)ceil(double(_option._image_height) / _option._cell_size);
    _option._grid_cols = (int)ceil(double(_option._image_width) / _option._cell_size);
    _old_features = vector<Feature *>(_option._grid_cols * _option._grid_rows, nullptr);
}

void FeatureDetector::LoadParams()
{
    _option._image_width = Config::Get<int>("image.width");
    _option._image_height = Config::Get<int>("image.height");
    _option._cell_size = Config::Get<int>("feature.cell");
    _option._grid_rows = (int)ceil(double(_option._image_height) / _option._cell_size);
    _option._grid_cols = (int)ceil(double(_option._image_width) / _option._cell_size);
    _option._detection_threshold = Config::Get<double>("feature.detection_threshold");
    _old_features = vector<Feature *>(_option._grid_cols * _option._grid_rows, nullptr);
}

void FeatureDetector::SetExistingFeatures(Frame *frame)
{
    for (Feature *&fea : _old_features) fea = nullptr;
    for (Feature *fea : frame->_features) {
        int gx = static_cast<int>(fea->_pixel[0] / _option._cell_size);
        int gy = static_cast<int>(fea->_pixel[1] / _option._cell_size);
        size_t k = gy * _option._grid_cols + gx;
        if (k >= _old_features.size()) continue;
        _old_features[k] = fea;
    }
}

// YGZ_HOST_TRACE=1: host clock per phase of a class-surface call (what the caller's thread does around the launches), printed when the process ends
namespace {
struct PhaseTrace {
    const char *name; const char *label[4];
    bool on = [] { const char *e = getenv("YGZ_HOST_TRACE"); return e && atoi(e) != 0; }();
    double ms[4] = { 0, 0, 0, 0 }; long calls = 0;
    std::chrono::steady_clock::time_point t;
    PhaseTrace(const char *n, std::initializer_list<const char *> l) : name(n) { int i = 0; for (const char *x : l) if (i < 4) label[i++] = x; for (; i < 4; ++i) label[i] = nullptr; }
    void start() { if (on) t = std::chrono::steady_clock::now(); }
    void lap(int k) { if (on) { const auto n = std::chrono::steady_clock::now(); ms[k] += std::chrono::duration<double, std::milli>(n - t).count(); t = n; } }
    void done() { ++calls; }
    ~PhaseTrace()
    {
        if (!on || !calls) return;
        fprintf(stderr, "%s x %ld:", name, calls);
        for (int k = 0; k < 4 && label[k]; ++k) fprintf(stderr, " %s %.3f ", label[k], ms[k] / calls);
        fprintf(stderr, "ms per call\n");
    }
};
PhaseTrace g_detect_trace{ "FeatureDetector::Detect", { "grid", "ygz_hip_detect", "ygz_hip_get_keypoints", "new Feature" } };
PhaseTrace g_align_trace{ "SparseImgAlign::run", { "gather", "ygz_hip_sparse_align", "write-back", nullptr } };
PhaseTrace g_pmp_trace{ "Matcher::ProjectMapPoints", { "gather", "ygz_hip_track_local_map", "new Feature", nullptr } };
PhaseTrace g_po_trace{ "ba::OptimizeCurrentPoseOnly", { "gather", "ygz_hip_optimize_pose_only", "write-back", nullptr } };
}
void FeatureDetector::Detect(Frame *frame, bool overwrite_existing_features)
{
    g_detect_trace.start();
    hip::Runtime &rt = hip::Runtime::Get();
    ygz_hip_ctx *c = rt.ctx();
    const int cells = rt.cells();
    if ((int)_old_features.size() != cells) { LOG(ERROR) << "FeatureDetector::Detect: grid does not match the context (call LoadParams())" << endl; return; }
    const int slot = rt.Resident(frame);
    std::vector<uint8_t> occ;
    if (overwrite_existing_features) {
        _old_features = vector<Feature *>(cells, nullptr);
        frame->CleanAllFeatures();
    } else {
        SetExistingFeatures(frame);
        occ.resize(cells);
        for (int k = 0; k < cells; ++k) occ[k] = _old_features[k] ? 1 : 0;
    }
    g_detect_trace.lap(0);
    if (!hip::check(ygz_hip_detect(c, slot, 1, occ.empty() ? nullptr : occ.data()), "detect")) return;
    g_detect_trace.lap(1);
    // (result buffers of a whole grid, kept between calls: 180 KB that would otherwise be allocated and zeroed per frame)
    static thread_local std::vector<double> px; static thread_local std::vector<int32_t> lvl; static thread_local std::vector<float> sc, ang;
    static thread_local std::vector<uint8_t> desc;
    if ((int)lvl.size() < cells) { px.resize(2 * (size_t)cells); lvl.resize(cells); sc.resize(cells); ang.resize(cells); desc.resize(32 * (size_t)cells); }
    ygz_kpt_soa soa = { px.data(), lvl.data(), sc.data(), ang.data(), desc.data() };
    int n = 0;
    if (!hip::check(ygz_hip_get_keypoints(c, slot, &soa, cells, &n), "get_keypoints")) return;
    g_detect_trace.lap(2);
    LOG(INFO) << "old features: " << frame->_features.size() << endl;
    for (int i = 0; i < n; ++i) {
        Feature *fea = new Feature(Vector2d(px[2 * i], px[2 * i + 1]), lvl[i], sc[i]);
        fea->_frame = frame;
        fea->_angle = ang[i];
        memcpy(fea->_desc.data, &desc[32 * (size_t)i], 32);
        frame->_features.push_back(fea);
    }
    LOG(INFO) << "add total " << n << " new features." << endl;
    g_detect_trace.lap(3); g_detect_trace.done();
}

static void describe_features(Frame *frame, const vector<Feature *> &feas, bool given_angle)
{
    hip::Runtime &rt = hip::Runtime::Get();
    ygz_hip_ctx *c = rt.ctx();
    const int slot = rt.Resident(frame);
    const int cells = rt.cells();
    for (size_t base = 0; base < feas.size(); base += cells) {
        const int n = (int)std::min((size_t)cells, feas.size() - base);
        std::vector<double> px(2 * (size_t)n); std::vector<int32_t> lvl(n); std::vector<float> ang(n);
        for (int i = 0; i < n; ++i) { const Feature *f = feas[base + i]; px[2 * i] = f->_pixel[0]; px[2 * i + 1] = f->_pixel[1]; lvl[i] = f->_level; ang[i] = (float)f->_angle; }
        if (!hip::check(given_angle ? ygz_hip_describe_given_angle(c, slot, px.data(), lvl.data(), ang.data(), n) : ygz_hip_describe(c, slot, px.data(), lvl.data(), n), "describe")) return;
        std::vector<float> oang(n); std::vector<uint8_t> desc(32 * (size_t)n);
        ygz_kpt_soa soa = { nullptr, nullptr, nullptr, oang.data(), desc.data() };
        int m = 0;
        if (!hip::check(ygz_hip_get_keypoints(c, slot, &soa, n, &m), "get_keypoints")) return;
        for (int i = 0; i < n; ++i) { Feature *f = feas[base + i]; if (!given_angle) f->_angle = oang[i]; memcpy(f->_desc.data, &desc[32 * (size_t)i], 32); }
    }
}

void FeatureDetector::ComputeAngleAndDescriptor(Frame *frame) { describe_features(frame, frame->_features, false); }
void FeatureDetector::ComputeDescriptor(Feature *fea) { describe_features(fea->_frame, vector<Feature *>(1, fea), true); }

// ------------------------------------------------------------------------------------------ Tracker
Tracker::Tracker() { _option._min_feature_tracking = Config::Get<int>("tracker.min_features"); }

void Tracker::SetReference(Frame *ref)
{   // Tracker.cpp:13-32
    const size_t n = ref->_features.size();
    if ((int)n < _option._min_feature_tracking) {
        LOG(WARNING) << "Tracker::SetReference: only " << n << " features in the reference frame (need " << _option._min_feature_tracking << ")" << endl;
        _status = TRACK_NOT_READY;
        return;
    }
    _ref = _curr = ref;
    _status = TRACK_GOOD;
    _tracks.feature.reserve(_tracks.size() + n);
    for (Feature *fea : ref->_features) {                    // appended, as the reference does when called twice
        const float x = (float)fea->_pixel[0], y = (float)fea->_pixel[1];
        _tracks.feature.push_back(fea);
        _tracks.ref_px.push_back(x); _tracks.ref_px.push_back(y);
        _tracks.cur_px.push_back(x); _tracks.cur_px.push_back(y);
    }
}

void Tracker::Track(Frame *curr)
{   // Tracker.cpp:34-53
    if (_status != TRACK_GOOD) {
        LOG(WARNING) << (_status == TRACK_NOT_READY ? "Tracker::Track: no reference set" : "Tracker::Track: tracking was lost, set a new reference") << endl;
        return;
    }
    _curr = curr;
    TrackKLT();
    if ((int)_tracks.size() < _option._min_feature_tracking) {
        _status = TRACK_LOST;
        LOG(WARNING) << "Tracker::Track: " << _tracks.size() << " tracks left, lost" << endl;
    }
}

void Tracker::GetTrackedPixel(vector<Feature *> &feature1, vector<Vector2d> &pixels2) const
{
    feature1.insert(feature1.end(), _tracks.feature.begin(), _tracks.feature.end());
    for (size_t i = 0; i < _tracks.size(); ++i) pixels2.push_back(Vector2d(_tracks.cur_px[2 * i], _tracks.cur_px[2 * i + 1]));
}

void Tracker::TrackKLT()
{   // cv::calcOpticalFlowPyrLK(ref, cur, pt_ref, pt_curr, ..., Size(21,21), 4, COUNT+EPS(30,1e-3), USE_INITIAL_FLOW) and the survivor rule
    // status && InFrame(pt, 20) (Tracker.cpp:92-112), both on the GPU; the host only compacts the rows that were kept
    hip::Runtime &rt = hip::Runtime::Get();
    ygz_hip_ctx *c = rt.ctx();
    const int cells = rt.cells(), n = (int)_tracks.size();
    ygz_klt_params prm;
    ygz_hip_default_klt_params(&prm);
    prm.win = (int)_option.klt_win_size; prm.max_iter = _option.klt_max_iter; prm.eps = _option.klt_eps;
    const int rs = rt.Resident(_ref), cs = rt.Resident(_curr);
    vector<uint8_t> status(n), keep(n); vector<float> err(n);
    for (int base = 0; base < n; base += cells) {
        const int m = std::min(cells, n - base);
        int kept = 0;
        if (!hip::check(ygz_hip_klt_track_filtered(c, rs, cs, &_tracks.ref_px[2 * base], &_tracks.cur_px[2 * base], m, &prm, 20, &status[base], &err[base],
                                                   &keep[base], &kept), "klt_track_filtered")) return;      // tracks as they were
    }
    size_t w = 0;
    for (int i = 0; i < n; ++i) {
        if (!keep[i]) continue;
        _tracks.feature[w] = _tracks.feature[i];
        _tracks.ref_px[2 * w] = _tracks.ref_px[2 * i]; _tracks.ref_px[2 * w + 1] = _tracks.ref_px[2 * i + 1];
        _tracks.cur_px[2 * w] = _tracks.cur_px[2 * i]; _tracks.cur_px[2 * w + 1] = _tracks.cur_px[2 * i + 1];
        ++w;
    }
    _tracks.feature.resize(w); _tracks.ref_px.resize(2 * w); _tracks.cur_px.resize(2 * w);
}

float Tracker::MeanDisparity() const
{   // Tracker.cpp:115-127: a float accumulator over |feature pixel (double) - current position| in track order
    float acc = 0;
    for (size_t i = 0; i < _tracks.size(); ++i) {
        const Vector2d &p = _tracks.feature[i]->_pixel;
        const double dx = p[0] - (double)_tracks.cur_px[2 * i], dy = p[1] - (double)_tracks.cur_px[2 * i + 1];
        acc += std::sqrt(dx * dx + dy * dy);              // float += double: the sum is formed in double and rounded, as in the reference
    }
    return acc / _tracks.size();
}

// ------------------------------------------------------------------------------------------ SparseImgAlign
SparseImgAlign::SparseImgAlign(int max_level, int min_level, int n_iter, Method method, bool, bool)
    : max_level_(max_level), min_level_(min_level), n_iter_(n_iter), method_(method) {}

// method_ = LevenbergMarquardt: NLLSSolver::optimizeLevenbergMarquardt (include/ygz/Algorithm/NLSSolver_impl.hpp:91-212) per level, as
// SparseImgAlign::run drives it (SparseImageAlign.cpp:21-50).  The bookkeeping of the solver runs here; every computeResiduals(model, ...) is one
// launch (ygz_hip_sparse_align_residuals: precomputeReferencePatches + the residual pass of that level, the float chi2 sum exact).  Reproduced as
// written: mu_ = 0.1 before every level (:41), nu_ / stop_ / n_meas_ carried from level to level -- the first computeResiduals of a level (:101)
// counts its measurements ON TOP of the last evaluation of the level before --, H_ += diag(H_) mu_, Eigen's ldlt, T exp(-x), rho_ = chi2_ -
// new_chi2, five failed trials stop.  (The reference evaluates a trial's new model without linearising; the launch linearises anyway, the sums are
// not read.)  oracle/sparse_align.c: yo_sparse_align_lm is the same loop on the CPU.
size_t SparseImgAlign::run_lm(Frame *ref_frame, Frame *cur_frame)
{
    hip::Runtime &rt = hip::Runtime::Get();
    ygz_hip_ctx *c = rt.ctx();
    const int n = (int)ref_frame->_features.size();
    vector<double> px(2 * (size_t)n), depth(n); vector<uint8_t> has(n);
    for (int i = 0; i < n; ++i) {
        const Feature *f = ref_frame->_features[i];
        px[2 * i] = f->_pixel[0]; px[2 * i + 1] = f->_pixel[1]; depth[i] = f->_depth; has[i] = f->_mappoint != nullptr;
    }
    const int rs = rt.Resident(ref_frame), cs = rt.Resident(cur_frame);
    size_t n_meas_ = 0;
    bool failed = false;
    // computeResiduals(model, linearize_system, false): chi2 / n_meas_ as float / size_t -> float (SparseImageAlign.cpp:222); n_meas_ accumulates
    auto compute_residuals = [&](const SE3 &model, int level, double *H, double *Jres) -> double {
        double t7[7], csum = 0; int cnt = 0;
        model.to7(t7);
        if (!hip::check(ygz_hip_sparse_align_residuals(c, rs, cs, t7, px.data(), depth.data(), has.data(), n, level, &csum, &cnt, H, Jres), "sparse_align_residuals")) failed = true;
        n_meas_ += (size_t)cnt;
        return (double)((float)csum / (float)n_meas_);
    };
    // reset() (NLSSolver_impl.hpp:283-293) with the defaults of NLSSolver.h:75-95
    double chi2_ = 1e10, mu_ = (double)0.01f, nu_ = 2.0, rho_ = 0;
    bool stop_ = false;
    const double eps_ = 0.000001;                              // SparseImageAlign.cpp:18
    const int n_trials_max_ = 5;
    trials_ = 0;
    for (int &it : iters_) it = 0;
    SE3 T = cur_frame->_TCW * ref_frame->_TCW.inverse();       // T_cur_from_ref (:37)
    double H[36], Jres[6], x[6];
    for (int level = max_level_; level >= min_level_ && !failed; --level) {
        mu_ = 0.1;                                             // :41
        chi2_ = compute_residuals(T, level, H, Jres);          // :101 (n_meas_ not reset: see above)
        if (mu_ < 0) { double mx = 0; for (int j = 0; j < 6; ++j) mx = std::max(mx, fabs(H[7 * j])); mu_ = 1e-4 * mx; }      // :113-120, never true here
        int it = 0;
        for (; it < n_iter_ && !failed; ++it) {
            rho_ = 0;
            int n_trials_ = 0;
            do {
                SE3 new_model = T;
                double new_chi2 = -1;
                n_meas_ = 0;
                compute_residuals(T, level, H, Jres);                          // H_, Jres_ zeroed, linearised at the model (:133-139)
                for (int j = 0; j < 6; ++j) H[7 * j] += H[7 * j] * mu_;        // :142
                if (ldlt6_solve_d(H, Jres, x)) {                               // SparseImgAlign::solve
                    Vector6d mx; for (int k = 0; k < 6; ++k) mx[k] = -x[k];
                    new_model = T * SE3::exp(mx);                              // update (:233-238)
                    n_meas_ = 0;
                    new_chi2 = compute_residuals(new_model, level, nullptr, nullptr);
                    rho_ = chi2_ - new_chi2;
                } else rho_ = -1;
                ++trials_;
                if (rho_ > 0) {
                    T = new_model;
                    chi2_ = new_chi2;
                    double nm = -1; for (int k = 0; k < 6; ++k) nm = std::max(nm, fabs(x[k]));
                    stop_ = nm <= eps_;
                    mu_ *= std::max(1. / 3., std::min(1. - pow(2 * rho_ - 1, 3), 2. / 3.));
                    nu_ = 2.;
                } else {
                    mu_ *= nu_;
                    nu_ *= 2.;
                    ++n_trials_;
                    if (n_trials_ >= n_trials_max_) stop_ = true;
                }
            } while (!(rho_ > 0 || stop_) && !failed);
            if (stop_) break;
        }
        if (level >= 0 && level < 8) iters_[level] = it;
    }
    if (failed) return 0;                                      // an ABI call failed (logged): the pose as it was
    cur_frame->_TCW = T * ref_frame->_TCW;                     // :48
    return n_meas_ / 16;
}

size_t SparseImgAlign::run(Frame *ref_frame, Frame *cur_frame)
{
    if (ref_frame->_features.empty()) return 0;
    if (method_ == LevenbergMarquardt) {
        if ((int)ref_frame->_features.size() > hip::Runtime::Get().cells()) { LOG(ERROR) << "SparseImgAlign::run: more features than grid cells" << endl; return 0; }
        return run_lm(ref_frame, cur_frame);
    }
    g_align_trace.start();
    hip::Runtime &rt = hip::Runtime::Get();
    ygz_hip_ctx *c = rt.ctx();
    const int n = (int)ref_frame->_features.size();
    if (n > rt.cells()) { LOG(ERROR) << "SparseImgAlign::run: " << n << " features, more than grid cells (" << rt.cells() << ")" << endl; return 0; }
    vector<double> px(2 * (size_t)n), depth(n); vector<uint8_t> has(n);
    for (int i = 0; i < n; ++i) {
        const Feature *f = ref_frame->_features[i];
        px[2 * i] = f->_pixel[0]; px[2 * i + 1] = f->_pixel[1]; depth[i] = f->_depth; has[i] = f->_mappoint != nullptr;
    }
    double Tr[7], Tc[7];
    ref_frame->_TCW.to7(Tr); cur_frame->_TCW.to7(Tc);
    int n_meas = 0;
    const int rs = rt.Resident(ref_frame), cs = rt.Resident(cur_frame);
    g_align_trace.lap(0);
    if (!hip::check(ygz_hip_sparse_align(c, rs, Tr, cs, Tc, px.data(), depth.data(), has.data(), n, max_level_, min_level_, n_iter_, &n_meas, iters_), "sparse_align")) return 0;
    g_align_trace.lap(1);
    cur_frame->_TCW = SE3::from7(Tc);
    g_align_trace.lap(2); g_align_trace.done();
    return (size_t)n_meas;
}

// ------------------------------------------------------------------------------------------ DBoW3 surface
}  // namespace ygz
namespace DBoW3 {
bool Vocabulary::loadFromMemory(const void *blob, size_t bytes)
{
    ygz::hip::Runtime &rt = ygz::hip::Runtime::Get();
    if (ygz_hip_vocab_load(rt.ctx(), blob, bytes) != YGZ_OK) return false;
    return ygz_hip_vocab_info(rt.ctx(), &k_, &L_, &n_nodes_, &n_words_) == YGZ_OK;
}
bool Vocabulary::loadFromBinaryFile(const std::string &filename)
{
    FILE *f = fopen(filename.c_str(), "rb");
    if (!f) return false;
    std::vector<uint8_t> buf;
    uint8_t tmp[65536]; size_t n;
    while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    fclose(f);
    return loadFromMemory(buf.data(), buf.size());
}
void Vocabulary::transform(const std::vector<cv::Mat> &features, BowVector &v, FeatureVector &fv, int levelsup) const
{   // Vocabulary.cpp:706-774 for TF_IDF / TF weighting with L1 scoring (what an ORB vocabulary file carries)
    v.clear(); fv.clear();
    if (empty() || features.empty()) return;
    const int n = (int)features.size();
    std::vector<uint8_t> desc((size_t)n * 32);
    for (int i = 0; i < n; ++i) memcpy(&desc[32 * (size_t)i], features[i].data, 32);
    std::vector<int32_t> word(n), node(n); std::vector<double> weight(n);
    if (!ygz::hip::check(ygz_hip_bow_transform(ygz::hip::Runtime::Get().ctx(), desc.data(), n, levelsup, word.data(), weight.data(), node.data()), "bow_transform")) return;
    for (int i = 0; i < n; ++i) {
        if (!(weight[i] > 0) || word[i] < 0) continue;          // stopped word
        v[(WordId)word[i]] += weight[i];                         // BowVector::addWeight
        fv[(NodeId)node[i]].push_back((unsigned int)i);          // FeatureVector::addFeature
    }
    double norm = 0.0;                                           // BowVector::normalize(L1)
    for (auto &kv : v) norm += fabs(kv.second);
    if (norm > 0.0) for (auto &kv : v) kv.second /= norm;
}
}  // namespace DBoW3
namespace ygz {
void Frame::ComputeBoW()
{   // src/Basic/Frame.cpp:190-201
    if (_vocab != nullptr && _bow_vec.empty()) {
        vector<Mat> alldesp;
        for (Feature *fea : _features) alldesp.push_back(fea->_desc);
        _vocab->transform(alldesp, _bow_vec, _feature_vec, 4);
    }
}

// ------------------------------------------------------------------------------------------ Matcher
Matcher::Matcher()
{
    _options.th_low = Config::Get<int>("matcher.th_low");
    _options.th_high = Config::Get<int>("matcher.th_high");
    _options.init_low = Config::Get<int>("matcher.init_low");
    _options.init_high = Config::Get<int>("matcher.init_high");
    _options.knnRatio = (float)Config::Get<int>("matcher.knnRatio");      // read through Get<int> in the reference (Matcher.cpp:17)
    _align = new SparseImgAlign(2, 0, 30, SparseImgAlign::GaussNewton, false, false);
}
Matcher::~Matcher() { delete _align; }

int Matcher::DescriptorDistance(const Mat &a, const Mat &b)
{   // one pair of 256-bit rows: host popcount (the batched form is BruteForceMatch on the GPU)
    const uint32_t *pa = a.ptr<uint32_t>(), *pb = b.ptr<uint32_t>();
    int dist = 0;
    for (int i = 0; i < 8; i++) dist += __builtin_popcount(pa[i] ^ pb[i]);
    return dist;
}

int Matcher::CheckFrameDescriptors(Frame *frame1, Frame *frame2, list<pair<int, int>> &matches)
{   // Matcher.cpp:45-84 through ygz_hip_check_descriptor_pairs: distances, clamped best and keep flags come back from the GPU,
    // the list is filtered with them
    const size_t n = matches.size();
    if (n == 0) return 0;
    vector<uint8_t> d1(32 * n), d2(32 * n), keep(n);
    size_t r = 0;
    for (const auto &m : matches) {
        memcpy(&d1[32 * r], frame1->_features[m.first]->_desc.data, 32);
        memcpy(&d2[32 * r], frame2->_features[m.second]->_desc.data, 32);
        ++r;
    }
    int n_good = 0, best = 0;
    if (!hip::check(ygz_hip_check_descriptor_pairs(hip::Runtime::Get().ctx(), d1.data(), d2.data(), (int)n, _options.init_low, _options.init_high,
                                                   _options.initMatchRatio, nullptr, keep.data(), &n_good, &best), "check_descriptor_pairs")) return 0;
    LOG(INFO) << "best dist = " << best << ", kept " << n_good << " of " << n << endl;
    r = 0;
    matches.remove_if([&](const pair<int, int> &) { return keep[r++] == 0; });
    return n_good;
}

namespace {
// descriptors, FeatureVector membership (node id or -1) and pixels of a frame as flat arrays
void bow_arrays(Frame *kf, std::vector<uint8_t> &desc, std::vector<int32_t> &node, std::vector<double> &px)
{
    const size_t n = kf->_features.size();
    desc.resize(n * 32); node.assign(n, -1); px.resize(n * 2);
    for (size_t i = 0; i < n; ++i) {
        memcpy(&desc[32 * i], kf->_features[i]->_desc.data, 32);
        px[2 * i] = kf->_features[i]->_pixel[0]; px[2 * i + 1] = kf->_features[i]->_pixel[1];
    }
    for (auto &kv : kf->_feature_vec) for (unsigned int idx : kv.second) if (idx < n) node[idx] = (int32_t)kv.first;
}
}  // namespace

int Matcher::SearchByBoW(Frame *kf1, Frame *kf2, map<int, int> &matches)
{   // Matcher.cpp:196-292: within equal vocabulary nodes, best < th_low and best < knnRatio * second best
    std::vector<uint8_t> d1, d2; std::vector<int32_t> n1, n2; std::vector<double> p1, p2;
    bow_arrays(kf1, d1, n1, p1); bow_arrays(kf2, d2, n2, p2);
    std::vector<int32_t> m(std::max<size_t>(n1.size(), 1), -1);
    int cnt = 0;
    if (!hip::check(ygz_hip_search_by_bow(hip::Runtime::Get().ctx(), 0, d1.data(), n1.data(), nullptr, (int)n1.size(), d2.data(), n2.data(), nullptr,
                                          (int)n2.size(), nullptr, _options.th_low, _options.knnRatio, 0.0, m.data(), &cnt), "search_by_bow")) return 0;
    for (size_t i = 0; i < n1.size(); ++i) if (m[i] >= 0) matches[(int)i] = m[i];
    if (_options.checkOrientation && !n1.empty()) {
        // Matcher.cpp:247-256, 271-289: the rotation histogram only lowers the returned count (the matches outside the three fullest bins stay in
        // the map: the reference's TODO at :284)
        std::vector<double> a1(n1.size()), a2(std::max<size_t>(n2.size(), 1));
        for (size_t i = 0; i < n1.size(); ++i) a1[i] = kf1->_features[i]->_angle;
        for (size_t i = 0; i < n2.size(); ++i) a2[i] = kf2->_features[i]->_angle;
        if (!hip::check(ygz_hip_bow_orientation(hip::Runtime::Get().ctx(), a1.data(), (int)n1.size(), a2.data(), (int)n2.size(), m.data(), &cnt, nullptr, nullptr),
                        "bow_orientation")) return 0;
    }
    return cnt;
}

// (checkOrientation: the reference fills a rotation histogram here too, Matcher.cpp:157-165, and never reads it -- "TODO" at :180-182 --, so the
// option changes nothing in this function)
int Matcher::SearchForTriangulation(Frame *kf1, Frame *kf2, const Matrix3d &E12, vector<pair<int, int>> &matched_points, const bool &)
{   // Matcher.cpp:86-193 (+ CheckDistEpipolarLine :338-354)
    assert(!kf1->_feature_vec.empty() && !kf2->_feature_vec.empty());
    std::vector<uint8_t> d1, d2; std::vector<int32_t> n1, n2; std::vector<double> p1, p2;
    bow_arrays(kf1, d1, n1, p1); bow_arrays(kf2, d2, n2, p2);
    std::vector<int32_t> m(std::max<size_t>(n1.size(), 1), -1);
    int cnt = 0;
    matched_points.clear();
    if (!hip::check(ygz_hip_search_by_bow(hip::Runtime::Get().ctx(), 1, d1.data(), n1.data(), p1.data(), (int)n1.size(), d2.data(), n2.data(), p2.data(),
                                          (int)n2.size(), E12.m, _options.th_low, _options.knnRatio, _options._epipolar_dsqr, m.data(), &cnt),
                    "search_for_triangulation")) return 0;
    matched_points.reserve(cnt);
    for (size_t i = 0; i < n1.size(); ++i) if (m[i] >= 0) matched_points.push_back(make_pair((int)i, (int)m[i]));
    _tri_kf1 = kf1; _tri_kf2 = kf2; _tri_pairs = matched_points;      // (what FindDirectProjection's Feature overload may be asked about next)
    return cnt;
}

int Matcher::BruteForceMatch(Frame *frame1, Frame *frame2, vector<DMatch> &matches, bool cross_check)
{
    hip::Runtime &rt = hip::Runtime::Get();
    Mat d1 = frame1->GetAllDescriptors(), d2 = frame2->GetAllDescriptors();
    // (the ABI takes up to grid cells x resident frames rows per set and reports YGZ_E_CAPACITY beyond: logged, 0 matches)
    vector<int32_t> idx(d1.rows), dist(d1.rows);
    matches.clear();
    if (!hip::check(ygz_hip_hamming_match(rt.ctx(), d1.data, d1.rows, d2.data, d2.rows, cross_check ? 1 : 0, idx.data(), dist.data(), nullptr), "hamming_match")) return 0;
    for (int i = 0; i < d1.rows; ++i) if (idx[i] >= 0) { DMatch m; m.queryIdx = i; m.trainIdx = idx[i]; m.distance = (float)dist[i]; matches.push_back(m); }
    return (int)matches.size();
}


// ------------------------------------------------------------------------------------------ FindDirectProjection behind its per-candidate callers
// LocalMapping::ProjectMapPoints calls Matcher::FindDirectProjection once per candidate (src/Module/LocalMapping.cpp:88-118, 1000-3800 calls per
// frame) and CreateNewMapPoints once per matched feature pair (:447).  One call = upload, launch, download, synchronise: ~40 us, i.e. tens of
// milliseconds per frame.  FindDirectProjection is a pure function of (both images, both poses, the reference observation, the map point's
// position / the feature's depth, the prediction), so the first call of a current frame that misses runs ONE launch over every candidate the caller
// can be expected to ask about (below) and keeps the answers; the calls that follow are a table look-up.  An answer is handed out only when every
// input of the call equals the memoised one BIT FOR BIT -- anything else takes the n = 1 launch --, so results are identical to n = 1 calls
// (tests/test_gpu_surface.py: the per-candidate loop with YGZ_FDP_MEMO=0 against the default).
//
// What is speculated, MapPoint overload: for the keyframe `ref` of the call and every keyframe the previous current frame asked about, every
// feature of the keyframe that observes a good map point (mp->_obs[ref->_keyframe_id]), with the prediction FindCandidates makes
// (Camera2Pixel(World2Camera(mp->_pos_world, curr->_TCW)), LocalMapping.cpp:58-59, evaluated by the launch itself and compared with the caller's).
// A keyframe that was not covered gets its own launch on its first miss.  Feature overload: the matches the same Matcher object's last
// SearchForTriangulation(ref, curr, ...) returned, with the depth and prediction CreateNewMapPoints forms from them (LocalMapping.cpp:416-446).
namespace {
inline bool same7(const SE3 &T, const double t7[7])
{ return memcmp(T.so3_.q_, t7, 32) == 0 && memcmp(T.t_, t7 + 4, 24) == 0; }
inline bool env_on(const char *name, bool dflt) { const char *e = getenv(name); return e ? atoi(e) != 0 : dflt; }

struct FdpMemo {
    struct Ref { Frame *f; double T[7]; };
    struct Entry {                                     // inputs (compared on every look-up) and outputs of one candidate
        const Frame *ref; const void *key;             // key: the MapPoint (MapPoint overload) or the reference Feature (Feature overload)
        double a[3];                                   // mp->_pos_world | (fea->_depth, 0, 0)
        double px_ref[2]; int32_t level;
        double px_in[2], px_out[2]; int32_t sl; uint8_t ok;
    };
    Frame *curr = nullptr;
    double T_cur[7];
    std::vector<Ref> refs;                             // keyframes whose map-point candidates are in the table (MapPoint overload)
    std::vector<Ref> feat_refs;                        // keyframes whose triangulation candidates are in the table (Feature overload)
    std::vector<Entry> entries;
    std::vector<int32_t> table;                        // open addressing over (ref, key), -1 = free
    std::vector<Frame *> asked, asked_prev;            // keyframes the calls of this / the previous current frame named
    hip::FdpMemoStats st;
    bool enabled = env_on("YGZ_FDP_MEMO", true);
    bool prelaunch = env_on("YGZ_FDP_PRELAUNCH", true);   // queue the frame's speculative launch at the end of Matcher::SparseImageAlignment (below)
    bool bypass = false;                               // calls take the n = 1 launch and leave the memo alone (A/B inside one loop)
    // a speculative launch that has been queued (ygz_hip_find_direct_projection_mp_begin) and not collected yet: what was asked
    struct Gathered {                                  // the candidates of some keyframes: everything a launch needs except the current frame's pose
        std::vector<Frame *> kfs; std::vector<int32_t> kf_slot; std::vector<double> kf_T;
        std::vector<int32_t> ck, cl; std::vector<double> pos, cpx; std::vector<const MapPoint *> cmp;
        std::vector<Entry> ent; std::vector<int32_t> tab;   // optional: the table entries of these candidates with their input fields, and the hash table over them (fdp_prebuild)
        void reset() { kfs.clear(); kf_slot.clear(); kf_T.clear(); ck.clear(); cl.clear(); pos.clear(); cpx.clear(); cmp.clear(); ent.clear(); tab.clear(); }
    };
    struct Pending : Gathered { int n = 0; size_t first_ref = 0; } pend;
    // candidates gathered while Matcher::SparseImageAlignment waited for its kernel (ygz_hip_set_wait_hook), for the launch that follows it
    struct Pre : Gathered { Frame *curr = nullptr; std::vector<Frame *> batch; bool valid = false; } pre;

    static size_t hash(const Frame *ref, const void *key)
    { uint64_t h = (uint64_t)(uintptr_t)key * 0x9E3779B97F4A7C15ull ^ (uint64_t)(uintptr_t)ref * 0xC2B2AE3D27D4EB4Full; return (size_t)(h ^ (h >> 29)); }
    void clear() { curr = nullptr; refs.clear(); feat_refs.clear(); entries.clear(); table.clear(); pend.n = 0; }   // (pre survives: it belongs to the frame about to begin)
    void begin(Frame *c)
    {   // a new current frame (or the same one with another pose): the answers of the last one are void, the keyframes it named are the guess
        if (!asked.empty()) asked_prev.swap(asked);
        asked.clear();
        clear();
        curr = c; c->_TCW.to7(T_cur);
    }
    bool valid_for(Frame *c) const { return curr == c && same7(c->_TCW, T_cur); }
    const Ref *ref_of(const Frame *f) const { for (const Ref &r : refs) if (r.f == f) return &r; return nullptr; }
    const Ref *feat_ref_of(const Frame *f) const { for (const Ref &r : feat_refs) if (r.f == f) return &r; return nullptr; }
    void note_asked(Frame *f) { for (Frame *a : asked) if (a == f) return; asked.push_back(f); }
    static void build_table(const std::vector<Entry> &ent, std::vector<int32_t> &tab)
    {
        size_t cap = 64;
        while (cap < 2 * ent.size() + 2) cap <<= 1;
        tab.assign(cap, -1);
        for (size_t i = 0; i < ent.size(); ++i) {
            size_t h = hash(ent[i].ref, ent[i].key) & (cap - 1);
            while (tab[h] >= 0) h = (h + 1) & (cap - 1);
            tab[h] = (int32_t)i;
        }
    }
    void rebuild_table() { build_table(entries, table); }
    const Entry *find(const Frame *ref, const void *key) const
    {
        if (table.empty()) return nullptr;
        const size_t mask = table.size() - 1;
        for (size_t h = hash(ref, key) & mask; table[h] >= 0; h = (h + 1) & mask) {
            const Entry &e = entries[table[h]];
            if (e.ref == ref && e.key == key) return &e;
        }
        return nullptr;
    }
};
FdpMemo &fdp_memo() { static FdpMemo m; return m; }
}  // namespace
void hip::fdp_memo_forget(const Frame *f)
{   // a frame that is (re)initialised or deleted takes every answer that involves it along
    FdpMemo &M = fdp_memo();
    auto drop = [&](std::vector<Frame *> &v) { v.erase(std::remove(v.begin(), v.end(), f), v.end()); };
    drop(M.asked); drop(M.asked_prev);
    M.pre.valid = false;
    if (M.curr == f || M.ref_of(f) || M.feat_ref_of(f)) M.clear();
}
void hip::SetFdpSpeculation(bool on) { fdp_memo().enabled = on; if (!on) fdp_memo().clear(); }
void hip::SetFdpBypass(bool on) { fdp_memo().bypass = on; }
hip::FdpMemoStats hip::GetFdpMemoStats() { return fdp_memo().st; }
void hip::ResetFdpMemoStats() { fdp_memo().st = hip::FdpMemoStats(); }

namespace {
// the answers of one launch over the gathered candidates become entries of the table
void fdp_absorb(FdpMemo &M, const std::vector<Frame *> &kfs, const std::vector<int32_t> &ck, const std::vector<int32_t> &cl, const std::vector<const MapPoint *> &cmp,
                const std::vector<double> &pos, const std::vector<double> &cpx, const std::vector<uint8_t> &vis, const std::vector<double> &proj,
                const std::vector<uint8_t> &ok, const std::vector<double> &out, const std::vector<int32_t> &sl)
{
    const int n = (int)ck.size();
    M.st.launches++; M.st.speculated += n;
    M.entries.reserve(M.entries.size() + n);
    for (int i = 0; i < n; ++i) {
        if (!vis[i]) continue;                                         // FindCandidates drops it (LocalMapping.cpp:60-63): nobody asks
        FdpMemo::Entry e;
        e.ref = kfs[ck[i]]; e.key = cmp[i];
        e.a[0] = pos[3 * i]; e.a[1] = pos[3 * i + 1]; e.a[2] = pos[3 * i + 2];
        e.px_ref[0] = cpx[2 * i]; e.px_ref[1] = cpx[2 * i + 1]; e.level = cl[i];
        e.px_in[0] = proj[2 * i]; e.px_in[1] = proj[2 * i + 1];
        e.px_out[0] = out[2 * i]; e.px_out[1] = out[2 * i + 1]; e.sl = sl[i]; e.ok = ok[i];
        M.entries.push_back(e);
    }
    M.rebuild_table();
}
// pure host work, no call into the context: every observation of a good map point in the keyframes of `batch` that hold an image (`skip_covered`: and
// are not in the table yet), with the keyframes' poses and the HBM slots they sit in as of now
void fdp_gather(const FdpMemo &M, const Frame *curr, const std::vector<Frame *> &batch, bool skip_covered, FdpMemo::Gathered &G)
{
    G.reset();
    const int levels = curr->_option._pyramid_level;
    for (Frame *r : batch)
        if (r != curr && !r->_pyramid.empty() && !(skip_covered && M.ref_of(r)) && std::find(G.kfs.begin(), G.kfs.end(), r) == G.kfs.end()) G.kfs.push_back(r);
    for (size_t k = 0; k < G.kfs.size(); ++k) {
        Frame *r = G.kfs[k];
        G.kf_slot.push_back(r->_hip_slot); double t7[7]; r->_TCW.to7(t7); G.kf_T.insert(G.kf_T.end(), t7, t7 + 7);
        const size_t n0 = r->_features.size();
        G.ck.reserve(G.ck.size() + n0); G.cl.reserve(G.cl.size() + n0); G.pos.reserve(G.pos.size() + 3 * n0); G.cpx.reserve(G.cpx.size() + 2 * n0); G.cmp.reserve(G.cmp.size() + n0);
        for (const Feature *f : r->_features) {
            const MapPoint *mp = f->_mappoint;
            if (!mp || mp->_bad) continue;
            // (whether f is the Feature the method reads, mp->_obs[ref->_keyframe_id] (Matcher.cpp:361), is settled at look-up time by comparing
            // pixel and level: a tree look-up per feature here was a third of the gather)
            if (f->_level < 0 || f->_level >= levels) continue;
            G.ck.push_back((int32_t)k); G.cl.push_back(f->_level); G.cmp.push_back(mp);
            G.pos.push_back(mp->_pos_world[0]); G.pos.push_back(mp->_pos_world[1]); G.pos.push_back(mp->_pos_world[2]);
            G.cpx.push_back(f->_pixel[0]); G.cpx.push_back(f->_pixel[1]);
        }
    }
}
// the table entries of the gathered candidates with everything that is known before the launch (the inputs the look-up compares), and the hash table over
// them -- host work that fits into the same wait as the gather; the launch's answers are filled in by fdp_collect
void fdp_prebuild(FdpMemo::Gathered &G)
{
    const size_t n = G.ck.size();
    G.ent.resize(n);
    const double nan = std::numeric_limits<double>::quiet_NaN();
    for (size_t i = 0; i < n; ++i) {
        FdpMemo::Entry &e = G.ent[i];
        e.ref = G.kfs[G.ck[i]]; e.key = G.cmp[i];
        e.a[0] = G.pos[3 * i]; e.a[1] = G.pos[3 * i + 1]; e.a[2] = G.pos[3 * i + 2];
        e.px_ref[0] = G.cpx[2 * i]; e.px_ref[1] = G.cpx[2 * i + 1]; e.level = G.cl[i];
        e.px_in[0] = e.px_in[1] = nan; e.px_out[0] = e.px_out[1] = 0; e.sl = 0; e.ok = 0;      // no prediction equals NaN: unanswered until collected
    }
    FdpMemo::build_table(G.ent, G.tab);
}
// the gathered candidates against `curr` in one launch, appended to the memo.  defer: the launch is queued and collected by fdp_collect at the first
// look-up (the caller's own FindCandidates runs in between)
void fdp_launch(FdpMemo &M, Frame *curr, FdpMemo::Gathered &G, bool defer)
{
    hip::Runtime &rt = hip::Runtime::Get();
    if (G.kfs.empty() || curr->_pyramid.empty()) return;
    const int cs = rt.Resident(curr);
    for (size_t k = 0; k < G.kfs.size(); ++k) if (rt.Resident(G.kfs[k]) != G.kf_slot[k] || G.kf_slot[k] < 0) return;   // (not resident when gathered, or more keyframes than HBM slots: no speculation)
    if (curr->_hip_slot != cs) return;
    for (size_t k = 0; k < G.kfs.size(); ++k) if (G.kfs[k]->_hip_slot != G.kf_slot[k]) return;
    const size_t first_ref = M.refs.size();
    for (size_t k = 0; k < G.kfs.size(); ++k) { FdpMemo::Ref R; R.f = G.kfs[k]; memcpy(R.T, &G.kf_T[7 * k], 56); M.refs.push_back(R); }
    const int n = (int)G.ck.size();
    if (n == 0) return;
    if (defer) {
        if (ygz_hip_find_direct_projection_mp_begin(rt.ctx(), cs, M.T_cur, (int)G.kfs.size(), G.kf_slot.data(), G.kf_T.data(), n, G.ck.data(), G.pos.data(), G.cpx.data(),
                                                    G.cl.data()) != YGZ_OK) { M.refs.resize(first_ref); return; }
        FdpMemo::Pending &P = M.pend;
        P.n = n; P.first_ref = first_ref;
        P.kfs.swap(G.kfs); P.kf_slot.swap(G.kf_slot); P.kf_T.swap(G.kf_T); P.ck.swap(G.ck); P.cl.swap(G.cl); P.pos.swap(G.pos); P.cpx.swap(G.cpx); P.cmp.swap(G.cmp);
        P.ent.swap(G.ent); P.tab.swap(G.tab);
        return;
    }
    std::vector<uint8_t> vis(n), ok(n); std::vector<double> proj(2 * (size_t)n), out(2 * (size_t)n); std::vector<int32_t> sl(n);
    if (ygz_hip_find_direct_projection_mp(rt.ctx(), cs, M.T_cur, (int)G.kfs.size(), G.kf_slot.data(), G.kf_T.data(), n, G.ck.data(), G.pos.data(), G.cpx.data(),
                                          G.cl.data(), nullptr, vis.data(), proj.data(), ok.data(), out.data(), sl.data()) != YGZ_OK) {
        M.refs.resize(first_ref);                                      // nothing learnt; the calls take the n = 1 path (and report the error there)
        return;
    }
    fdp_absorb(M, G.kfs, G.ck, G.cl, G.cmp, G.pos, G.cpx, vis, proj, ok, out, sl);
}
void fdp_speculate_mp(FdpMemo &M, Frame *curr, const std::vector<Frame *> &batch)
{
    struct Clock { double &acc; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
                   ~Clock() { acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } } clock_{ M.st.speculate_ms };
    FdpMemo::Gathered G;
    fdp_gather(M, curr, batch, true, G);
    fdp_launch(M, curr, G, false);
}
// the queued launch, waited for and turned into table entries (first look-up of the frame)
void fdp_collect(FdpMemo &M)
{
    struct Clock { double &acc; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
                   ~Clock() { acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } } clock_{ M.st.speculate_ms };
    FdpMemo::Pending &P = M.pend;
    const int n = P.n;
    P.n = 0;
    if (n <= 0) return;
    std::vector<uint8_t> vis(n), ok(n); std::vector<double> proj(2 * (size_t)n), out(2 * (size_t)n); std::vector<int32_t> sl(n);
    if (ygz_hip_find_direct_projection_mp_end(hip::Runtime::Get().ctx(), n, vis.data(), proj.data(), ok.data(), out.data(), sl.data()) != YGZ_OK) {
        M.refs.resize(P.first_ref);                                    // (another _begin took its place, or the run failed): nothing learnt
        return;
    }
    if (P.ent.size() == (size_t)n && M.entries.empty()) {              // entries and table were prepared with the gather: the answers go in, the table is adopted
        M.st.launches++; M.st.speculated += n;
        for (int i = 0; i < n; ++i) {
            if (!vis[i]) continue;                                     // FindCandidates drops it (LocalMapping.cpp:60-63): nobody asks; its prediction stays NaN
            FdpMemo::Entry &e = P.ent[i];
            e.px_in[0] = proj[2 * i]; e.px_in[1] = proj[2 * i + 1];
            e.px_out[0] = out[2 * i]; e.px_out[1] = out[2 * i + 1]; e.sl = sl[i]; e.ok = ok[i];
        }
        M.entries.swap(P.ent); M.table.swap(P.tab);
        return;
    }
    fdp_absorb(M, P.kfs, P.ck, P.cl, P.cmp, P.pos, P.cpx, vis, proj, ok, out, sl);
}
// Queue the frame's speculative launch as soon as its pose is known -- the end of Matcher::SparseImageAlignment -- when the previous current frame was
// served per candidate (an unchanged caller: Tracker -> LocalMapping::TrackLocalMap, LocalMapping.cpp:24-33).  LocalMapping::FindCandidates (0.2 ms of the
// caller's std::map work per frame) then runs while the device evaluates the candidates; a caller that changes the pose afterwards, or asks about other
// keyframes, falls back to the launch at its first call as before.
// the wait hook of Matcher::SparseImageAlignment (called by ygz_hip_sparse_align between its launch and its wait): the gather of the launch that follows,
// which does not depend on the pose being estimated
void fdp_pregather_hook(void *user)
{
    FdpMemo &M = fdp_memo();
    Frame *curr = static_cast<Frame *>(user);
    struct Clock { double &acc; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
                   ~Clock() { acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } } clock_{ M.st.speculate_ms };
    M.pre.valid = false; M.pre.curr = curr; M.pre.batch = M.asked;
    fdp_gather(M, curr, M.pre.batch, false, M.pre);
    fdp_prebuild(M.pre);
    M.pre.valid = true;
}
bool fdp_prelaunch_wanted(Frame *curr)
{
    FdpMemo &M = fdp_memo();
    return M.enabled && M.prelaunch && !M.bypass && !M.asked.empty() && !M.valid_for(curr);
}
void fdp_prelaunch(Frame *curr)
{
    FdpMemo &M = fdp_memo();
    if (!fdp_prelaunch_wanted(curr)) { M.pre.valid = false; return; }
    struct Clock { double &acc; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
                   ~Clock() { acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } } clock_{ M.st.speculate_ms };
    M.begin(curr);
    if (!(M.pre.valid && M.pre.curr == curr && M.pre.batch == M.asked_prev)) { fdp_gather(M, curr, M.asked_prev, false, M.pre); fdp_prebuild(M.pre); }   // (the hook did not run: now)
    M.pre.valid = false;
    fdp_launch(M, curr, M.pre, true);
}
// Feature overload: the pairs the same Matcher's last SearchForTriangulation(ref, curr, ...) returned, with the depth and prediction
// LocalMapping::CreateNewMapPoints forms from them before it calls (src/Module/LocalMapping.cpp:405-447): both features without a map point, rays not
// parallel (cos < 0.9998), DepthFromTriangulation(T12^-1, pt1, pt2) positive -> fea1->_depth = depth1, prediction = fea2->_pixel.  One launch; a call is
// answered only if its feature, depth and prediction equal the speculated ones bit for bit.
void fdp_speculate_feat(FdpMemo &M, Frame *ref, Frame *curr, const vector<pair<int, int>> &pairs)
{
    struct Clock { double &acc; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
                   ~Clock() { acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } } clock_{ M.st.speculate_ms };
    hip::Runtime &rt = hip::Runtime::Get();
    PinholeCamera *cam = Frame::_camera;
    FdpMemo::Ref R; R.f = ref; ref->_TCW.to7(R.T);
    M.feat_refs.push_back(R);
    if (!cam || pairs.empty() || ref->_pyramid.empty() || curr->_pyramid.empty()) return;
    const int levels = curr->_option._pyramid_level;
    const SE3 T12 = ref->_TCW * curr->_TCW.inverse(), T21 = T12.inverse();
    std::vector<const Feature *> cf; std::vector<double> pr, dep, pc; std::vector<int32_t> lvl;
    for (const auto &pq : pairs) {
        if (pq.first < 0 || pq.second < 0 || pq.first >= (int)ref->_features.size() || pq.second >= (int)curr->_features.size()) continue;
        const Feature *fea1 = ref->_features[pq.first], *fea2 = curr->_features[pq.second];
        if (fea1->_mappoint || fea2->_mappoint || fea1->_level < 0 || fea1->_level >= levels) continue;
        const Vector3d pt1 = cam->Pixel2Camera(fea1->_pixel), pt2 = cam->Pixel2Camera(fea2->_pixel);
        if (pt1.dot(pt2) / (pt1.norm() * pt2.norm()) >= 0.9998) continue;
        double d1 = 0, d2 = 0;
        if (!cvutils::DepthFromTriangulation(T21, pt1, pt2, d1, d2) || d1 < 0 || d2 < 0) continue;
        cf.push_back(fea1); dep.push_back(d1); lvl.push_back(fea1->_level);
        pr.push_back(fea1->_pixel[0]); pr.push_back(fea1->_pixel[1]); pc.push_back(fea2->_pixel[0]); pc.push_back(fea2->_pixel[1]);
    }
    const int n = (int)cf.size();
    if (n == 0) return;
    ygz_align_pair pair;
    pair.ref_slot = rt.Resident(ref); pair.cur_slot = rt.Resident(curr);
    if (ref->_hip_slot != pair.ref_slot || pair.ref_slot < 0 || pair.cur_slot < 0) return;
    memcpy(pair.T_ref, R.T, 56); memcpy(pair.T_cur, M.T_cur, 56);
    std::vector<double> out = pc; std::vector<int32_t> sl(n); std::vector<uint8_t> ok(n);
    if (ygz_hip_find_direct_projection(rt.ctx(), &pair, pr.data(), dep.data(), lvl.data(), out.data(), sl.data(), ok.data(), n) != YGZ_OK) return;
    M.st.launches++; M.st.speculated += n;
    for (int i = 0; i < n; ++i) {
        FdpMemo::Entry e;
        e.ref = ref; e.key = cf[i];
        e.a[0] = dep[i]; e.a[1] = e.a[2] = 0;
        e.px_ref[0] = pr[2 * i]; e.px_ref[1] = pr[2 * i + 1]; e.level = lvl[i];
        e.px_in[0] = pc[2 * i]; e.px_in[1] = pc[2 * i + 1];
        e.px_out[0] = out[2 * i]; e.px_out[1] = out[2 * i + 1]; e.sl = sl[i]; e.ok = ok[i];
        M.entries.push_back(e);
    }
    M.rebuild_table();
}
}  // namespace

int Matcher::FindDirectProjectionBatch(Frame *ref, Frame *curr, const vector<Feature *> &feas, vector<Vector2d> &px_curr,
                                       vector<int> &search_level, vector<bool> &ok)
{
    hip::Runtime &rt = hip::Runtime::Get();
    const int n = (int)feas.size();
    search_level.assign(n, 0); ok.assign(n, false);
    if (n == 0) return 0;
    ygz_align_pair pair;
    pair.ref_slot = rt.Resident(ref); pair.cur_slot = rt.Resident(curr);
    ref->_TCW.to7(pair.T_ref); curr->_TCW.to7(pair.T_cur);
    int good = 0;
    const int cells = rt.cells();
    for (int base = 0; base < n; base += cells) {
        const int m = std::min(cells, n - base);
        vector<double> pr(2 * (size_t)m), dep(m), pc(2 * (size_t)m); vector<int32_t> lvl(m), sl(m); vector<uint8_t> o(m);
        for (int i = 0; i < m; ++i) {
            const Feature *f = feas[base + i];
            pr[2 * i] = f->_pixel[0]; pr[2 * i + 1] = f->_pixel[1]; dep[i] = f->_depth; lvl[i] = f->_level;
            pc[2 * i] = px_curr[base + i][0]; pc[2 * i + 1] = px_curr[base + i][1];
        }
        if (!hip::check(ygz_hip_find_direct_projection(rt.ctx(), &pair, pr.data(), dep.data(), lvl.data(), pc.data(), sl.data(), o.data(), m), "find_direct_projection")) return good;
        for (int i = 0; i < m; ++i) {
            px_curr[base + i] = Vector2d(pc[2 * i], pc[2 * i + 1]); search_level[base + i] = sl[i]; ok[base + i] = o[i] != 0; good += o[i] != 0;
        }
    }
    return good;
}

bool Matcher::FindDirectProjection(Frame *ref, Frame *curr, Feature *fea_ref, Vector2d &px_curr, int &search_level)
{   // Matcher.cpp:385-417.  Called once per matched pair by LocalMapping::CreateNewMapPoints (:447): answered from one launch over the pairs of this
    // object's last SearchForTriangulation(ref, curr, ...) when every input equals the speculated one bit for bit (FdpMemo above), else n = 1
    if (fea_ref->_depth < 0) { LOG(WARNING) << "invalid depth: " << fea_ref->_depth << endl; return false; }
    assert(fea_ref->_frame == ref);
    FdpMemo &M = fdp_memo();
    if (M.enabled && !M.bypass && ref == _tri_kf1 && curr == _tri_kf2) {
        if (!M.valid_for(curr)) M.begin(curr);
        const FdpMemo::Ref *R = M.feat_ref_of(ref);
        if (R && !same7(ref->_TCW, R->T)) { Frame *c = curr; M.clear(); M.curr = c; c->_TCW.to7(M.T_cur); R = nullptr; }
        if (!R) fdp_speculate_feat(M, ref, curr, _tri_pairs);
        const FdpMemo::Entry *e = M.find(ref, fea_ref);
        if (e && e->a[0] == fea_ref->_depth && e->px_ref[0] == fea_ref->_pixel[0] && e->px_ref[1] == fea_ref->_pixel[1] && e->level == fea_ref->_level
              && e->px_in[0] == px_curr[0] && e->px_in[1] == px_curr[1]) {
            M.st.hits++;
            px_curr = Vector2d(e->px_out[0], e->px_out[1]); search_level = e->sl;
            return e->ok != 0;
        }
        M.st.single++;
    }
    vector<Vector2d> px(1, px_curr); vector<int> sl; vector<bool> ok;
    FindDirectProjectionBatch(ref, curr, vector<Feature *>(1, fea_ref), px, sl, ok);
    px_curr = px[0]; search_level = sl[0];
    return ok[0];
}

bool Matcher::FindDirectProjection(Frame *ref, Frame *curr, MapPoint *mp, Vector2d &px_curr, int &search_level)
{   // Matcher.cpp:356-383.  The reference does not test the sign of the depth in this overload; neither does the kernel (k_lmap_match)
    Feature *fea = mp->_obs[ref->_keyframe_id];
    if (!fea) {                                               // (the reference dereferences the null operator[] just inserted, Matcher.cpp:361: defined here as "no projection")
        LOG(WARNING) << "Matcher::FindDirectProjection: map point " << mp->_id << " has no observation in keyframe " << ref->_keyframe_id << endl;
        return false;
    }
    FdpMemo &M = fdp_memo();
    if (M.enabled && !M.bypass) {
        if (!M.valid_for(curr)) M.begin(curr);
        if (M.pend.n) fdp_collect(M);                                   // the launch Matcher::SparseImageAlignment queued for this frame
        M.note_asked(ref);
        for (int pass = 0; pass < 2; ++pass) {
            const FdpMemo::Ref *R = M.ref_of(ref);
            if (R && !same7(ref->_TCW, R->T)) { Frame *c = curr; M.clear(); M.curr = c; c->_TCW.to7(M.T_cur); R = nullptr; }   // the keyframe moved (local BA)
            if (R) {
                const FdpMemo::Entry *e = M.find(ref, mp);
                if (e && e->a[0] == mp->_pos_world[0] && e->a[1] == mp->_pos_world[1] && e->a[2] == mp->_pos_world[2]
                      && e->px_ref[0] == fea->_pixel[0] && e->px_ref[1] == fea->_pixel[1] && e->level == fea->_level
                      && e->px_in[0] == px_curr[0] && e->px_in[1] == px_curr[1]) {
                    M.st.hits++;
                    px_curr = Vector2d(e->px_out[0], e->px_out[1]); search_level = e->sl;
                    return e->ok != 0;
                }
                break;                                                  // covered keyframe, unknown or changed candidate: n = 1
            }
            if (pass == 0) {
                std::vector<Frame *> batch(1, ref);
                if (M.refs.empty()) batch.insert(batch.end(), M.asked_prev.begin(), M.asked_prev.end());
                fdp_speculate_mp(M, curr, batch);
            }
        }
        M.st.single++;
    }
    hip::Runtime &rt = hip::Runtime::Get();
    const int32_t kf_slot = rt.Resident(ref), cs = rt.Resident(curr), ck = 0, lvl = fea->_level;
    double Tr[7], Tc[7]; ref->_TCW.to7(Tr); curr->_TCW.to7(Tc);
    double out[2] = { 0, 0 }; const double pin[2] = { px_curr[0], px_curr[1] };
    int32_t sl = 0; uint8_t ok = 0;
    if (!hip::check(ygz_hip_find_direct_projection_mp(rt.ctx(), cs, Tc, 1, &kf_slot, Tr, 1, &ck, mp->_pos_world.data(), fea->_pixel.data(), &lvl, pin,
                                                      nullptr, nullptr, &ok, out, &sl), "find_direct_projection_mp")) return false;
    px_curr = Vector2d(out[0], out[1]); search_level = sl;
    return ok != 0;
}

int Matcher::ProjectMapPoints(Frame *current, const std::set<Frame *> &local_keyframes, const std::set<MapPoint *> &local_map_points)
{
    g_pmp_trace.start();
    hip::Runtime &rt = hip::Runtime::Get();
    vector<Frame *> kfs(local_keyframes.begin(), local_keyframes.end());
    vector<int32_t> kf_slot; vector<double> kf_T;
    for (Frame *f : kfs) {
        kf_slot.push_back(rt.Resident(f));
        double t7[7]; f->_TCW.to7(t7); kf_T.insert(kf_T.end(), t7, t7 + 7);
    }
    // which local keyframe a feature belongs to: a handful of keyframes (LocalMapping.local_keyframes: 3) -- a linear scan of the pointer array,
    // not a std::map look-up per observation (thousands per frame)
    const int n_kfs = (int)kfs.size();
    Frame *const *kfp = kfs.data();
    auto kf_of = [&](const Frame *f) { for (int k = 0; k < n_kfs; ++k) if (kfp[k] == f) return k; return -1; };
    vector<MapPoint *> mps(local_map_points.begin(), local_map_points.end());
    vector<double> pos; vector<uint8_t> bad;
    vector<int32_t> cp, ck, cl; vector<double> cpx, cscore;              // (the candidate's score rides along: reading it from the matched Feature later is a cache miss per match)
    { const size_t P0 = mps.size(), C0 = P0 * kfs.size();
      pos.reserve(3 * P0); bad.reserve(P0); cp.reserve(C0); ck.reserve(C0); cl.reserve(C0); cpx.reserve(2 * C0); cscore.reserve(C0); }
    for (size_t p = 0; p < mps.size(); ++p) {
        MapPoint *mp = mps[p];
        pos.push_back(mp->_pos_world[0]); pos.push_back(mp->_pos_world[1]); pos.push_back(mp->_pos_world[2]);
        bad.push_back(mp->_bad ? 1 : 0);
        if (mp->_bad) continue;
        for (auto &obs : mp->_obs) {                                   // LocalMapping.cpp:66-75
            Feature *fea = obs.second;
            const int k = fea ? kf_of(fea->_frame) : -1;
            if (k < 0) continue;
            cp.push_back((int32_t)p); ck.push_back(k); cl.push_back(fea->_level);
            cpx.push_back(fea->_pixel[0]); cpx.push_back(fea->_pixel[1]); cscore.push_back(fea->_score);
        }
    }
    const int P = (int)mps.size();
    if (P == 0) return 0;
    ygz_local_map m;
    m.n_points = P; m.pos_world = pos.data(); m.point_bad = bad.data();
    m.n_keyframes = (int)kf_slot.size(); m.kf_slot = kf_slot.data(); m.kf_T = kf_T.data();
    m.n_candidates = (int)cp.size(); m.cand_point = cp.data(); m.cand_kf = ck.data(); m.cand_level = cl.data(); m.cand_px_ref = cpx.data();
    vector<uint8_t> in_view(P); vector<double> px_proj(2 * (size_t)P), px_match(2 * (size_t)P); vector<int32_t> match(P), level(P);
    double Tc[7]; current->_TCW.to7(Tc);
    int32_t n = 0;
    g_pmp_trace.lap(0);
    if (!hip::check(ygz_hip_track_local_map(rt.ctx(), rt.Resident(current), Tc, &m, in_view.data(), px_proj.data(), match.data(), px_match.data(),
                                            level.data(), &n), "track_local_map")) return 0;
    g_pmp_trace.lap(1);
    current->_features.reserve(current->_features.size() + (size_t)n);
    for (int p = 0; p < P; ++p) {
        MapPoint *mp = mps[p];
        if (mp->_bad) continue;
        if (!in_view[p]) { mp->_track_in_view = false; continue; }     // :59-62
        mp->_cnt_visible++;                                            // :64
        if (match[p] < 0) continue;
        Feature *feature = new Feature(Vector2d(px_match[2 * p], px_match[2 * p + 1]), level[p], cscore[match[p]]);   // :104-111
        feature->_frame = current;
        feature->_mappoint = mp;
        current->_features.push_back(feature);
    }
    g_pmp_trace.lap(2); g_pmp_trace.done();
    return (int)n;
}

bool Matcher::SparseImageAlignment(Frame *ref, Frame *current)
{
    current->_TCW = ref->_TCW;
    // (an unchanged caller's next step is LocalMapping::TrackLocalMap: the candidates of its speculative FindDirectProjection launch are gathered while
    // the alignment kernel runs and the launch is queued as soon as the pose is known -- fdp_prelaunch)
    const bool pre = fdp_prelaunch_wanted(current);
    if (pre) ygz_hip_set_wait_hook(hip::Runtime::Get().ctx(), &fdp_pregather_hook, current);
    _align->run(ref, current);
    if (pre) ygz_hip_set_wait_hook(hip::Runtime::Get().ctx(), nullptr, nullptr);
    _TCR_esti = current->_TCW * ref->_TCW.inverse();
    if (_TCR_esti.log().norm() > _options._max_alignment_motion) {
        LOG(WARNING) << "Too large motion: " << _TCR_esti.log().norm() << ". Reject this estimation. " << endl;
        _TCR_esti = SE3();
        current->_TCW = ref->_TCW;
        fdp_prelaunch(current);
        return false;
    }
    fdp_prelaunch(current);
    return true;
}

// ------------------------------------------------------------------------------------------ cvutils
namespace cvutils {
int Align2DBatch(const cv::Mat &cur_img, const uint8_t *pwb, int n, const int n_iter, vector<Vector2d> &px, vector<bool> &ok)
{
    hip::Runtime &rt = hip::Runtime::Get();
    Frame *f = nullptr; int level = 0;
    ok.assign(n, false);
    if (!rt.FindLevel(cur_img.data, &f, &level)) { LOG(ERROR) << "cvutils::Align2D: cur_img is not a pyramid level of an initialised Frame" << endl; return 0; }
    const int slot = rt.Resident(f);
    vector<double> uv(2 * (size_t)n); vector<uint8_t> o(n);
    for (int i = 0; i < n; ++i) { uv[2 * i] = px[i][0]; uv[2 * i + 1] = px[i][1]; }
    if (!hip::check(ygz_hip_align2d(rt.ctx(), slot, level, pwb, nullptr, uv.data(), o.data(), nullptr, n, n_iter), "align2d")) return 0;
    int good = 0;
    for (int i = 0; i < n; ++i) { px[i] = Vector2d(uv[2 * i], uv[2 * i + 1]); ok[i] = o[i] != 0; good += o[i] != 0; }
    return good;
}
bool Align2D(const cv::Mat &cur_img, uint8_t *ref_patch_with_border, uint8_t *, const int n_iter, Vector2d &cur_px_estimate, bool)
{
    vector<Vector2d> px(1, cur_px_estimate); vector<bool> ok;
    Align2DBatch(cur_img, ref_patch_with_border, 1, n_iter, px, ok);
    cur_px_estimate = px[0];
    return ok[0];
}
bool DepthFromTriangulation(const SE3 &T_search_ref, const Vector3d &f_ref, const Vector3d &f_cur, double &depth1, double &depth2, const double &determinant_th)
{   // CVUtils.h:18-38
    const Vector3d a0 = T_search_ref.rotation_matrix() * f_ref, a1 = -f_cur;
    const double m00 = a0.dot(a0), m01 = a0.dot(a1), m11 = a1.dot(a1);
    const double det = m00 * m11 - m01 * m01;
    if (det < determinant_th) return false;
    const Vector3d t = T_search_ref.translation();
    const double b0 = a0.dot(t), b1 = a1.dot(t);
    const double d0 = -(m11 * b0 - m01 * b1) / det, d1 = -(-m01 * b0 + m00 * b1) / det;
    depth1 = fabs(d0); depth2 = fabs(d1);
    return true;
}
}  // namespace cvutils

// ------------------------------------------------------------------------------------------ ba::LocalBAG2O
namespace ba {
void LocalBAG2O(std::set<Frame *> &local_keyframes, std::set<MapPoint *> &local_map_points) { LocalBAG2O(local_keyframes, local_map_points, nullptr); }

namespace { PhaseTrace g_ba_trace{ "LocalBAG2O", { "graph", "ygz_hip_ba_optimize_chi2", "write-back", nullptr } }; }
void LocalBAG2O(std::set<Frame *> &local_keyframes, std::set<MapPoint *> &local_map_points, LocalBAStats *stats)
{   // graph build exactly as src/Algorithm/BA.cpp:397-497, then optimize(20) and the write-back of :504-541
    hip::Runtime &rt = hip::Runtime::Get();
    g_ba_trace.start();
    std::map<unsigned long, int> pose_index;            // keyframe id -> vertex
    std::vector<Frame *> pose_frame;
    std::vector<double> poses; std::vector<uint8_t> fixed;
    auto add_pose = [&](Frame *frame, bool fix) {
        pose_index[frame->_keyframe_id] = (int)pose_frame.size();
        pose_frame.push_back(frame);
        const Vector6d lg = frame->_TCW.log();            // esti = [log.tail<3>(); log.head<3>()]  (BA.cpp:407-409)
        for (int i = 0; i < 3; ++i) poses.push_back(lg[3 + i]);
        for (int i = 0; i < 3; ++i) poses.push_back(lg[i]);
        fixed.push_back(fix ? 1 : 0);
    };
    for (Frame *frame : local_keyframes) add_pose(frame, frame->_keyframe_id == 0);
    std::vector<MapPoint *> pts; std::vector<double> points;
    std::vector<int32_t> edge_pose, edge_point; std::vector<double> obs; std::vector<Feature *> features;
    // keyframe id -> vertex, resolved once per keyframe (a window has a handful): per edge the reference walks three trees (Memory::GetKeyFrame,
    // local_keyframes.find, the vertex map) -- 9000 look-ups per local BA of the surface loop
    struct KfSeen { unsigned long id; int vertex; };
    std::vector<KfSeen> seen;
    auto vertex_of = [&](unsigned long kf_id) {
        for (const KfSeen &k : seen) if (k.id == kf_id) return k.vertex;
        Frame *frame = Memory::GetKeyFrame(kf_id);
        assert(frame != nullptr);
        if (local_keyframes.find(frame) == local_keyframes.end()) {
            // keyframes that see local map points but are not local: fixed (BA.cpp:458-477)
            auto it = pose_index.find(frame->_keyframe_id);
            if (it == pose_index.end()) add_pose(frame, true);
            else fixed[it->second] = 1;
        }
        const int v = pose_index[frame->_keyframe_id];
        seen.push_back({ kf_id, v });
        return v;
    };
    for (MapPoint *mp : local_map_points) {
        if (mp->_bad) continue;
        const int il = (int)pts.size();
        pts.push_back(mp);
        for (int i = 0; i < 3; ++i) points.push_back(mp->_pos_world[i]);
        for (auto &obs_pair : mp->_obs) {
            if (obs_pair.second->_bad) continue;
            edge_pose.push_back(vertex_of(obs_pair.first));
            edge_point.push_back(il);
            obs.push_back(obs_pair.second->_pixel[0]); obs.push_back(obs_pair.second->_pixel[1]);
            features.push_back(obs_pair.second);
        }
    }
    if (pts.empty() || edge_pose.empty()) return;
    ygz_ba_problem pb;
    memset(&pb, 0, sizeof(pb));
    pb.n_poses = (int)pose_frame.size(); pb.n_points = (int)pts.size(); pb.n_edges = (int)edge_pose.size();
    pb.poses = poses.data(); pb.pose_fixed = fixed.data(); pb.points = points.data();
    pb.edge_pose = edge_pose.data(); pb.edge_point = edge_point.data(); pb.obs = obs.data();
    PinholeCamera *cam = Frame::_camera;
    pb.fx = cam->fx(); pb.fy = cam->fy(); pb.cx = cam->cx(); pb.cy = cam->cy();      // EdgeSophusSE3ProjectXYZ::setCamera
    pb.huber_delta = 5.991; pb.formulation = 0;                                       // BA.cpp:451
    ygz_ba_stats st;
    // optimize(20), then the inlier test on the optimised state: chi2 > 5.991 -> Feature::_bad (BA.cpp:501-515) -- one call, the graph is uploaded once
    std::vector<double> chi2_edge(pb.n_edges);
    g_ba_trace.lap(0);
    if (!hip::check(ygz_hip_ba_optimize_chi2(rt.ctx(), &pb, poses.data(), points.data(), 20, &st, chi2_edge.data()), "ba_optimize_chi2")) return;      // the map as it was
    g_ba_trace.lap(1);
    int cntOutlier = 0;
    for (size_t i = 0; i < features.size(); ++i) if (chi2_edge[i] > 5.991) { cntOutlier++; features[i]->_bad = true; }
    for (Frame *frame : local_keyframes) {             // BA.cpp:520-531
        const double *p = &poses[6 * (size_t)pose_index[frame->_keyframe_id]];
        Vector6d pose;
        for (int i = 0; i < 3; ++i) { pose[i] = p[3 + i]; pose[3 + i] = p[i]; }
        frame->_TCW = SE3::exp(pose);
    }
    for (size_t l = 0; l < pts.size(); ++l) pts[l]->_pos_world = Vector3d(points[3 * l], points[3 * l + 1], points[3 * l + 2]);
    if (stats) { stats->iterations = st.iterations; stats->lm_trials = st.lm_trials; stats->outliers = cntOutlier; stats->chi2_initial = st.chi2_initial; stats->chi2_final = st.chi2_final; }
    g_ba_trace.lap(2); g_ba_trace.done();
}
// ---- the ceres-based entry points (BA.cpp:11-384) ------------------------------------------------------------------
namespace {
// one ceres::Problem as arrays: parameter blocks = poses [t; angle-axis] and points, residual blocks = edges
struct CeresArrays {
    std::vector<double> poses, points, obs, huber;
    std::vector<uint8_t> pose_fixed, point_fixed;
    std::vector<int32_t> edge_pose, edge_point;
    bool any_huber = false;
    int add_pose(const SE3 &T, bool fixed)
    {   // pose.head<3>() = t, pose.tail<3>() = so3().log()  (BA.cpp:96-99)
        const Vector3d t = T.translation(), r = T.so3().log();
        for (int i = 0; i < 3; ++i) poses.push_back(t[i]);
        for (int i = 0; i < 3; ++i) poses.push_back(r[i]);
        pose_fixed.push_back(fixed ? 1 : 0);
        return (int)pose_fixed.size() - 1;
    }
    int add_point(const Vector3d &p, bool fixed)
    {
        for (int i = 0; i < 3; ++i) points.push_back(p[i]);
        point_fixed.push_back(fixed ? 1 : 0);
        return (int)point_fixed.size() - 1;
    }
    void add_edge(int ip, int il, const Vector2d &obs_n, double huber_a)
    {
        edge_pose.push_back(ip); edge_point.push_back(il); obs.push_back(obs_n[0]); obs.push_back(obs_n[1]);
        huber.push_back(huber_a); any_huber |= huber_a > 0;
    }
    SE3 pose(int k) const
    {   // SE3(SO3::exp(pose.tail<3>()), pose.head<3>())  (BA.cpp:65,146)
        const double *p = &poses[6 * (size_t)k];
        return SE3(SO3::exp(Vector3d(p[3], p[4], p[5])), Vector3d(p[0], p[1], p[2]));
    }
    bool solve(ygz_ceres_summary *sum = nullptr, int trust_region_strategy = YGZ_CERES_LEVENBERG_MARQUARDT)
    {
        if (edge_pose.empty()) return false;
        ygz_ba_problem pb; memset(&pb, 0, sizeof(pb));
        pb.n_poses = (int)pose_fixed.size(); pb.n_points = (int)point_fixed.size(); pb.n_edges = (int)edge_pose.size();
        pb.poses = poses.data(); pb.pose_fixed = pose_fixed.data(); pb.points = points.data(); pb.point_fixed = point_fixed.data();
        pb.edge_pose = edge_pose.data(); pb.edge_point = edge_point.data(); pb.obs = obs.data();
        pb.edge_huber = any_huber ? huber.data() : nullptr; pb.formulation = 2;
        ygz_ceres_summary s;
        ygz_ceres_options opt;
        ygz_hip_ceres_default_options(&opt);
        opt.trust_region_strategy = trust_region_strategy;
        if (!hip::check(ygz_hip_ba_solve_ceres(hip::Runtime::Get().ctx(), &pb, poses.data(), points.data(), &opt, &s), "ba_solve_ceres")) return false;
        if (sum) *sum = s;
        return s.termination != YGZ_CERES_FAILURE;
    }
};
}  // namespace

void TwoViewBACeres(const SE3 &ref, SE3 &curr, const vector<Vector2d> px_ref, const vector<Vector2d> px_curr,
                    vector<bool> &inlier, vector<Vector3d> &pts_ref)
{   // BA.cpp:11-89
    // options.trust_region_strategy_type = ceres::DOGLEG (BA.cpp:58-62): the DoglegStrategy restatement (round 6; the loop runs on the host around the
    // GPU linearisations -- this is called once per sequence, by the Initializer)
    assert(px_ref.size() == px_curr.size());
    PinholeCamera *cam = Frame::GetCamera();
    assert(cam != nullptr);
    CeresArrays A;
    const int kr = A.add_pose(ref, true), kc = A.add_pose(curr, false);      // ref: PointOnly functor; curr: pose + point
    for (size_t i = 0; i < px_ref.size(); ++i) {
        if (inlier[i] == false) pts_ref[i] = Vector3d(0, 0, 1);
        const int il = A.add_point(pts_ref[i], false);
        const double a = inlier[i] ? 0.0 : 0.1;                                // HuberLoss(0.1) on the outliers only
        A.add_edge(kr, il, cam->Pixel2Camera2D(px_ref[i]), a);
        A.add_edge(kc, il, cam->Pixel2Camera2D(px_curr[i]), a);
    }
    A.solve(nullptr, YGZ_CERES_DOGLEG);
    curr = A.pose(kc);
    const double ch2 = 5.991;
    for (size_t i = 0; i < px_ref.size(); ++i) {
        pts_ref[i] = Vector3d(A.points[3 * i], A.points[3 * i + 1], A.points[3 * i + 2]);
        const Vector2d e1 = px_ref[i] - cam->World2Pixel(pts_ref[i], ref), e2 = px_curr[i] - cam->World2Pixel(pts_ref[i], curr);
        const double depth1 = cam->World2Camera(pts_ref[i], ref)[2], depth2 = cam->World2Camera(pts_ref[i], curr)[2];
        if (e1.dot(e1) > ch2 || e2.dot(e2) > ch2) inlier[i] = false;
        else if (depth1 < 0 || depth2 < 0) inlier[i] = false;
        else inlier[i] = true;
    }
}

void OptimizeCurrent(Frame *current)
{   // BA.cpp:91-186: current pose + its map points, every observing keyframe constant, HuberLoss(0.1) everywhere
    const float chi2Mono = 5.991 * 4;
    CeresArrays A;
    const int kc = A.add_pose(current->_TCW, false);
    std::map<unsigned long, int> kf_index;
    std::map<MapPoint *, int> pt_index;                  // one parameter block per _pos_world.data()
    for (Feature *fea : current->_features) {
        assert(fea->_mappoint != nullptr);
        auto it = pt_index.find(fea->_mappoint);
        const int il = it != pt_index.end() ? it->second : (pt_index[fea->_mappoint] = A.add_point(fea->_mappoint->_pos_world, false));
        A.add_edge(kc, il, Frame::_camera->Pixel2Camera2D(fea->_pixel), 0.1);
        for (auto &obs_pair : fea->_mappoint->_obs) {
            Frame *frame = Memory::GetKeyFrame(obs_pair.first);
            auto kt = kf_index.find(obs_pair.first);
            const int k = kt != kf_index.end() ? kt->second : (kf_index[obs_pair.first] = A.add_pose(frame->_TCW, true));
            A.add_edge(k, il, Frame::_camera->Pixel2Camera2D(obs_pair.second->_pixel), 0.1);
        }
    }
    A.solve();
    current->_TCW = A.pose(kc);
    for (auto &pi : pt_index) pi.first->_pos_world = Vector3d(A.points[3 * pi.second], A.points[3 * pi.second + 1], A.points[3 * pi.second + 2]);
    for (Feature *fea : current->_features) {
        const Vector2d delta = Frame::_camera->World2Pixel(fea->_mappoint->_pos_world, current->_TCW) - fea->_pixel;
        if (delta.dot(delta) > chi2Mono) fea->_bad = true;
        else fea->_depth = Frame::_camera->World2Camera(fea->_mappoint->_pos_world, current->_TCW)[2];
    }
}

void OptimizeCurrentPoseOnlyBatch(const vector<Frame *> &frames)
{   // BA.cpp:188-264 for every frame of the batch in one launch
    g_po_trace.start();
    std::vector<int32_t> off(1, 0);
    std::vector<double> px, pw, poses, depth;
    { size_t n0 = 0; for (Frame *f : frames) n0 += f->_features.size(); px.reserve(2 * n0); pw.reserve(3 * n0); depth.reserve(n0 + 1); poses.reserve(6 * frames.size()); }
    for (Frame *f : frames) {
        for (Feature *fea : f->_features) {
            assert(fea->_mappoint != nullptr);
            px.push_back(fea->_pixel[0]); px.push_back(fea->_pixel[1]);
            for (int i = 0; i < 3; ++i) pw.push_back(fea->_mappoint->_pos_world[i]);
            depth.push_back(fea->_depth);
        }
        off.push_back((int32_t)(px.size() / 2));
        const Vector3d t = f->_TCW.translation(), r = f->_TCW.so3().log();
        for (int i = 0; i < 3; ++i) poses.push_back(t[i]);
        for (int i = 0; i < 3; ++i) poses.push_back(r[i]);
    }
    std::vector<uint8_t> bad(std::max<size_t>(depth.size(), 1));
    if (depth.empty()) depth.push_back(0);
    g_po_trace.lap(0);
    if (!hip::check(ygz_hip_optimize_pose_only(hip::Runtime::Get().ctx(), (int)frames.size(), off.data(), px.data(), pw.data(), poses.data(),
                                               bad.data(), depth.data(), nullptr, nullptr), "optimize_pose_only")) return;
    g_po_trace.lap(1);
    size_t g = 0;
    for (size_t fi = 0; fi < frames.size(); ++fi) {
        Frame *f = frames[fi];
        const double *p = &poses[6 * fi];
        f->_TCW = SE3(SO3::exp(Vector3d(p[3], p[4], p[5])), Vector3d(p[0], p[1], p[2]));
        for (Feature *fea : f->_features) {
            fea->_bad = bad[g] != 0;
            fea->_depth = depth[g];               // the depth of the LAST round in which the feature was an inlier (BA.cpp:236-240); never an inlier: unchanged
            if (fea->_bad == false && fea->_mappoint && fea->_mappoint->_bad == false) fea->_mappoint->_cnt_found++;   // BA.cpp:257-263
            ++g;
        }
    }
    g_po_trace.lap(2); g_po_trace.done();
}
void OptimizeCurrentPoseOnly(Frame *current) { OptimizeCurrentPoseOnlyBatch(vector<Frame *>(1, current)); }

void OptimizeCurrentPointOnly(Frame *current)
{   // BA.cpp:266-322: the frame's map points against the (constant) current pose and every observing keyframe
    CeresArrays A;
    const int kc = A.add_pose(current->_TCW, true);
    std::map<unsigned long, int> kf_index;
    std::map<MapPoint *, int> pt_index;
    for (Feature *fea : current->_features) {
        if (fea->_bad || fea->_mappoint == nullptr) continue;
        auto it = pt_index.find(fea->_mappoint);
        const int il = it != pt_index.end() ? it->second : (pt_index[fea->_mappoint] = A.add_point(fea->_mappoint->_pos_world, false));
        A.add_edge(kc, il, Frame::_camera->Pixel2Camera2D(fea->_pixel), 0.0);
        for (auto &obs_pair : fea->_mappoint->_obs) {
            Frame *frame = Memory::GetKeyFrame(obs_pair.first);
            auto kt = kf_index.find(obs_pair.first);
            const int k = kt != kf_index.end() ? kt->second : (kf_index[obs_pair.first] = A.add_pose(frame->_TCW, true));
            A.add_edge(k, il, Frame::_camera->Pixel2Camera2D(obs_pair.second->_pixel), 0.0);
        }
    }
    if (!A.solve()) return;
    for (auto &pi : pt_index) pi.first->_pos_world = Vector3d(A.points[3 * pi.second], A.points[3 * pi.second + 1], A.points[3 * pi.second + 2]);
}

void LocalBA(std::set<Frame *> &local_keyframes, std::set<MapPoint *> &local_map_points)
{   // BA.cpp:324-384: keyframe 0 enters through the PointOnly functor (constant), no loss function, default options
    CeresArrays A;
    std::map<Frame *, int> pose_index;
    std::vector<MapPoint *> order;                      // owner of each point block
    for (MapPoint *mp : local_map_points) {
        int il = -1;
        for (auto &obs_pair : mp->_obs) {
            Frame *frame = Memory::GetKeyFrame(obs_pair.first);
            assert(frame != nullptr);
            if (local_keyframes.find(frame) == local_keyframes.end()) continue;
            if (il < 0) { il = A.add_point(mp->_pos_world, false); order.push_back(mp); }
            auto it = pose_index.find(frame);
            const int k = it != pose_index.end() ? it->second : (pose_index[frame] = A.add_pose(frame->_TCW, frame->_keyframe_id == 0));
            A.add_edge(k, il, Frame::_camera->Pixel2Camera2D(obs_pair.second->_pixel), 0.0);
        }
    }
    if (A.edge_pose.empty()) return;
    A.solve();
    for (auto &pp : pose_index) if (pp.first->_keyframe_id != 0) pp.first->_TCW = A.pose(pp.second);      // BA.cpp:378-382
    for (size_t l = 0; l < order.size(); ++l) order[l]->_pos_world = Vector3d(A.points[3 * l], A.points[3 * l + 1], A.points[3 * l + 2]);
}
}  // namespace ba
}  // namespace ygz
