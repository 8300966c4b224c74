// Exercises the ygz:: class surfaces the way the reference's test programs do (test/test_feature_extraction.cpp,
// test_orb_match.cpp, test_LK_tracking.cpp, test_feature_alignment.cpp, test_feature_projection.cpp,
// test_local_ba.cpp), on inputs written by tests/test_gpu_surface.py, and dumps every result as text so the Python
// side can compare it with the oracle.  Usage: test_surface <in_dir> <out_file>
#include "ygz/Basic.h"
#include "ygz/Algorithm.h"
#include "ygz/hip/Runtime.h"
#include <fstream>
#include <cstdio>
using namespace ygz;

static std::vector<uint8_t> slurp(const std::string &p) { std::ifstream f(p, std::ios::binary); return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>()); }
static std::vector<double> slurp_f64(const std::string &p) { auto b = slurp(p); std::vector<double> v(b.size() / 8); memcpy(v.data(), b.data(), v.size() * 8); return v; }

int main(int argc, char **argv)
{
    if (argc >= 2 && std::string(argv[1]) == "config") {
        // `test_surface config [file] -- key ...`: ygz::Config alone (no device): the built-in defaults, or what SetParameterFile reads
        int i = 2;
        if (i < argc && std::string(argv[i]) != "--") { if (!Config::SetParameterFile(argv[i])) return 3; ++i; }
        for (++i; i < argc; ++i) printf("%s %s %.17g\n", argv[i], Config::Raw(argv[i]).c_str(), Config::Get<double>(argv[i]));
        return 0;
    }
    if (argc >= 2 && std::string(argv[1]) == "covis") {
        // `test_surface covis`: the covisibility members of ygz::Frame (src/Basic/Frame.cpp:73-176) on a hand-made map, no device.
        // keyframes 0..3; keyframe 3 shares 20 map points with keyframe 0, 16 with keyframe 1, 5 with keyframe 2 (one of those bad -> 4 counted + ...)
        Memory::Clean();
        std::vector<Frame *> kf;
        for (int i = 0; i < 4; ++i) { Frame *fr = new Frame; Memory::RegisterKeyFrame(fr); kf.push_back(fr); }
        auto share = [&](int a, int b, int n, bool bad) {
            for (int i = 0; i < n; ++i) {
                MapPoint *mp = Memory::CreateMapPoint(); mp->_bad = bad;
                for (int k : { a, b }) { Feature *fe = new Feature(Vector2d(10 + i, 20 + k)); fe->_frame = kf[k]; fe->_mappoint = mp; kf[k]->_features.push_back(fe); mp->_obs[kf[k]->_keyframe_id] = fe; }
            }
        };
        share(3, 0, 20, false); share(3, 1, 16, false); share(3, 2, 5, false); share(3, 2, 7, true); share(2, 0, 3, false);
        { Feature *lone = new Feature(Vector2d(1, 1)); lone->_frame = kf[3]; kf[3]->_features.push_back(lone); }      // no map point
        kf[3]->UpdateConnections();
        printf("kf3 cov");
        for (size_t i = 0; i < kf[3]->_cov_keyframes.size(); ++i) printf(" %lu:%d", kf[3]->_cov_keyframes[i]->_keyframe_id, kf[3]->_cov_weights[i]);
        printf(" | connected");
        for (int i = 0; i < 4; ++i) if (kf[3]->_connected_keyframe_weights.count(kf[i])) printf(" %d:%d", i, kf[3]->_connected_keyframe_weights[kf[i]]);
        printf(" | best1 %zu:%lu best10 %zu\n", kf[3]->GetBestCovisibilityKeyframes(1).size(), kf[3]->GetBestCovisibilityKeyframes(1)[0]->_keyframe_id,
               kf[3]->GetBestCovisibilityKeyframes().size());
        // keyframe 2: nothing reaches the threshold of 15 -> the single best neighbour is kept, and told (AddConnection)
        kf[2]->UpdateConnections();
        printf("kf2 cov");
        for (size_t i = 0; i < kf[2]->_cov_keyframes.size(); ++i) printf(" %lu:%d", kf[2]->_cov_keyframes[i]->_keyframe_id, kf[2]->_cov_weights[i]);
        printf(" | kf3 now sees kf2 with %d | in frustum %d\n", kf[3]->_connected_keyframe_weights[kf[2]], (int)kf[2]->IsInFrustum(nullptr));
        // a second call replaces the sorted lists; UpdateBestCovisibles appends every connection, heaviest first
        kf[3]->UpdateConnections();
        const size_t n_before = kf[3]->_cov_keyframes.size();
        kf[3]->UpdateBestCovisibles();
        printf("kf3 again %zu then", n_before);
        for (size_t i = n_before; i < kf[3]->_cov_keyframes.size(); ++i) printf(" %lu:%d", kf[3]->_cov_keyframes[i]->_keyframe_id, kf[3]->_cov_weights[i]);
        Frame none;
        none.UpdateConnections();
        printf(" | empty %zu %zu\n", none._cov_keyframes.size(), none.GetBestCovisibilityKeyframes(5).size());
        return 0;
    }
    if (argc < 3) { fprintf(stderr, "usage: %s in_dir out_file | %s config [file] -- key ... | %s covis\n", argv[0], argv[0], argv[0]); return 2; }
    const std::string in = argv[1];
    FILE *out = fopen(argv[2], "w");
    Config::SetParameterFile(in + "/default.yaml");
    PinholeCamera *cam = new PinholeCamera();
    Frame::SetCamera(cam);
    const int W = Config::Get<int>("image.width"), H = Config::Get<int>("image.height");
    auto poses = slurp_f64(in + "/poses.f64");           // [2][7]
    auto depth0 = slurp_f64(in + "/depth0.f64");         // [H][W]
    Frame f[2];
    std::vector<uint8_t> raw[2] = { slurp(in + "/frame0.bgr"), slurp(in + "/frame1.bgr") };
    for (int i = 0; i < 2; ++i) {
        f[i]._color = cv::Mat(H, W, CV_8UC3, raw[i].data());
        f[i].InitFrame();
        f[i]._TCW = SE3::from7(&poses[7 * i]);
        fprintf(out, "pyr %d %zu %d %d %d\n", i, f[i]._pyramid.size(), f[i]._pyramid[2].cols, f[i]._pyramid[2].rows, (int)f[i]._pyramid[2].at<uchar>(5, 7));
    }
    // --- test_feature_extraction / test_orb_match
    FeatureDetector detector;
    detector.LoadParams();
    for (int i = 0; i < 2; ++i) {
        detector.Detect(&f[i]);
        fprintf(out, "kp %d %zu\n", i, f[i]._features.size());
        for (Feature *fe : f[i]._features) {
            fprintf(out, "%.17g %.17g %d %.9g %.9g", fe->_pixel[0], fe->_pixel[1], fe->_level, fe->_score, fe->_angle);
            for (int k = 0; k < 32; ++k) fprintf(out, " %d", (int)fe->_desc.data[k]);
            fprintf(out, "\n");
        }
    }
    // --- FeatureDetector::ComputeAngleAndDescriptor(Frame*) and ComputeDescriptor(Feature*) (FeatureDetector.cpp:580-594)
    {
        const size_t n = f[0]._features.size();
        std::vector<double> ang(n); std::vector<std::array<uint8_t, 32>> desc(n);
        for (size_t i = 0; i < n; ++i) { Feature *fe = f[0]._features[i]; ang[i] = fe->_angle; memcpy(desc[i].data(), fe->_desc.data, 32); fe->_angle = -1; memset(fe->_desc.data, 0, 32); }
        detector.ComputeAngleAndDescriptor(&f[0]);
        size_t same = 0;
        for (size_t i = 0; i < n; ++i) { const Feature *fe = f[0]._features[i]; same += fe->_angle == ang[i] && memcmp(fe->_desc.data, desc[i].data(), 32) == 0; }
        fprintf(out, "cad %zu %zu\n", same, n);
        for (size_t i = 0; i < 40 && i < n; ++i) {              // the caller's own angle: the descriptor follows it, the angle stays
            Feature *fe = f[0]._features[i];
            fe->_angle = fmod(ang[i] + 33.25 * (double)(i + 1), 360.0);
            detector.ComputeDescriptor(fe);
            fprintf(out, "cd %zu %.17g", i, fe->_angle);
            for (int k = 0; k < 32; ++k) fprintf(out, " %d", (int)fe->_desc.data[k]);
            fprintf(out, "\n");
            fe->_angle = ang[i]; memcpy(fe->_desc.data, desc[i].data(), 32);
        }
    }
    Matcher matcher;
    std::vector<DMatch> matches;
    matcher.BruteForceMatch(&f[0], &f[1], matches, true);
    fprintf(out, "matches %zu\n", matches.size());
    for (auto &m : matches) fprintf(out, "%d %d %d\n", m.queryIdx, m.trainIdx, (int)m.distance);
    fprintf(out, "ddist %d\n", Matcher::DescriptorDistance(f[0]._features[0]->_desc, f[1]._features[0]->_desc));
    // --- test_orb_match / test_match_for_triangulation: vocabulary, Frame::ComputeBoW, SearchByBoW, SearchForTriangulation
    {
        ORBVocabulary vocab;
        const bool okv = vocab.loadFromBinaryFile(in + "/vocab.bin");
        Frame::SetORBVocabulary(&vocab);
        f[0].ComputeBoW(); f[1].ComputeBoW();
        fprintf(out, "bow %d %d %d %zu %zu %zu %zu\n", (int)okv, vocab.k_, vocab.L_, f[0]._bow_vec.size(), f[0]._feature_vec.size(), f[1]._bow_vec.size(), f[1]._feature_vec.size());
        double s0 = 0; for (auto &kv : f[0]._bow_vec) s0 += kv.second;
        fprintf(out, "bow_sum %.17g %u %.17g\n", s0, f[0]._bow_vec.begin()->first, f[0]._bow_vec.begin()->second);
        map<int, int> bm;
        const int c_quirk = matcher.SearchByBoW(&f[0], &f[1], bm);      // knnRatio read through Get<int> (Matcher.cpp:17) = 0: nothing can pass
        matcher._options.knnRatio = 0.7f; bm.clear();
        const int c = matcher.SearchByBoW(&f[0], &f[1], bm);
        fprintf(out, "sbow %d %d %zu\n", c_quirk, c, bm.size());
        {   // Matcher::Options::checkOrientation (Matcher.h:24): the same matches, the count after the rotation histogram
            map<int, int> bo;
            matcher._options.checkOrientation = true;
            const int co = matcher.SearchByBoW(&f[0], &f[1], bo);
            matcher._options.checkOrientation = false;
            fprintf(out, "sbow_o %d %d\n", co, (int)(bo == bm));
        }
        for (auto &kv : bm) fprintf(out, "sbow_m %d %d\n", kv.first, kv.second);
        // E12 of the true relative pose: x2 = R x1 + t, line in frame 2 = pt1^T E12 with E12 = ([t]x R)^T
        const SE3 T21 = f[1]._TCW * f[0]._TCW.inverse();
        const Matrix3d R = T21.rotation_matrix(); const Vector3d t = T21.translation();
        Matrix3d tx; tx(0, 1) = -t[2]; tx(0, 2) = t[1]; tx(1, 0) = t[2]; tx(1, 2) = -t[0]; tx(2, 0) = -t[1]; tx(2, 1) = t[0];
        Matrix3d E12;
        for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) { double v = 0; for (int k = 0; k < 3; ++k) v += tx(cc, k) * R(k, r); E12(r, cc) = v; }
        vector<pair<int, int>> tri;
        const int ct = matcher.SearchForTriangulation(&f[0], &f[1], E12, tri);
        fprintf(out, "stri %d %zu %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", ct, tri.size(), E12(0, 0), E12(0, 1), E12(0, 2), E12(1, 0), E12(1, 1), E12(1, 2), E12(2, 0), E12(2, 1), E12(2, 2));
        for (auto &pr : tri) fprintf(out, "stri_m %d %d\n", pr.first, pr.second);
        // the call LocalMapping::CreateNewMapPoints makes once per matched pair (src/Module/LocalMapping.cpp:405-447): Matcher::FindDirectProjection,
        // Feature overload, with the triangulated depth in fea1->_depth and fea2's pixel as prediction -- answered from ONE launch over the pairs of the
        // SearchForTriangulation above, against the same call as its own n = 1 launch
        {
            hip::ResetFdpMemoStats();
            const SE3 T12 = f[0]._TCW * f[1]._TCW.inverse();
            int n_calls = 0, n_diff = 0, n_ok = 0;
            for (int im = 0; im < ct; ++im) {
                Feature *fea1 = f[0]._features[tri[im].first], *fea2 = f[1]._features[tri[im].second];
                if (fea1->_mappoint || fea2->_mappoint) continue;
                const Vector3d pt1 = cam->Pixel2Camera(fea1->_pixel), pt2 = cam->Pixel2Camera(fea2->_pixel);
                if (pt1.dot(pt2) / (pt1.norm() * pt2.norm()) >= 0.9998) continue;
                double depth1 = 0, depth2 = 0;
                if (!cvutils::DepthFromTriangulation(T12.inverse(), pt1, pt2, depth1, depth2) || depth1 < 0 || depth2 < 0) continue;
                fea1->_depth = depth1;
                Vector2d p1 = fea2->_pixel, p2 = fea2->_pixel; int l1 = 0, l2 = 0;
                const bool o1 = matcher.FindDirectProjection(&f[0], &f[1], fea1, p1, l1);
                hip::SetFdpBypass(true);
                const bool o2 = matcher.FindDirectProjection(&f[0], &f[1], fea1, p2, l2);
                hip::SetFdpBypass(false);
                ++n_calls; n_ok += o1;
                n_diff += o1 != o2 || l1 != l2 || memcmp(p1.data(), p2.data(), 16) != 0;
                fea1->_depth = -1;
            }
            const hip::FdpMemoStats st = hip::GetFdpMemoStats();
            fprintf(out, "fdpfeat %d %d %d %llu %llu %llu %llu\n", n_calls, n_diff, n_ok, st.hits, st.single, st.launches, st.speculated);
        }
        Frame::SetORBVocabulary(nullptr);
    }
    // Detect(frame, false) keeps the old features and fills only free cells
    {
        const size_t before = f[1]._features.size();
        for (size_t k = 0; k < before; k += 2) { delete f[1]._features[k]; f[1]._features[k] = nullptr; }
        f[1]._features.erase(std::remove(f[1]._features.begin(), f[1]._features.end(), (Feature *)nullptr), f[1]._features.end());
        const size_t kept = f[1]._features.size();
        detector.Detect(&f[1], false);
        fprintf(out, "redetect %zu %zu %zu\n", before, kept, f[1]._features.size());
    }
    // --- test_LK_tracking
    Tracker tracker;
    tracker.SetReference(&f[0]);
    tracker.Track(&f[1]);
    std::vector<Feature *> tf; std::vector<Vector2d> tp;
    tracker.GetTrackedPixel(tf, tp);
    fprintf(out, "klt %zu %d %.9g\n", tp.size(), (int)tracker.Status(), tracker.MeanDisparity());
    for (size_t i = 0; i < tp.size(); ++i) fprintf(out, "%.17g %.17g %.9g %.9g\n", tf[i]->_pixel[0], tf[i]->_pixel[1], tp[i][0], tp[i][1]);
    // --- depth + map points for the direct methods
    std::vector<MapPoint *> mps;
    for (size_t i = 0; i < f[0]._features.size(); ++i) {
        Feature *fe = f[0]._features[i];
        fe->_depth = depth0[(size_t)fe->_pixel[1] * W + (size_t)fe->_pixel[0]];
        if (i % 9 != 0) { MapPoint *mp = new MapPoint; mp->_pos_world = cam->Pixel2World(fe->_pixel, f[0]._TCW, fe->_depth); fe->_mappoint = mp; mps.push_back(mp); }
    }
    // --- test_feature_projection: FindDirectProjection per feature (single calls) and batched
    std::vector<Vector2d> px(f[0]._features.size()); std::vector<int> sl; std::vector<bool> ok;
    for (size_t i = 0; i < px.size(); ++i) px[i] = f[0]._features[i]->_pixel + Vector2d(1.5, -1.0);
    matcher.FindDirectProjectionBatch(&f[0], &f[1], f[0]._features, px, sl, ok);
    fprintf(out, "fdp %zu\n", px.size());
    for (size_t i = 0; i < px.size(); ++i) fprintf(out, "%d %d %.17g %.17g\n", (int)ok[i], sl[i], px[i][0], px[i][1]);
    { Vector2d p1 = f[0]._features[3]->_pixel + Vector2d(1.5, -1.0); int l1 = 0;
      bool o1 = matcher.FindDirectProjection(&f[0], &f[1], f[0]._features[3], p1, l1);
      fprintf(out, "fdp1 %d %d %.17g %.17g\n", (int)o1, l1, p1[0], p1[1]); }
    // --- LocalMapping::FindCandidates + ProjectMapPoints through Matcher::ProjectMapPoints
    {
        std::set<Frame *> kfs; kfs.insert(&f[0]);
        std::set<MapPoint *> mpset;
        for (Feature *fe : f[0]._features) if (fe->_mappoint) { fe->_mappoint->_obs[f[0]._keyframe_id] = fe; fe->_frame = &f[0]; mpset.insert(fe->_mappoint); }
        mps[2]->_bad = true;
        const size_t n_before = f[1]._features.size();
        const int nm = matcher.ProjectMapPoints(&f[1], kfs, mpset);
        fprintf(out, "lmap %zu\n", mps.size());
        for (MapPoint *mp : mps) {
            Feature *found = nullptr;
            for (size_t i = n_before; i < f[1]._features.size(); ++i) if (f[1]._features[i]->_mappoint == mp) found = f[1]._features[i];
            fprintf(out, "%d %d %d %.17g %.17g %.17g %.17g %.17g\n", mp->_cnt_visible, found ? 1 : 0, found ? found->_level : 0, found ? found->_pixel[0] : 0.0,
                    found ? found->_pixel[1] : 0.0, mp->_pos_world[0], mp->_pos_world[1], mp->_pos_world[2]);
        }
        fprintf(out, "lmap_n %d %zu\n", nm, f[1]._features.size() - n_before);
        for (size_t i = n_before; i < f[1]._features.size(); ++i) delete f[1]._features[i];
        f[1]._features.resize(n_before);
        mps[2]->_bad = false;
    }
    // --- the call LocalMapping::ProjectMapPoints makes once per candidate (LocalMapping.cpp:98): Matcher::FindDirectProjection, MapPoint overload,
    // answered from ONE speculative launch -- against the same call as its own n = 1 launch, and (Python side) against Matcher::ProjectMapPoints above
    {
        hip::ResetFdpMemoStats();
        for (MapPoint *mp : mps) {
            const Vector3d pc = cam->World2Camera(mp->_pos_world, f[1]._TCW);
            const Vector2d px_in = cam->Camera2Pixel(pc);
            if (pc[2] < 0 || !f[1].InFrame(px_in, 20)) { fprintf(out, "fdpmp -1 0 0 0 -1 0 0 0\n"); continue; }      // FindCandidates drops it (:60-63)
            Vector2d p1 = px_in, p2 = px_in; int l1 = 0, l2 = 0;
            const bool o1 = matcher.FindDirectProjection(&f[0], &f[1], mp, p1, l1);
            hip::SetFdpBypass(true);
            const bool o2 = matcher.FindDirectProjection(&f[0], &f[1], mp, p2, l2);
            hip::SetFdpBypass(false);
            fprintf(out, "fdpmp %d %d %.17g %.17g %d %d %.17g %.17g\n", (int)o1, l1, p1[0], p1[1], (int)o2, l2, p2[0], p2[1]);
        }
        hip::FdpMemoStats st = hip::GetFdpMemoStats();
        fprintf(out, "fdpmp_stats %llu %llu %llu %llu\n", st.hits, st.single, st.launches, st.speculated);
        // inputs the speculation cannot have seen take the n = 1 launch: another prediction, a moved map point; then the keyframe moves (as after a
        // local BA) and the frame is asked about again: a new speculative launch, no stale answer
        MapPoint *mp = mps[5];
        Vector2d pa = cam->World2Pixel(mp->_pos_world, f[1]._TCW) + Vector2d(0.75, -0.5), pb = pa; int la = 0, lb = 0;
        const bool oa = matcher.FindDirectProjection(&f[0], &f[1], mp, pa, la);
        hip::SetFdpBypass(true); const bool ob = matcher.FindDirectProjection(&f[0], &f[1], mp, pb, lb); hip::SetFdpBypass(false);
        fprintf(out, "fdpmp_other %d %d %.17g %.17g %d %d %.17g %.17g\n", (int)oa, la, pa[0], pa[1], (int)ob, lb, pb[0], pb[1]);
        const Vector3d keep = mp->_pos_world;
        mp->_pos_world = keep + Vector3d(0.002, -0.001, 0.003);
        pa = pb = cam->World2Pixel(mp->_pos_world, f[1]._TCW);
        const bool oc = matcher.FindDirectProjection(&f[0], &f[1], mp, pa, la);
        hip::SetFdpBypass(true); const bool od = matcher.FindDirectProjection(&f[0], &f[1], mp, pb, lb); hip::SetFdpBypass(false);
        fprintf(out, "fdpmp_moved %d %d %.17g %.17g %d %d %.17g %.17g\n", (int)oc, la, pa[0], pa[1], (int)od, lb, pb[0], pb[1]);
        mp->_pos_world = keep;
        st = hip::GetFdpMemoStats();
        const SE3 T0 = f[0]._TCW;
        Vector6d d6; d6[0] = 1e-3; d6[4] = 5e-4;
        f[0]._TCW = SE3::exp(d6) * T0;
        pa = pb = cam->World2Pixel(mps[7]->_pos_world, f[1]._TCW);
        const bool oe = matcher.FindDirectProjection(&f[0], &f[1], mps[7], pa, la);
        hip::SetFdpBypass(true); const bool of = matcher.FindDirectProjection(&f[0], &f[1], mps[7], pb, lb); hip::SetFdpBypass(false);
        const hip::FdpMemoStats st2 = hip::GetFdpMemoStats();
        fprintf(out, "fdpmp_kfmoved %d %d %.17g %.17g %d %d %.17g %.17g %llu %llu %llu\n", (int)oe, la, pa[0], pa[1], (int)of, lb, pb[0], pb[1],
                st.single, st2.launches - st.launches, st2.hits - st.hits);
        f[0]._TCW = T0;
    }
    // cvutils::Align2D on a host patch against a pyramid level of frame 1
    { uint8_t pwb[100], patch[64];
      const cv::Mat &img = f[0]._pyramid[0];
      for (int y = 0; y < 10; ++y) for (int x = 0; x < 10; ++x) pwb[y * 10 + x] = img.at<uchar>(200 - 5 + y, 300 - 5 + x);
      for (int y = 0; y < 8; ++y) for (int x = 0; x < 8; ++x) patch[y * 8 + x] = pwb[(y + 1) * 10 + x + 1];
      Vector2d p(301.2, 199.1);
      bool o = cvutils::Align2D(f[0]._pyramid[0], pwb, patch, 10, p);
      fprintf(out, "align2d %d %.17g %.17g\n", (int)o, p[0], p[1]); }
    // --- test_feature_alignment: Matcher::SparseImageAlignment
    const SE3 T1_true = f[1]._TCW;
    bool sa = matcher.SparseImageAlignment(&f[0], &f[1]);
    double T7[7]; f[1]._TCW.to7(T7);
    fprintf(out, "sparse %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", (int)sa, T7[0], T7[1], T7[2], T7[3], T7[4], T7[5], T7[6]);
    fprintf(out, "sparse_err %.9g\n", (f[1]._TCW * T1_true.inverse()).log().norm());
    // ... and SparseImgAlign with method LevenbergMarquardt (NLSSolver_impl.hpp:91-212; nobody in the reference asks for it) from the same start
    {
        const SE3 T_gn = f[1]._TCW;
        f[1]._TCW = f[0]._TCW;
        SparseImgAlign lm(2, 0, 30, SparseImgAlign::LevenbergMarquardt, false, false);
        const size_t nm = lm.run(&f[0], &f[1]);
        double L7[7]; f[1]._TCW.to7(L7);
        fprintf(out, "sparse_lm %zu %d %d %d %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", nm, lm.iterations(2), lm.iterations(1), lm.iterations(0), lm.trials(),
                L7[0], L7[1], L7[2], L7[3], L7[4], L7[5], L7[6]);
        fprintf(out, "sparse_lm_err %.9g %.9g\n", (f[1]._TCW * T1_true.inverse()).log().norm(), (f[1]._TCW * T_gn.inverse()).log().norm());
        f[1]._TCW = T_gn;
    }
    // --- test_local_ba: 8 keyframes x 16 points, noisy state read from the input directory
    {
        auto kp = slurp_f64(in + "/ba_poses7.f64");        // [8][7] noisy T_cw
        auto pt = slurp_f64(in + "/ba_points.f64");        // [16][3] noisy
        auto ob = slurp_f64(in + "/ba_obs.f64");           // [16][8][2]
        Memory::Clean();
        std::set<Frame *> frames; std::set<MapPoint *> map_points;
        std::vector<Frame *> by_id;
        for (int i = 0; i < 8; ++i) { Frame *nf = new Frame(); Memory::RegisterKeyFrame(nf); nf->_TCW = SE3::from7(&kp[7 * i]); frames.insert(nf); by_id.push_back(nf); }
        std::vector<MapPoint *> mpv;
        for (int i = 0; i < 16; ++i) {
            MapPoint *mp = new MapPoint; mp->_id = i; mp->_pos_world = Vector3d(pt[3 * i], pt[3 * i + 1], pt[3 * i + 2]);
            for (int j = 0; j < 8; ++j) { Feature *fea = new Feature(Vector2d(ob[2 * (8 * i + j)], ob[2 * (8 * i + j) + 1])); fea->_frame = by_id[j]; fea->_mappoint = mp; by_id[j]->_features.push_back(fea); mp->_obs[j] = fea; }
            map_points.insert(mp); mpv.push_back(mp);
        }
        ba::LocalBAStats st;
        ba::LocalBAG2O(frames, map_points, &st);
        fprintf(out, "ba %d %d %d %.17g %.17g\n", st.iterations, st.lm_trials, st.outliers, st.chi2_initial, st.chi2_final);
        for (int i = 0; i < 8; ++i) { double t[7]; by_id[i]->_TCW.to7(t); fprintf(out, "%.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", t[0], t[1], t[2], t[3], t[4], t[5], t[6]); }
        for (int i = 0; i < 16; ++i) fprintf(out, "%.17g %.17g %.17g\n", mpv[i]->_pos_world[0], mpv[i]->_pos_world[1], mpv[i]->_pos_world[2]);
    }
    // --- the ceres-based entry points of ba:: on the same fixture (rebuilt from the input files)
    {
        auto kp = slurp_f64(in + "/ba_poses7.f64"); auto pt = slurp_f64(in + "/ba_points.f64"); auto ob = slurp_f64(in + "/ba_obs.f64");
        auto build = [&](std::set<Frame *> &frames, std::set<MapPoint *> &map_points, std::vector<Frame *> &by_id, std::vector<MapPoint *> &mpv) {
            Memory::Clean(); frames.clear(); map_points.clear(); by_id.clear(); mpv.clear();
            for (int i = 0; i < 8; ++i) { Frame *nf = new Frame(); Memory::RegisterKeyFrame(nf); nf->_TCW = SE3::from7(&kp[7 * i]); frames.insert(nf); by_id.push_back(nf); }
            for (int i = 0; i < 16; ++i) {
                MapPoint *mp = new MapPoint; mp->_id = i; mp->_pos_world = Vector3d(pt[3 * i], pt[3 * i + 1], pt[3 * i + 2]);
                for (int j = 0; j < 8; ++j) { Feature *fea = new Feature(Vector2d(ob[2 * (8 * i + j)], ob[2 * (8 * i + j) + 1])); fea->_frame = by_id[j]; fea->_mappoint = mp; by_id[j]->_features.push_back(fea); mp->_obs[j] = fea; }
                map_points.insert(mp); mpv.push_back(mp);
            }
        };
        auto reproj = [&](Frame *fr) { double s = 0; for (Feature *fea : fr->_features) { Vector2d d = Frame::_camera->World2Pixel(fea->_mappoint->_pos_world, fr->_TCW) - fea->_pixel; s += d.dot(d); } return s; };
        std::set<Frame *> frames; std::set<MapPoint *> map_points; std::vector<Frame *> by_id; std::vector<MapPoint *> mpv;
        build(frames, map_points, by_id, mpv);
        ba::LocalBA(frames, map_points);
        fprintf(out, "ba_ceres 0\n");
        for (int i = 0; i < 8; ++i) { double t[7]; by_id[i]->_TCW.to7(t); fprintf(out, "%.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", t[0], t[1], t[2], t[3], t[4], t[5], t[6]); }
        for (int i = 0; i < 16; ++i) fprintf(out, "%.17g %.17g %.17g\n", mpv[i]->_pos_world[0], mpv[i]->_pos_world[1], mpv[i]->_pos_world[2]);
        // OptimizeCurrentPointOnly / OptimizeCurrent on keyframe 5 of the noisy scene: the reprojection error over all views must drop
        build(frames, map_points, by_id, mpv);
        auto reproj_all = [&]() { double s2 = 0; for (Frame *fr : by_id) s2 += reproj(fr); return s2; };     // the cost both calls minimise
        const double e0 = reproj_all();
        ba::OptimizeCurrentPointOnly(by_id[5]);
        const double e1 = reproj_all();
        for (int i = 0; i < 16; ++i) fprintf(out, "ocpo_pt %.17g %.17g %.17g\n", mpv[i]->_pos_world[0], mpv[i]->_pos_world[1], mpv[i]->_pos_world[2]);
        build(frames, map_points, by_id, mpv);
        by_id[5]->_features[2]->_pixel = by_id[5]->_features[2]->_pixel + Vector2d(40.0, -25.0);     // a gross error on one feature (which is also MapPoint 2's observation in keyframe 5)
        ba::OptimizeCurrent(by_id[5]);
        const double e2 = reproj_all();
        int nbad = 0; for (Feature *fea : by_id[5]->_features) nbad += fea->_bad;
        fprintf(out, "opt_current %.17g %.17g %.17g %d\n", e0, e1, e2, nbad);
        { double t[7]; by_id[5]->_TCW.to7(t); fprintf(out, "oc_pose %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", t[0], t[1], t[2], t[3], t[4], t[5], t[6]); }
        for (int i = 0; i < 16; ++i) fprintf(out, "oc_pt %.17g %.17g %.17g %d %.17g\n", mpv[i]->_pos_world[0], mpv[i]->_pos_world[1], mpv[i]->_pos_world[2],
                                             (int)by_id[5]->_features[i]->_bad, by_id[5]->_features[i]->_depth);
        // TwoViewBACeres: keyframes 0 and 7, points 0..15, every correspondence an inlier
        build(frames, map_points, by_id, mpv);
        vector<Vector2d> px_ref, px_curr; vector<bool> inl(16, true); vector<Vector3d> pts;
        for (int i = 0; i < 16; ++i) { px_ref.push_back(mpv[i]->_obs[0]->_pixel); px_curr.push_back(mpv[i]->_obs[7]->_pixel); pts.push_back(mpv[i]->_pos_world); }
        inl[3] = false;
        SE3 curr = by_id[7]->_TCW;
        ba::TwoViewBACeres(by_id[0]->_TCW, curr, px_ref, px_curr, inl, pts);
        double t7[7]; curr.to7(t7); int ninl = 0; for (bool b : inl) ninl += b;
        fprintf(out, "two_view %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", ninl, t7[0], t7[1], t7[2], t7[3], t7[4], t7[5], t7[6]);
        for (int i = 0; i < 16; ++i) fprintf(out, "tv_pt %.17g %.17g %.17g %d\n", pts[i][0], pts[i][1], pts[i][2], (int)inl[i]);
    }
    // --- ba::OptimizeCurrentPoseOnly on a frame written by the harness: pose [t; aa], px [n][2], pw [n][3]
    {
        auto pe = slurp_f64(in + "/po_entry.f64"); auto ppx = slurp_f64(in + "/po_px.f64"); auto ppw = slurp_f64(in + "/po_pw.f64");
        const int n = (int)(ppx.size() / 2);
        Frame fr;
        fr._TCW = SE3(SO3::exp(Vector3d(pe[3], pe[4], pe[5])), Vector3d(pe[0], pe[1], pe[2]));
        std::vector<MapPoint *> mps;
        for (int i = 0; i < n; ++i) {
            MapPoint *mp = new MapPoint; mp->_pos_world = Vector3d(ppw[3 * i], ppw[3 * i + 1], ppw[3 * i + 2]);
            Feature *fea = new Feature(Vector2d(ppx[2 * i], ppx[2 * i + 1])); fea->_frame = &fr; fea->_mappoint = mp; fea->_depth = -1;
            fr._features.push_back(fea); mps.push_back(mp);
        }
        ba::OptimizeCurrentPoseOnly(&fr);
        const Vector3d t = fr._TCW.translation(), r = fr._TCW.so3().log();
        fprintf(out, "pose_only %d %.17g %.17g %.17g %.17g %.17g %.17g\n", n, t[0], t[1], t[2], r[0], r[1], r[2]);
        for (int i = 0; i < n; ++i) fprintf(out, "po_f %d %.17g %d\n", (int)fr._features[i]->_bad, fr._features[i]->_depth, mps[i]->_cnt_found);
    }
    fclose(out);
    return 0;
}
