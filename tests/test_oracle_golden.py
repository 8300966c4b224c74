"""CPU tests: the oracle against the committed golden vectors + size-independent properties.

The reference has no asserting tests or golden vectors; the known-answer content
that does exist -- the 8 poses / 16 points of test/test_local_ba.cpp:9-37 -- is checked here
(zero-noise reprojection residuals must vanish, Jacobians must match finite differences)."""
import os
import zlib
import numpy as np
import pytest
import fixtures
from conftest import golden, ROOT
from ygz_slam_amd import synth


def test_orb_pattern_table_matches_reference_when_present():
    import re
    hdr = open(os.path.join(ROOT, "include", "ygz_orb_pattern.h")).read()
    body = hdr[hdr.index("YGZ_ORB_PATTERN_VALUES") + len("YGZ_ORB_PATTERN_VALUES"):hdr.index("static const")]
    vals = [int(t) for t in re.findall(r"-?\d+", body)]
    assert len(vals) == 1024
    crc = zlib.crc32(bytes(v & 0xFF for v in vals))
    assert "0x%08x" % crc in hdr
    ref = "/root/reference/src/Algorithm/FeatureDetector.cpp"
    if os.path.exists(ref):          # build container only; the GPU box has no reference tree
        src = open(ref).read()
        b = src[src.index("bit_pattern_31_[256*4]"):]
        b = re.sub(r"/\*.*?\*/", "", b[b.index("{") + 1:b.index("};")], flags=re.S)
        assert [int(t) for t in re.findall(r"-?\d+", b)] == vals


def test_image_golden(oracle):
    g = golden("image")
    assert np.array_equal(oracle.bgr2gray(g["bgr"]), g["gray"])
    assert np.array_equal(oracle.pyr_down(g["img"]), g["l1"])
    assert np.array_equal(oracle.pyr_down(g["l1"]), g["l2"])


def test_pyrdown_properties(oracle):
    c = np.full((33, 47), 117, np.uint8)
    assert np.all(oracle.pyr_down(c) == 117)            # kernel sums to 256
    assert oracle.pyr_down(c).shape == (17, 24)         # ((h+1)/2, (w+1)/2)
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (40, 64), dtype=np.uint8)
    d = oracle.pyr_down(a)
    # interior pixel against a float64 5x5 binomial evaluation
    k = np.array([1, 4, 6, 4, 1.0])
    y, x = 7, 11
    ref = (np.outer(k, k) * a[2 * y - 2:2 * y + 3, 2 * x - 2:2 * x + 3]).sum()
    assert d[y, x] == int(ref + 128) >> 8
    # BGR->gray of a gray image is the identity (coefficients sum to 1<<14)
    g = np.stack([a, a, a], -1)
    assert np.array_equal(oracle.bgr2gray(g), a)


def test_fast_golden_and_score_closed_form(oracle):
    g = golden("extract")
    lv = oracle.pyramid(g["imgs"][0], 3)
    for L in range(3):
        xy = oracle.fast_detect(lv[L], 15)
        assert np.array_equal(xy, g["xy%d" % L])
        sc = oracle.fast_score(lv[L], xy, 15)
        assert np.array_equal(sc, g["sc%d" % L])
        assert sc.min() >= 15 and sc.max() <= 254
        cf = np.array([oracle.fast_score_closed_form(lv[L], int(x), int(y)) for x, y in xy[:400]])
        assert np.array_equal(cf, sc[:400])              # bisection == closed form
        assert np.array_equal(oracle.fast_nonmax(xy, sc, 0), g["nm%d" % L])
        assert np.array_equal(oracle.fast_nonmax(xy, sc, 1), g["nmtie%d" % L])
        # raster order and border
        key = xy[:, 1].astype(np.int64) * 10000 + xy[:, 0]
        assert np.all(np.diff(key) > 0)
        assert xy[:, 0].min() >= 3 and xy[:, 1].min() >= 3


def test_fast_synthetic_corner(oracle):
    img = np.full((32, 32), 50, np.uint8)
    img[16:, 16:] = 200                                   # an L corner: pixel (16,16) is bright, 3/4 of the ring dark
    xy = oracle.fast_detect(img, 15)
    assert len(xy) > 0
    assert any((x, y) == (16, 16) for x, y in xy)
    flat = np.full((32, 32), 77, np.uint8)
    assert len(oracle.fast_detect(flat, 15)) == 0


def test_nonmax_variants(oracle):
    xy = np.array([[5, 5], [6, 5], [5, 6]], np.int16)
    sc = np.array([20, 20, 19], np.int32)
    assert list(oracle.fast_nonmax(xy, sc, 0)) == [0, 1]   # ties survive when only strictly greater suppresses
    assert list(oracle.fast_nonmax(xy, sc, 1)) == []       # >= variant kills both tied corners (and the weaker one)


def _kp_equal(a, b):
    assert len(a) == len(b)
    for f in ("px", "py", "level"):
        assert np.array_equal(a[f], b[f]), f
    assert np.array_equal(np.isnan(a["score"]), np.isnan(b["score"]))
    m = ~np.isnan(a["score"])
    assert np.array_equal(a["score"][m], b["score"][m])
    assert np.array_equal(a["angle"], b["angle"])
    assert np.array_equal(a["desc"], b["desc"])


def test_detect_golden(oracle):
    g = golden("extract")
    prm = oracle.default_params(320, 240, 3)
    k0 = oracle.detect(oracle.pyramid(g["imgs"][0], 3), prm)
    _kp_equal(k0, g["k0"])
    _kp_equal(oracle.detect(oracle.pyramid(g["imgs"][1], 3), prm), g["k1"])
    k0occ = oracle.detect(oracle.pyramid(g["imgs"][0], 3), prm, g["occ"])
    _kp_equal(k0occ, g["k0occ"])
    # one feature per cell, cell order, occupied cells empty, pixel = level pixel * 2^level
    cell = (k0["py"].astype(int) // 10) * 32 + k0["px"].astype(int) // 10
    assert np.all(np.diff(cell) > 0)
    cocc = (k0occ["py"].astype(int) // 10) * 32 + k0occ["px"].astype(int) // 10
    assert not np.any(g["occ"][cocc])
    assert np.all(k0["px"] % (1 << k0["level"]) == 0)
    assert np.all((k0["angle"] >= 0) & (k0["angle"] < 360))
    # describe() on the detected pixels reproduces angle+descriptor (ComputeAngleAndDescriptor)
    again = oracle.describe(oracle.pyramid(g["imgs"][0], 3), k0)
    assert np.array_equal(again["desc"], k0["desc"]) and np.array_equal(again["angle"], k0["angle"])


def test_fast_atan2(oracle):
    rng = np.random.default_rng(1)
    for _ in range(500):
        y, x = rng.normal(0, 1000, 2)
        a = oracle.fast_atan2(y, x)
        t = np.degrees(np.arctan2(y, x)) % 360
        assert min(abs(a - t), 360 - abs(a - t)) < 0.05
    assert oracle.fast_atan2(0, 0) == 0.0
    assert oracle.fast_atan2(0, -5) == 180.0


def test_descriptor_distance_and_matcher_golden(oracle):
    g = golden("hamming")
    q, t = g["q"], g["t"]
    lut = np.array([bin(i).count("1") for i in range(256)])
    for i, j in [(0, 0), (3, 10), (20, 5), (69, 52)]:
        assert oracle.descriptor_distance(q[i], t[j]) == lut[q[i] ^ t[j]].sum()
    assert oracle.descriptor_distance(q[3], t[10]) == 0
    D = lut[q[:, None, :] ^ t[None, :, :]].sum(-1)
    idx, d, d2 = oracle.hamming_nn(q, t)
    assert np.array_equal(idx, D.argmin(1)) and np.array_equal(d, D.min(1))     # argmin = first minimum
    assert idx[3] == 10                                                         # tie (t[10]==t[11]) -> lower index
    srt = np.sort(D, 1)
    assert np.array_equal(d2, srt[:, 1])
    for cc in (0, 1, 2):
        i_, d_, n_ = oracle.bf_match(q, t, cc)
        assert np.array_equal(i_, g["idx%d" % cc]) and np.array_equal(d_, g["dist%d" % cc])
    # numpy restatement of the OpenCV cross-check and of mutual NN
    tq = D.argmin(0)
    exp = np.full(len(q), -1); expd = np.full(len(q), 2 ** 31 - 1)
    for j in range(len(t)):
        if D[tq[j], j] < expd[tq[j]]:
            expd[tq[j]] = D[tq[j], j]; exp[tq[j]] = j
    assert np.array_equal(g["idx1"], exp)
    mutual = np.where(tq[D.argmin(1)] == np.arange(len(q)), D.argmin(1), -1)
    assert np.array_equal(g["idx2"], mutual)
    assert set(np.nonzero(g["idx2"] >= 0)[0]) <= set(np.nonzero(g["idx1"] >= 0)[0])   # mutual NN is a subset
    # empty sets
    e = np.zeros((0, 32), np.uint8)
    i_, d_, n_ = oracle.bf_match(q, e, 1)
    assert n_ == 0 and np.all(i_ == -1)
    i_, d_, n_ = oracle.bf_match(e, t, 1)
    assert n_ == 0 and len(i_) == 0


def test_se3_properties(oracle):
    rng = np.random.default_rng(3)
    for _ in range(50):
        v = rng.normal(0, 0.5, 6)
        T = oracle.se3_exp(v)
        assert np.allclose(oracle.se3_log(T), v, atol=1e-12)
        I = oracle.se3_mul(T, oracle.se3_inv(T))
        assert np.allclose(I, [0, 0, 0, 1, 0, 0, 0], atol=1e-12)
        p = rng.normal(0, 2, 3)
        R = synth.quat_to_R(T[:4])
        assert np.allclose(oracle.se3_act(T, p), R @ p + T[4:], atol=1e-12)
    assert np.allclose(oracle.se3_exp(np.zeros(6)), [0, 0, 0, 1, 0, 0, 0])
    assert np.allclose(oracle.se3_exp(np.array([1, 2, 3, 0, 0, 0.0])), [0, 0, 0, 1, 1, 2, 3])


def test_local_ba_known_answer(oracle):
    """test/test_local_ba.cpp:9-37 without noise: points[i] seen through keyframe_poses[j] must
    reproject exactly on the observation (residual 0) and a zero gradient."""
    g = golden("ba_exact")
    r = oracle.ba_linearize(g["poses"], g["fixed"], g["points"], g["edge_pose"], g["edge_point"], g["obs"])
    assert np.abs(r["err"]).max() < 1e-10
    assert abs(r["chi2"]) < 1e-18
    assert np.abs(r["bp"]).max() < 1e-7 and np.abs(r["bl"]).max() < 1e-7
    assert np.all(r["Hpp"][0] == 0)                       # keyframe 0 is fixed (BA.cpp:404-405)
    for k in r:
        assert np.array_equal(np.asarray(r[k]), g["o_" + k]), k
    # the projection itself against an independent numpy pinhole model
    for e in range(0, 128, 17):
        j, i = g["edge_pose"][e], g["edge_point"][e]
        om, t = fixtures.TEST_LOCAL_BA_POSES[j]
        T = np.concatenate([synth.se3_exp([0, 0, 0, *om])[:4], t])
        uv, _ = synth.project(T, np.array([fixtures.TEST_LOCAL_BA_POINTS[i]], float))
        assert np.allclose(uv[0], g["obs"][e], atol=1e-9)


def test_local_ba_noisy_golden_and_jacobians(oracle):
    g = golden("ba_noisy")
    r = oracle.ba_linearize(g["poses"], g["fixed"], g["points"], g["edge_pose"], g["edge_point"], g["obs"])
    for k in r:
        assert np.array_equal(np.asarray(r[k]), g["o_" + k]), k
    # Huber: edge 5 was pushed beyond delta -> rho' < 1 there
    e2 = r["chi2_edge"]
    assert e2[5] > 5.991 ** 2
    assert r["chi2"] < e2.sum()
    # finite differences of the residual (left perturbation exp(d) T for the pose, additive for the point)
    pose, pt, ob = g["poses"][3], g["points"][5], g["obs"][0]
    e0, Jp, Jx = oracle.ba_edge(pose, pt, ob)
    h = 1e-6
    for k in range(3):
        d = pt.copy(); d[k] += h
        assert np.allclose((oracle.ba_edge(pose, d, ob)[0] - e0) / h, Jp[:, k], rtol=1e-4, atol=1e-4)
    for k in range(6):
        u = np.zeros(6); u[k] = h
        assert np.allclose((oracle.ba_edge(oracle.ba_pose_oplus(pose, u), pt, ob)[0] - e0) / h, Jx[:, k], rtol=1e-4, atol=1e-3)
    # blocks against a dense numpy J^T W J assembled from the per-edge Jacobians
    K, P = len(g["poses"]), len(g["points"])
    Hll = np.zeros((P, 3, 3)); Hpp = np.zeros((K, 6, 6))
    for e in range(len(g["obs"])):
        ip, il = g["edge_pose"][e], g["edge_point"][e]
        er, Jp, Jx = oracle.ba_edge(g["poses"][ip], g["points"][il], g["obs"][e])
        w = 1.0 if er @ er <= 5.991 ** 2 else 5.991 / np.sqrt(er @ er)
        Hll[il] += w * Jp.T @ Jp
        if not g["fixed"][ip]:
            Hpp[ip] += w * Jx.T @ Jx
            assert np.allclose(r["Hpl"][e], w * Jx.T @ Jp, rtol=1e-12, atol=1e-9)
    assert np.allclose(Hll, r["Hll"], rtol=1e-12, atol=1e-9) and np.allclose(Hpp, r["Hpp"], rtol=1e-12, atol=1e-6)


def test_normalised_plane_edge(oracle):
    """legacy include/ygz/g2o_types.h edge == pixel edge divided by the focal lengths, columns permuted"""
    rng = np.random.default_rng(5)
    pose_rt = rng.normal(0, 0.1, 6)                        # [omega; t]
    pt = np.array([0.3, -0.2, 3.0])
    cam = oracle.camera()
    ob = np.array([300.0, 200.0])
    e, Jp, Jx = oracle.ba_edge(pose_rt, pt, ob, cam)
    pose_tr = np.concatenate([pose_rt[3:], pose_rt[:3]])
    obn = np.array([(ob[0] - cam.cx) / cam.fx, (ob[1] - cam.cy) / cam.fy])
    en, Jpn, Jxn = oracle.ba_edge_norm(pose_tr, pt, obn)
    f = np.array([cam.fx, cam.fy], float)
    assert np.allclose(e / f, en, atol=1e-12)
    assert np.allclose(Jp / f[:, None], Jpn, atol=1e-12)
    assert np.allclose(Jx[:, [3, 4, 5, 0, 1, 2]] / f[:, None], Jxn, atol=1e-12)


def test_align_golden(oracle):
    g = golden("align")
    e = golden("extract")
    lv0, lv1 = oracle.pyramid(e["imgs"][0], 3), oracle.pyramid(e["imgs"][1], 3)
    T_ref, T_cur = e["poses"][0], e["poses"][1]
    n = len(g["depth"])
    for i in range(0, n, 3):
        ok, px, sl = oracle.find_direct_projection(lv0, T_ref, lv1, T_cur, g["px_ref"][i], g["depth"][i], int(g["level"][i]), g["pred"][i])
        assert ok == g["fdp_ok"][i] and sl == g["fdp_sl"][i]
        assert np.array_equal(px, g["fdp_px"][i], equal_nan=True)
    ok_n, px_n, sl_n = oracle.find_direct_projection_n(lv0, T_ref, lv1, T_cur, g["px_ref"], g["depth"], g["level"], g["pred"])
    assert np.array_equal(ok_n, g["fdp_ok"].astype(bool)) and np.array_equal(px_n, g["fdp_px"], equal_nan=True)       # the batched form (bench CPU leg)
    assert g["fdp_ok"].mean() > 0.5                        # the synthetic pair is trackable
    nm, T, st = oracle.sparse_align(lv0, T_ref, lv1, g["T_init"], g["px_ref"], g["depth"], g["has_mp"])
    assert nm == int(g["sa_nmeas"]) and np.array_equal(T, g["sa_T"])
    # alignment must move the perturbed pose towards the true one
    err0 = np.linalg.norm(oracle.se3_log(oracle.se3_mul(g["T_init"], oracle.se3_inv(T_cur))))
    err1 = np.linalg.norm(oracle.se3_log(oracle.se3_mul(T, oracle.se3_inv(T_cur))))
    assert err1 < 0.5 * err0
    # method_ = LevenbergMarquardt (NLSSolver_impl.hpp:91-212), reproduced as written -- quirks included: stop_ is set by the five failed trials that
    # end the coarsest level and never reset (reset() runs once, in run(): SparseImageAlign.cpp:23), and the first computeResiduals of a finer level
    # divides by the measurements of BOTH levels (n_meas_ is not cleared in front of NLSSolver_impl.hpp:101), so the finer levels end after one failed
    # trial each: the reference's LM aligns on the coarsest level only (which is presumably why its only caller asks for Gauss-Newton, Matcher.cpp:18)
    nm_lm, T_lm, st_lm = oracle.sparse_align(lv0, T_ref, lv1, g["T_init"], g["px_ref"], g["depth"], g["has_mp"], method="lm")
    err_lm = np.linalg.norm(oracle.se3_log(oracle.se3_mul(T_lm, oracle.se3_inv(T_cur))))
    assert nm_lm == nm and err1 < err_lm < err0
    assert list(st_lm.iters_per_level)[:3] == [0, 0, 4] and st_lm.n_iter_total == 11
    kp, kst, kerr = oracle.klt_track(e["imgs"][0], e["imgs"][1], g["px_ref"].astype(np.float32), g["klt_init"])
    assert np.array_equal(kp, g["klt_pts"]) and np.array_equal(kst, g["klt_status"]) and np.array_equal(kerr, g["klt_err"])


def test_track_local_map_structure(oracle):
    """yo_track_local_map (LocalMapping.cpp:47-120) = frustum filter + per-candidate MapPoint-overload FindDirectProjection with
    the first success per point; checked against single calls, and the MapPoint overload against the Feature overload."""
    g = golden("align")
    e = golden("extract")
    lv0, lv1 = oracle.pyramid(e["imgs"][0], 3), oracle.pyramid(e["imgs"][1], 3)
    T_ref, T_cur = e["poses"][0], e["poses"][1]
    n = 120
    px_ref, depth, level = g["px_ref"][:n], g["depth"][:n], g["level"][:n].astype(np.int32)
    keep = depth > 0
    px_ref, depth, level = px_ref[keep], depth[keep], level[keep]
    n = len(depth)
    Twc = oracle.se3_inv(T_ref)
    pc = np.stack([(px_ref[:, 0] - synth.CX) / synth.FX * depth, (px_ref[:, 1] - synth.CY) / synth.FY * depth, depth], 1)
    pos = pc @ synth.quat_to_R(Twc[:4]).T + Twc[4:]
    bad = np.zeros(n, np.uint8); bad[1] = 1
    # two candidates per point: a deliberately wrong observation first (pixel far from the true one), then the right one
    cp = np.repeat(np.arange(n), 2).astype(np.int32)
    cx = np.repeat(px_ref, 2, axis=0); cx[0::2] += [37.0, -23.0]
    cl = np.repeat(level, 2).astype(np.int32)
    cnt, vis, proj, match, pxm, lvl = oracle.track_local_map([lv0], [T_ref], lv1, T_cur, pos, bad, cp, np.zeros(2 * n, np.int32), cx, cl)
    assert vis[1] == 0 and match[1] == -1 and cnt == (match >= 0).sum() and cnt > 0.4 * n
    second = 0
    for p in range(n):
        if not vis[p]:
            assert match[p] == -1
            continue
        want = -1
        for c in (2 * p, 2 * p + 1):
            ok, px, sl = oracle.find_direct_projection_mp(lv0, T_ref, lv1, T_cur, pos[p], cx[c], int(cl[c]), proj[p])
            if ok:
                want = c
                assert np.array_equal(px, pxm[p]) and sl == lvl[p]
                break
        assert match[p] == want
        second += want == 2 * p + 1
    assert second > 0.3 * n                                  # the wrong observation fails, the right one is taken
    # MapPoint overload == Feature overload given the same depth (z of the point in the reference keyframe)
    for p in range(0, n, 5):
        z = (synth.quat_to_R(np.asarray(T_ref)[:4]) @ pos[p] + np.asarray(T_ref)[4:])[2]
        a = oracle.find_direct_projection_mp(lv0, T_ref, lv1, T_cur, pos[p], px_ref[p], int(level[p]), proj[p])
        b = oracle.find_direct_projection(lv0, T_ref, lv1, T_cur, px_ref[p], z, int(level[p]), proj[p])
        assert a[0] == b[0] and a[2] == b[2] and np.allclose(a[1], b[1], rtol=0, atol=1e-6)


def test_align2d_recovers_known_shift(oracle):
    rng = np.random.default_rng(9)
    yy, xx = np.mgrid[0:80, 0:96]
    img = (128 + 60 * np.sin(xx / 4.0) * np.cos(yy / 5.0) + 30 * np.sin((xx + yy) / 3.0)).astype(np.uint8)
    cx, cy = 40, 37
    pwb = img[cy - 5:cy + 5, cx - 5:cx + 5].copy()          # 10x10 around (cx,cy): patch pixel (5,5) == centre
    ok, u, v, chi2, it = oracle.align2d(img, pwb, pwb[1:9, 1:9].copy(), cx + 1.3, cy - 0.8)
    assert ok and abs(u - cx) < 0.05 and abs(v - cy) < 0.05 and it < 10
    ok, u, v, chi2, it = oracle.align2d(img, pwb, pwb[1:9, 1:9].copy(), 2.0, 2.0)     # too close to the border
    assert not ok and (u, v) == (2.0, 2.0)


def test_klt_recovers_translation(oracle):
    tex, m = synth.make_texture(5, 320, 240, margin=40)
    a, _ = synth.render(tex, m, synth.se3_exp(np.zeros(6)), 320, 240)
    b = np.roll(np.roll(a, 3, axis=1), -2, axis=0)          # +3 px in x, -2 px in y
    pts = np.array([[100.0, 100.0], [160.0, 120.0], [200.0, 80.0], [5.0, 5.0]], np.float32)
    out, st, err = oracle.klt_track(a, b, pts, pts.copy())
    good = st[:3].astype(bool)
    assert good.sum() >= 2
    assert np.allclose((out - pts)[:3][good], [3, -2], atol=0.15)


def test_ldlt6(oracle):
    rng = np.random.default_rng(2)
    A = rng.normal(0, 1, (6, 6)); H = A @ A.T + 0.1 * np.eye(6); b = rng.normal(0, 1, 6)
    ok, x = oracle.ldlt6_solve(H, b)
    assert ok and np.allclose(H @ x, b, atol=1e-10)
    ok, x = oracle.ldlt6_solve(np.zeros((6, 6)), b)
    assert ok and np.all(x == 0)                             # pseudo-inverse of a zero D
