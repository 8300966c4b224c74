"""GPU parity tests (pytest -m gpu, MI355X): every stage of the hot path through the C ABI
(ygz_slam_amd._lib -> libygz_hip.so) against the CPU oracle and the committed golden vectors.
Integer / byte / index stages: bit-exact.  Float stages: the tolerance is written at the assert
(north_star: LK tracks and BA residuals within 1e-5 relative)."""
import os
import numpy as np
import pytest
import fixtures
from conftest import golden, make_ctx
from ygz_slam_amd import synth

pytestmark = pytest.mark.gpu
LUT = np.array([bin(i).count("1") for i in range(256)])


def _frames(n, w, h, seed=1, step=0.5, identity_first=True):
    tex, m = synth.make_texture(seed, w, h, margin=max(80, w // 4))
    poses = synth.trajectory(n, seed + 10, step)
    if identity_first:
        poses[0] = [0, 0, 0, 1, 0, 0, 0]
    out = [synth.render(tex, m, poses[i], w, h, 1.0, seed * 1000 + i) for i in range(n)]
    return np.stack([o[0] for o in out]), poses, np.stack([o[1] for o in out])


def _corner_maps(oracle, img, thr, tie):
    xy = oracle.fast_detect(img, thr)
    sc = oracle.fast_score(img, xy, thr)
    nm = oracle.fast_nonmax(xy, sc, tie)
    smap = np.zeros(img.shape, np.uint8)
    nmap = np.zeros(img.shape, np.uint8)
    smap[xy[:, 1], xy[:, 0]] = sc + 1
    nmap[xy[nm, 1], xy[nm, 0]] = 1
    return smap, nmap


def _kp_check(kp, ok_):
    assert len(kp["level"]) == len(ok_)
    assert np.array_equal(kp["px"][:, 0], ok_["px"]) and np.array_equal(kp["px"][:, 1], ok_["py"])
    assert np.array_equal(kp["level"], ok_["level"])
    assert np.array_equal(np.isnan(kp["score"]), np.isnan(ok_["score"]))
    m = ~np.isnan(ok_["score"])
    assert np.array_equal(kp["score"][m], ok_["score"][m])
    assert np.array_equal(kp["angle"], ok_["angle"])
    assert np.array_equal(kp["desc"], ok_["desc"])


# ------------------------------------------------------------------------------------- A1
@pytest.mark.parametrize("w,h", [(640, 480), (67, 45), (333, 251)])
def test_gray_and_pyramid_bit_exact(hip_lib, oracle, w, h):
    rng = np.random.default_rng(w * 7 + h)
    bgr = rng.integers(0, 256, (2, h, w, 3), dtype=np.uint8)
    ctx = make_ctx(hip_lib, width=w, height=h, levels=3, max_frames=2)
    for s in range(2):
        ctx.upload_bgr(s, bgr[s])
    ctx.build_pyramid(0, 2, from_bgr=True)
    for s in range(2):
        lv = oracle.pyramid(oracle.bgr2gray(bgr[s]), 3)
        for L in range(3):
            assert np.array_equal(ctx.download_level(s, L), lv[L]), (s, L)
    ctx.close()


def test_pyramid_golden(hip_lib):
    g = golden("image")
    h, w = g["img"].shape
    ctx = make_ctx(hip_lib, width=w, height=h, levels=3, max_frames=1)
    ctx.upload_gray(0, g["img"])
    ctx.build_pyramid(0, 1)
    assert np.array_equal(ctx.download_level(0, 1), g["l1"]) and np.array_equal(ctx.download_level(0, 2), g["l2"])
    hb, wb, _ = g["bgr"].shape
    c2 = make_ctx(hip_lib, width=wb, height=hb, levels=1, max_frames=1)
    c2.upload_bgr(0, g["bgr"])
    c2.build_pyramid(0, 1, from_bgr=True)
    assert np.array_equal(c2.download_level(0, 0), g["gray"])
    ctx.close(); c2.close()


# ------------------------------------------------------------------------------------- A2-A7
@pytest.mark.parametrize("w,h,tie", [(640, 480, 0), (640, 480, 1), (333, 251, 0)])
def test_fast_maps_and_keypoints_bit_exact(hip_lib, oracle, w, h, tie):
    imgs, _, _ = _frames(2, w, h, seed=2)
    ctx = make_ctx(hip_lib, width=w, height=h, levels=3, max_frames=2, debug_maps=True, nms_tie_suppress=tie)
    for s in range(2):
        ctx.upload_gray(s, imgs[s])
    ctx.build_pyramid(0, 2)
    ctx.detect(0, 2)
    prm = oracle.default_params(w, h, 3)
    prm.nms_tie_suppress = tie
    for s in range(2):
        lv = oracle.pyramid(imgs[s], 3)
        for L in range(3):
            smap, nmap = _corner_maps(oracle, lv[L], 15, tie)
            gs, gn = ctx.get_fast_maps(s, L)
            assert np.array_equal(gs, smap), ("fast score map", s, L)        # bit-exact FAST corners + scores
            assert np.array_equal(gn, nmap), ("nms map", s, L)
        ok_ = oracle.detect(lv, prm)
        assert len(ok_) > 200
        _kp_check(ctx.get_keypoints(s), ok_)
    ctx.close()


def test_detect_golden_and_occupied(hip_lib, oracle):
    g = golden("extract")
    ctx = make_ctx(hip_lib, width=320, height=240, levels=3, max_frames=2)
    for s in range(2):
        ctx.upload_gray(s, g["imgs"][s])
    ctx.build_pyramid(0, 2)
    ctx.detect(0, 2)
    _kp_check(ctx.get_keypoints(0), g["k0"])
    _kp_check(ctx.get_keypoints(1), g["k1"])
    occ = np.stack([g["occ"], np.zeros_like(g["occ"])])
    ctx.detect(0, 2, occupied=occ)                       # Detect(frame, overwrite_existing_features=false)
    _kp_check(ctx.get_keypoints(0), g["k0occ"])
    _kp_check(ctx.get_keypoints(1), g["k1"])
    # call order is enforced
    with pytest.raises(hip_lib.YgzHipError):
        c2 = make_ctx(hip_lib, width=320, height=240, levels=3, max_frames=1)
        c2.detect(0, 1)
    ctx.close()


def test_describe_arbitrary_pixels(hip_lib, oracle):
    imgs, _, _ = _frames(1, 640, 480, seed=4)
    rng = np.random.default_rng(0)
    n = 500
    level = rng.integers(0, 3, n).astype(np.int32)
    px = np.stack([rng.uniform(0, 640, n), rng.uniform(0, 480, n)], 1)      # includes border cases (reads wrap / 0)
    px[:20] = np.floor(px[:20]) + 0.5                                       # cvRound half-to-even cases
    ctx = make_ctx(hip_lib, max_frames=1)
    ctx.upload_gray(0, imgs[0]); ctx.build_pyramid(0, 1)
    ctx.describe(0, px, level)
    kp = ctx.get_keypoints(0)
    k = np.zeros(n, oracle.detect(oracle.pyramid(imgs[0], 3)).dtype)
    k["px"], k["py"], k["level"] = px[:, 0], px[:, 1], level
    ok_ = oracle.describe(oracle.pyramid(imgs[0], 3), k)
    assert np.array_equal(kp["angle"], ok_["angle"])
    assert np.array_equal(kp["desc"], ok_["desc"])
    ctx.close()


def test_find_direct_projection_per_candidate_bit_exact(hip_lib, oracle):
    """ygz_hip_find_direct_projection_mp: the call LocalMapping::ProjectMapPoints makes once per candidate (Matcher::FindDirectProjection, MapPoint
    overload, Matcher.cpp:356-383), n candidates over 3 keyframes in one launch, against the oracle one candidate at a time -- with the
    FindCandidates projection made by the launch, with the caller's predictions, and as n = 1 calls (what the class surface falls back to)"""
    imgs, poses, depths = _frames(4, 640, 480, seed=8, step=0.5)
    rng = np.random.default_rng(4)
    pos, cp, ck, cx, cl = _local_map_fixture(oracle, imgs, poses, depths, 3, 300, rng)
    pos[3] = [0.0, 0.0, -5.0]; pos[4] = [50.0, 0.0, 2.0]
    T_cur = oracle.se3_mul(synth.se3_exp([0.004, -0.003, 0.002, 0.001, -0.001, 0.0015]), poses[3])
    ctx = make_ctx(hip_lib, max_frames=4)
    for s in range(4):
        ctx.upload_gray(s, imgs[s])
    ctx.build_pyramid(0, 4)
    lv = [oracle.pyramid(imgs[s], 3) for s in range(4)]
    kfT = np.stack([poses[0], poses[1], poses[2]])
    r = ctx.find_direct_projection_mp(3, T_cur, [0, 1, 2], kfT, ck, pos[cp], cx, cl)
    # FindCandidates' projection and in-view test, per candidate (LocalMapping.cpp:58-63)
    _, ovis, oproj, omatch, opxm, olvl = oracle.track_local_map(lv[:3], poses[:3], lv[3], T_cur, pos, None, cp, ck, cx, cl)
    assert np.array_equal(r["in_view"], ovis[cp].astype(bool)) and not r["in_view"].all()
    v = r["in_view"]
    assert np.array_equal(r["px_proj"][v], oproj[cp][v])
    assert not r["ok"][~v].any()
    n_ok = 0
    for c in np.nonzero(v)[0]:
        o_ok, o_px, o_sl = oracle.find_direct_projection_mp(lv[ck[c]], poses[ck[c]], lv[3], T_cur, pos[cp[c]], cx[c], int(cl[c]), oproj[cp[c]])
        assert bool(o_ok) == bool(r["ok"][c]) and o_sl == r["level"][c] and np.array_equal(np.asarray(o_px), r["px"][c]), c
        n_ok += bool(o_ok)
    assert 500 < n_ok < v.sum() - 500                                          # successes and failures both occur (sub-pixel offsets, other levels)
    # the first success per point in candidate order is what ProjectMapPoints keeps
    for p_ in np.unique(cp):
        cs = np.nonzero((cp == p_) & r["ok"])[0]
        assert omatch[p_] == (cs[0] if len(cs) else -1)
        if len(cs):
            assert np.array_equal(opxm[p_], r["px"][cs[0]]) and olvl[p_] == r["level"][cs[0]]
    # the caller's predictions instead of the projection (they need not be in view): same answers where they coincide ...
    idx = np.nonzero(v)[0]
    g = ctx.find_direct_projection_mp(3, T_cur, [0, 1, 2], kfT, ck[idx], pos[cp[idx]], cx[idx], cl[idx], px_in=r["px_proj"][idx])
    assert np.array_equal(g["ok"], r["ok"][idx]) and np.array_equal(g["px"], r["px"][idx]) and np.array_equal(g["level"], r["level"][idx])
    # ... other predictions against the oracle, and one candidate per launch == the batch
    pin = r["px_proj"][idx[:60]] + rng.uniform(-1.5, 1.5, (60, 2))
    g = ctx.find_direct_projection_mp(3, T_cur, [0, 1, 2], kfT, ck[idx[:60]], pos[cp[idx[:60]]], cx[idx[:60]], cl[idx[:60]], px_in=pin)
    for j, c in enumerate(idx[:60]):
        o_ok, o_px, o_sl = oracle.find_direct_projection_mp(lv[ck[c]], poses[ck[c]], lv[3], T_cur, pos[cp[c]], cx[c], int(cl[c]), pin[j])
        assert bool(o_ok) == bool(g["ok"][j]) and o_sl == g["level"][j] and np.array_equal(np.asarray(o_px), g["px"][j])
        one = ctx.find_direct_projection_mp(3, T_cur, [int(ck[c])], kfT[ck[c]][None], [0], pos[cp[c]][None], cx[c][None], cl[c:c + 1], px_in=pin[j][None])
        assert one["ok"][0] == g["ok"][j] and one["level"][0] == g["level"][j] and np.array_equal(one["px"][0], g["px"][j])
    # the launch in two halves (_begin queues, _end collects) with other calls on the context in between -- among them the synchronous form, which
    # uses its own staging block -- returns what the one-call form returned; a second _begin replaces a pending run; _end without one is refused
    nb = ctx.find_direct_projection_mp_begin(3, T_cur, [0, 1, 2], kfT, ck, pos[cp], cx, cl)
    g2 = ctx.find_direct_projection_mp(3, T_cur, [0, 1, 2], kfT, ck[idx[:60]], pos[cp[idx[:60]]], cx[idx[:60]], cl[idx[:60]], px_in=pin)
    ctx.detect(0, 1); ctx.get_keypoints(0)
    h = ctx.find_direct_projection_mp_end(nb)
    assert all(np.array_equal(h[k], r[k]) for k in ("in_view", "ok", "level")) and np.array_equal(h["px_proj"][v], r["px_proj"][v]) and np.array_equal(h["px"][r["ok"]], r["px"][r["ok"]])
    assert np.array_equal(g2["px"], g["px"]) and np.array_equal(g2["ok"], g["ok"])
    ctx.find_direct_projection_mp_begin(3, T_cur, [0, 1, 2], kfT, ck[:100], pos[cp[:100]], cx[:100], cl[:100])
    nb = ctx.find_direct_projection_mp_begin(3, T_cur, [0, 1], kfT[:2], ck[idx[:200]] % 2, pos[cp[idx[:200]]], cx[idx[:200]], cl[idx[:200]])
    with pytest.raises(hip_lib.YgzHipError):
        ctx.find_direct_projection_mp_end(nb + 1)
    h = ctx.find_direct_projection_mp_end(nb)
    h1 = ctx.find_direct_projection_mp(3, T_cur, [0, 1], kfT[:2], ck[idx[:200]] % 2, pos[cp[idx[:200]]], cx[idx[:200]], cl[idx[:200]])
    assert np.array_equal(h["ok"], h1["ok"]) and np.array_equal(h["px"][h["ok"]], h1["px"][h1["ok"]]) and np.array_equal(h["level"], h1["level"])
    with pytest.raises(hip_lib.YgzHipError):
        ctx.find_direct_projection_mp_end(nb)
    # nothing to do / refused inputs
    e = ctx.find_direct_projection_mp(3, T_cur, [0], kfT[:1], [], np.zeros((0, 3)), np.zeros((0, 2)), [])
    assert len(e["ok"]) == 0
    with pytest.raises(hip_lib.YgzHipError):
        ctx.find_direct_projection_mp(3, T_cur, [0], kfT[:1], [0], pos[:1], cx[:1], [7])
    ctx.close()


def test_describe_given_angle(hip_lib, oracle):
    """ygz_hip_describe_given_angle (what FeatureDetector::ComputeDescriptor(Feature*) calls, FeatureDetector.cpp:588-594): the rotated BRIEF with the
    angles the caller supplies -- border pixels, half-to-even pixels and arbitrary angles -- against ComputeOrbDescriptor of the oracle"""
    imgs, _, _ = _frames(1, 640, 480, seed=4)
    rng = np.random.default_rng(1)
    n = 500
    level = rng.integers(0, 3, n).astype(np.int32)
    px = np.stack([rng.uniform(0, 640, n), rng.uniform(0, 480, n)], 1)
    px[:20] = np.floor(px[:20]) + 0.5
    angle = rng.uniform(0, 360, n).astype(np.float32)
    angle[:8] = [0.0, 90.0, 180.0, 270.0, 359.99997, 45.0, 1e-3, 123.456]
    ctx = make_ctx(hip_lib, max_frames=1)
    ctx.upload_gray(0, imgs[0]); ctx.build_pyramid(0, 1)
    ctx.describe_given_angle(0, px, level, angle)
    kp = ctx.get_keypoints(0)
    lv = oracle.pyramid(imgs[0], 3)
    want = np.stack([oracle.orb_descriptor(lv[level[i]], px[i, 0], px[i, 1], int(level[i]), float(angle[i])) for i in range(n)])
    assert np.array_equal(kp["desc"], want)
    assert np.array_equal(kp["angle"], angle)                                  # the angles are kept as given
    # with the intensity-centroid angles of ygz_hip_describe it is the same descriptor as ygz_hip_describe's
    ctx.describe(0, px, level)
    k0 = ctx.get_keypoints(0)
    ctx.describe_given_angle(0, px, level, k0["angle"])
    assert np.array_equal(ctx.get_keypoints(0)["desc"], k0["desc"])
    dev, cus = ctx.get_device()
    assert dev == 0 and cus == 256                                             # MI355X: 256 CUs
    ctx.close()


# ------------------------------------------------------------------------------------- M1-M3
@pytest.mark.parametrize("nq,nt", [(1000, 1000), (70, 53), (1, 1), (257, 3), (3, 700), (3072, 3072)])
def test_hamming_bit_exact_indices(hip_lib, oracle, nq, nt):
    q = fixtures.random_descriptors(nq, 42 + nq)
    t = fixtures.random_descriptors(nt, 43 + nt)
    if nq > 30 and nt > 12:
        t[10] = q[3]; t[11] = q[3]; q[20] = q[21]
    ctx = make_ctx(hip_lib, max_frames=1)
    for cc in (0, 1, 2):
        idx, d = ctx.hamming_match(q, t, cc)
        oi, od, _ = oracle.bf_match(q, t, cc)
        assert np.array_equal(idx, oi) and np.array_equal(d, od), cc
    idx, d, d2 = ctx.hamming_match(q, t, 0, want_second=True)
    oi, od, od2 = oracle.hamming_nn(q, t)
    assert np.array_equal(idx, oi) and np.array_equal(d, od) and np.array_equal(d2, od2)
    if nq <= 1000:                                     # independent numpy check of distances
        D = LUT[q[:, None, :] ^ t[None, :, :]].sum(-1)
        assert np.array_equal(d, D.min(1)) and np.array_equal(idx, D.argmin(1))
    ctx.close()


def test_hamming_edge_cases_and_golden(hip_lib, oracle):
    g = golden("hamming")
    ctx = make_ctx(hip_lib, max_frames=1)
    for cc in (0, 1, 2):
        idx, d = ctx.hamming_match(g["q"], g["t"], cc)
        assert np.array_equal(idx, g["idx%d" % cc]) and np.array_equal(d, g["dist%d" % cc])
    e = np.zeros((0, 32), np.uint8)
    idx, d = ctx.hamming_match(g["q"], e, 1)
    assert np.all(idx == -1) and np.all(d == 2 ** 31 - 1)
    idx, d = ctx.hamming_match(e, g["t"], 1)
    assert len(idx) == 0
    with pytest.raises(hip_lib.YgzHipError):             # capacity: more rows than the result buffers hold (grid cells x max_frames)
        ctx.hamming_match(fixtures.random_descriptors(4000, 1), g["t"], 0)
    c4 = make_ctx(hip_lib, max_frames=4)                  # ... and with four frames' worth of rows a 4000 x 9000 search fits
    q4, t4 = fixtures.random_descriptors(4000, 1), fixtures.random_descriptors(9000, 2)
    for cc in (0, 1):
        idx, d = c4.hamming_match(q4, t4, cc)
        oi, od, _ = oracle.bf_match(q4, t4, cc)
        assert np.array_equal(idx, oi) and np.array_equal(d, od), cc
    c4.close()
    z = np.zeros((5, 32), np.uint8); f = np.full((4, 32), 255, np.uint8)
    idx, d = ctx.hamming_match(z, f, 0)
    assert np.all(d == 256) and np.all(idx == 0)         # maximum distance, first index on ties
    ctx.close()


def test_hamming_more_than_65535_rows(hip_lib, oracle):
    """sets larger than 65535 rows (a 4K frame has 82944 grid cells): the kernel that packs (distance, row) into one 32-bit key
    must not be used; cross-check and second-best paths stay exact"""
    ctx = make_ctx(hip_lib, width=3840, height=2160, levels=3, max_frames=1)
    q = fixtures.random_descriptors(70000, 5)
    t = fixtures.random_descriptors(700, 6)
    t[650] = q[69990]; q[66000] = q[65999]                  # an exact match and a duplicate beyond row 65535
    for cc in (0, 1, 2):
        idx, d = ctx.hamming_match(t, q, cc)                # the 70000-row set is scanned: indices up to 69999
        oi, od, _ = oracle.bf_match(t, q, cc)
        assert np.array_equal(idx, oi) and np.array_equal(d, od), cc
    assert idx[650] == 69990
    idx, d = ctx.hamming_match(q, t, 1)                     # and the other way round (70000 query rows)
    oi, od, _ = oracle.bf_match(q, t, 1)
    assert np.array_equal(idx, oi) and np.array_equal(d, od)
    ctx.close()


def test_match_slots_on_extracted_frames(hip_lib, oracle):
    imgs, _, _ = _frames(4, 640, 480, seed=5, step=0.3)
    ctx = make_ctx(hip_lib, max_frames=4)
    for s in range(4):
        ctx.upload_gray(s, imgs[s])
    ctx.build_pyramid(0, 4); ctx.detect(0, 4)
    ks = [oracle.detect(oracle.pyramid(imgs[s], 3)) for s in range(4)]
    for cc in (1, 0, 2):
        ctx.match_slots([0, 1, 2], [1, 2, 3], cc)
        for p in range(3):
            idx, d = ctx.get_matches(p)
            oi, od, _ = oracle.bf_match(ks[p]["desc"], ks[p + 1]["desc"], cc)
            assert np.array_equal(idx, oi) and np.array_equal(d, od), (cc, p)
    ctx.match_slots_again(1)
    idx, d = ctx.get_matches(0)
    oi, od, n = oracle.bf_match(ks[0]["desc"], ks[1]["desc"], 1)
    assert np.array_equal(idx, oi) and n > 100
    ctx.close()


# ------------------------------------------------------------------------------------- L1-L2
def _fdp_inputs(oracle, imgs, poses, depths, n, rng, noise=2.0):
    k0 = oracle.detect(oracle.pyramid(imgs[0], 3))[:n]
    px_ref = np.stack([k0["px"], k0["py"]], 1)
    depth = np.array([depths[0][int(p[1]), int(p[0])] for p in px_ref])
    Tcr = oracle.se3_mul(poses[1], oracle.se3_inv(poses[0]))
    R = synth.quat_to_R(Tcr[:4])
    pc = np.stack([(px_ref[:, 0] - synth.CX) / synth.FX * depth, (px_ref[:, 1] - synth.CY) / synth.FY * depth, depth], 1) @ R.T + Tcr[4:]
    pred = np.stack([synth.FX * pc[:, 0] / pc[:, 2] + synth.CX, synth.FY * pc[:, 1] / pc[:, 2] + synth.CY], 1)
    return px_ref, depth, k0["level"].astype(np.int32), pred + rng.uniform(-noise, noise, pred.shape)


def test_find_direct_projection_bit_exact(hip_lib, oracle):
    imgs, poses, depths = _frames(2, 640, 480, seed=6, step=0.6)
    rng = np.random.default_rng(1)
    px_ref, depth, level, pred = _fdp_inputs(oracle, imgs, poses, depths, 1000, rng)
    depth[5] = -1.0                                        # invalid depth -> false (Matcher.cpp:388-392)
    pred[7] = [3.0, 3.0]                                   # border
    for T_ref_case in (poses[0], synth.se3_exp([0.02, -0.01, 0.03, 0.01, 0.02, -0.01])):
        ctx = make_ctx(hip_lib, max_frames=2)
        for s in range(2):
            ctx.upload_gray(s, imgs[s])
        ctx.build_pyramid(0, 2)
        ok, px, sl = ctx.find_direct_projection(0, T_ref_case, 1, poses[1], px_ref, depth, level, pred)
        lv0, lv1 = oracle.pyramid(imgs[0], 3), oracle.pyramid(imgs[1], 3)
        for i in range(len(depth)):
            o_ok, o_px, o_sl = oracle.find_direct_projection(lv0, T_ref_case, lv1, poses[1], px_ref[i], depth[i], int(level[i]), pred[i])
            assert ok[i] == o_ok, i
            if depth[i] >= 0:
                assert sl[i] == o_sl, i
                assert np.array_equal(px[i], o_px, equal_nan=True), (i, px[i], o_px)      # bit-exact float path
        if T_ref_case is poses[0]:
            assert ok.mean() > 0.5
        ctx.close()


def test_single_call_entry_points_take_more_rows_than_grid_cells(hip_lib, oracle):
    """ygz_hip_find_direct_projection / ygz_hip_klt_track(_filtered) serve any number of candidates / points (the reference's
    Matcher::FindDirectProjection and cv::calcOpticalFlowPyrLK have no limit): in pieces of `cells` rows inside the call, equal to the caller's
    own pieces."""
    imgs, poses, depths = _frames(2, 320, 240, seed=9, step=0.5)
    ctx = make_ctx(hip_lib, width=320, height=240, max_frames=2)
    for s_ in range(2):
        ctx.upload_gray(s_, imgs[s_])
    ctx.build_pyramid(0, 2)
    cells = ctx.cells
    n = 2 * cells + 37
    rng = np.random.default_rng(3)
    px_ref = np.stack([rng.uniform(30, 290, n), rng.uniform(30, 210, n)], 1)
    depth = depths[0][px_ref[:, 1].astype(int), px_ref[:, 0].astype(int)].astype(np.float64)
    level = rng.integers(0, 3, n).astype(np.int32)
    pred = px_ref + rng.uniform(-3, 3, px_ref.shape)
    ok, px, sl = ctx.find_direct_projection(0, poses[0], 1, poses[1], px_ref, depth, level, pred)
    for b in range(0, n, cells):
        e = min(n, b + cells)
        ok_p, px_p, sl_p = ctx.find_direct_projection(0, poses[0], 1, poses[1], px_ref[b:e], depth[b:e], level[b:e], pred[b:e])
        assert np.array_equal(ok[b:e], ok_p) and np.array_equal(px[b:e], px_p, equal_nan=True) and np.array_equal(sl[b:e], sl_p)
    assert ok.sum() > n // 4
    pts = px_ref.astype(np.float32)
    out, st, err = ctx.klt_track(0, 1, pts, pts)
    out_f, st_f, err_f, keep, n_keep = ctx.klt_track_filtered(0, 1, pts, pts, border=20)
    assert np.array_equal(out, out_f, equal_nan=True) and np.array_equal(st, st_f) and n_keep == int(keep.sum())
    for b in range(0, n, cells):
        e = min(n, b + cells)
        o_p, s_p, e_p = ctx.klt_track(0, 1, pts[b:e], pts[b:e])
        assert np.array_equal(out[b:e], o_p, equal_nan=True) and np.array_equal(st[b:e], s_p) and np.array_equal(err[b:e], e_p, equal_nan=True)
    assert st.sum() > n // 2
    ctx.close()


def _local_map_fixture(oracle, imgs, poses, depths, n_kf, per_kf, rng):
    """map points from the first n_kf frames; every point gets an observation in each keyframe that sees it"""
    pos, cand = [], []
    for kf in range(n_kf):
        kps = oracle.detect(oracle.pyramid(imgs[kf], 3))
        sel = rng.permutation(len(kps))[:per_kf]
        Twc = oracle.se3_inv(poses[kf])
        R = synth.quat_to_R(Twc[:4])
        for i in sel:
            x, y = kps["px"][i], kps["py"][i]
            d = depths[kf][int(y), int(x)]
            pc = np.array([(x - synth.CX) / synth.FX * d, (y - synth.CY) / synth.FY * d, d])
            pw = R @ pc + Twc[4:] + rng.normal(0, 0.002, 3)
            p = len(pos); pos.append(pw)
            cand.append((p, kf, x, y, int(kps["level"][i])))
            for o in range(n_kf):                                  # co-visible observations (sub-pixel positions, other levels)
                if o == kf:
                    continue
                q = synth.project(poses[o], pw[None])[0][0]
                if 30 < q[0] < 610 and 30 < q[1] < 450 and rng.random() < 0.7:
                    cand.append((p, o, q[0] + rng.uniform(-0.5, 0.5), q[1] + rng.uniform(-0.5, 0.5), int(rng.integers(0, 3))))
    pos = np.array(pos)
    cand = [cand[i] for i in rng.permutation(len(cand))]          # the reference's order is heap-address order: any order must work
    cp = np.array([c[0] for c in cand], np.int32); ck = np.array([c[1] for c in cand], np.int32)
    cx = np.array([[c[2], c[3]] for c in cand]); cl = np.array([c[4] for c in cand], np.int32)
    return pos, cp, ck, cx, cl


def test_track_local_map(hip_lib, oracle):
    """LocalMapping::FindCandidates + ProjectMapPoints in one launch vs the oracle's sequential restatement"""
    imgs, poses, depths = _frames(4, 640, 480, seed=8, step=0.5)
    rng = np.random.default_rng(4)
    pos, cp, ck, cx, cl = _local_map_fixture(oracle, imgs, poses, depths, 3, 300, rng)
    P = len(pos)
    bad = (rng.random(P) < 0.05).astype(np.uint8)
    pos[3] = [0.0, 0.0, -5.0]                               # behind the current camera
    pos[4] = [50.0, 0.0, 2.0]                               # far outside the image
    pos[6] = oracle.se3_inv(poses[3])[4:]                   # at the camera centre: z = 0 -> inf / nan pixel -> not in view
    T_cur = oracle.se3_mul(synth.se3_exp([0.004, -0.003, 0.002, 0.001, -0.001, 0.0015]), poses[3])     # prediction a little off
    ctx = make_ctx(hip_lib, max_frames=4)
    for s in range(4):
        ctx.upload_gray(s, imgs[s])
    ctx.build_pyramid(0, 4)
    lv = [oracle.pyramid(imgs[s], 3) for s in range(4)]
    n, vis, proj, match, pxm, lvl = ctx.track_local_map(3, T_cur, [0, 1, 2], [poses[0], poses[1], poses[2]], pos, bad, cp, ck, cx, cl)
    on, ovis, oproj, omatch, opxm, olvl = oracle.track_local_map(lv[:3], poses[:3], lv[3], T_cur, pos, bad, cp, ck, cx, cl)
    assert np.array_equal(vis, ovis) and vis[3] == 0 and vis[4] == 0 and vis[6] == 0
    inv = vis.astype(bool)
    assert np.array_equal(proj[inv], oproj[inv])                               # bit-exact doubles
    assert np.array_equal(match, omatch) and n == on
    assert np.array_equal(pxm, opxm) and np.array_equal(lvl, olvl)
    assert n > 0.5 * inv.sum() and (match[~inv] == -1).all() and (match[bad.astype(bool)] == -1).all()
    # some point's first candidate failed and a later one matched: the "first success in order" rule is exercised
    first = {}
    for c, p_ in enumerate(cp):
        first.setdefault(int(p_), c)
    assert any(match[p_] >= 0 and match[p_] != first[p_] for p_ in first)
    # empty inputs
    n0, v0, *_ = ctx.track_local_map(3, T_cur, [0], [poses[0]], pos[:5], None, [], [], np.zeros((0, 2)), [])
    assert n0 == 0 and len(v0) == 5
    n0, v0, *_ = ctx.track_local_map(3, T_cur, [0], [poses[0]], np.zeros((0, 3)), None, [], [], np.zeros((0, 2)), [])
    assert n0 == 0 and len(v0) == 0
    # candidates that name no point / no keyframe are ignored by both; an impossible pyramid level is refused
    cp2, ck2 = cp[:50].copy(), ck[:50].copy()
    cp2[3] = -1; cp2[4] = P + 7; ck2[5] = 9; ck2[6] = -2
    r_g = ctx.track_local_map(3, T_cur, [0, 1, 2], [poses[0], poses[1], poses[2]], pos, bad, cp2, ck2, cx[:50], cl[:50])
    r_o = oracle.track_local_map(lv[:3], poses[:3], lv[3], T_cur, pos, bad, cp2, ck2, cx[:50], cl[:50])
    assert r_g[0] == r_o[0] and np.array_equal(r_g[3], r_o[3]) and np.array_equal(r_g[4], r_o[4])
    with pytest.raises(hip_lib.YgzHipError):
        ctx.track_local_map(3, T_cur, [0], [poses[0]], pos[:5], None, [0], [0], [[100.0, 100.0]], [7])
    with pytest.raises(hip_lib.YgzHipError):                # a keyframe slot whose pyramid was never built
        ctx2 = make_ctx(hip_lib, max_frames=2)
        try:
            ctx2.track_local_map(0, T_cur, [1], [poses[0]], pos[:5], None, [0], [0], [[100.0, 100.0]], [0])
        finally:
            ctx2.close()
    ctx.close()


def test_align2d_patches_bit_exact(hip_lib, oracle):
    imgs, _, _ = _frames(2, 640, 480, seed=7, step=0.2)
    rng = np.random.default_rng(2)
    n = 600
    cx = rng.integers(10, 630, n); cy = rng.integers(10, 470, n)
    pwb = np.stack([imgs[0][y - 5:y + 5, x - 5:x + 5] for x, y in zip(cx, cy)]).reshape(n, 100)
    uv = np.stack([cx + rng.uniform(-2, 2, n), cy + rng.uniform(-2, 2, n)], 1)
    uv[:10] = [[1.0, 1.0]] * 10                             # outside the 4 px margin: breaks at once
    ctx = make_ctx(hip_lib, max_frames=1)
    ctx.upload_gray(0, imgs[1]); ctx.build_pyramid(0, 1)
    ok, out, chi2 = ctx.align2d(0, 0, pwb, uv)
    for i in range(n):
        o_ok, u, v, c2, it = oracle.align2d(imgs[1], pwb[i], pwb[i].reshape(10, 10)[1:9, 1:9].copy(), uv[i, 0], uv[i, 1])
        assert ok[i] == o_ok and out[i, 0] == u and out[i, 1] == v and chi2[i] == np.float32(c2), i
    ctx.close()


def test_fdp_golden(hip_lib):
    g, e = golden("align"), golden("extract")
    ctx = make_ctx(hip_lib, width=320, height=240, max_frames=2)
    for s in range(2):
        ctx.upload_gray(s, e["imgs"][s])
    ctx.build_pyramid(0, 2)
    ok, px, sl = ctx.find_direct_projection(0, e["poses"][0], 1, e["poses"][1], g["px_ref"], g["depth"], g["level"], g["pred"])
    assert np.array_equal(ok, g["fdp_ok"]) and np.array_equal(sl, g["fdp_sl"]) and np.array_equal(px, g["fdp_px"], equal_nan=True)
    ctx.close()


# ------------------------------------------------------------------------------------- L3
@pytest.mark.parametrize("lanes", ["256", "512"])
def test_sparse_align(hip_lib, oracle, lanes, monkeypatch):
    # both workgroup shapes of the kernel (the launcher picks by the number of problems; YGZ_SA_THREADS pins one)
    monkeypatch.setenv("YGZ_SA_THREADS", lanes)
    for (w, h, n, seed) in ((320, 240, 200, 3), (640, 480, 1000, 8)):
        imgs, poses, depths = _frames(2, w, h, seed=seed, step=0.4)
        k0 = oracle.detect(oracle.pyramid(imgs[0], 3), oracle.default_params(w, h, 3))[:n]
        px = np.stack([k0["px"], k0["py"]], 1)
        depth = np.array([depths[0][int(p[1]), int(p[0])] for p in px])
        has_mp = np.ones(len(px), np.uint8); has_mp[::9] = 0
        T_init = oracle.se3_mul(synth.se3_exp([0.004, -0.003, 0.002, 0.001, -0.002, 0.001]), poses[1])
        ctx = make_ctx(hip_lib, width=w, height=h, max_frames=2)
        for s in range(2):
            ctx.upload_gray(s, imgs[s])
        ctx.build_pyramid(0, 2)
        nm, T, iters = ctx.sparse_align(0, poses[0], 1, T_init, px, depth, has_mp)
        onm, oT, st = oracle.sparse_align(oracle.pyramid(imgs[0], 3), poses[0], oracle.pyramid(imgs[1], 3), T_init, px, depth, has_mp)
        assert nm == onm
        assert iters == list(st.iters_per_level)[:3]          # same Gauss-Newton trajectory (chi2 reproduced exactly)
        assert np.allclose(T, oT, rtol=1e-9, atol=1e-11)      # FP64 H/Jres sums differ only in order
        e0 = np.linalg.norm(oracle.se3_log(oracle.se3_mul(T_init, oracle.se3_inv(poses[1]))))
        e1 = np.linalg.norm(oracle.se3_log(oracle.se3_mul(T, oracle.se3_inv(poses[1]))))
        assert e1 < 0.5 * e0
        ctx.close()


def test_sparse_align_residuals(hip_lib, oracle):
    """ygz_hip_sparse_align_residuals = one SparseImgAlign::computeResiduals(model, linearize = true) (SparseImageAlign.cpp:124-223 with
    precomputeReferencePatches :59-122) at a model the caller holds, per level: the float chi2 sum and the measurement count equal the oracle's bit for
    bit, H_ and Jres_ to 1e-9 -- the step the class surface's Levenberg-Marquardt is built from"""
    w, h, n = 640, 480, 1000
    imgs, poses, depths = _frames(2, w, h, seed=8, step=0.4)
    k0 = oracle.detect(oracle.pyramid(imgs[0], 3), oracle.default_params(w, h, 3))[:n]
    px = np.stack([k0["px"], k0["py"]], 1)
    depth = np.array([depths[0][int(p[1]), int(p[0])] for p in px])
    has_mp = np.ones(len(px), np.uint8); has_mp[::9] = 0
    ctx = make_ctx(hip_lib, width=w, height=h, max_frames=2)
    for s in range(2):
        ctx.upload_gray(s, imgs[s])
    ctx.build_pyramid(0, 2)
    lv = [oracle.pyramid(imgs[s], 3) for s in range(2)]
    T_rel = oracle.se3_mul(oracle.se3_mul(synth.se3_exp([0.004, -0.003, 0.002, 0.001, -0.002, 0.001]), poses[1]), oracle.se3_inv(poses[0]))
    for level in (2, 1, 0):
        csum, nm, H, J = ctx.sparse_align_residuals(0, 1, T_rel, px, depth, has_mp, level)
        o_chi2, oH, oJ, onm, _ = oracle.sparse_align_linearize(lv[0], lv[1], T_rel, px, depth, has_mp, level)
        assert nm == onm and nm > 8000
        assert float(np.float32(csum) / np.float32(nm)) == o_chi2                  # chi2 / n_meas_ as the reference forms it (float / size_t)
        assert np.allclose(H, oH, rtol=1e-9, atol=1e-9 * np.abs(oH).max()) and np.allclose(J, oJ, rtol=1e-9, atol=1e-9 * np.abs(oJ).max())
    csum, nm, H, J = ctx.sparse_align_residuals(0, 1, T_rel, np.zeros((0, 2)), np.zeros(0), np.zeros(0, np.uint8), 0)
    assert csum == 0 and nm == 0 and not H.any()
    with pytest.raises(hip_lib.YgzHipError):
        ctx.sparse_align_residuals(0, 1, T_rel, px, depth, has_mp, 5)
    ctx.close()


def test_sparse_align_more_than_16384_grid_cells(hip_lib, oracle):
    """a 1920x1080 frame has 192 x 108 = 20 736 grid cells: beyond the 32 x 512 features the first form of the kernel could mark per problem (it
    returned YGZ_E_CAPACITY); the second form marks entering / leaving features in their flags byte.  ~6000 features against the oracle."""
    w, h = 1920, 1080
    imgs, poses, depths = _frames(2, w, h, seed=5, step=0.3)
    k0 = oracle.detect(oracle.pyramid(imgs[0], 3), oracle.default_params(w, h, 3))
    px = np.stack([k0["px"], k0["py"]], 1)
    depth = np.array([depths[0][int(p[1]), int(p[0])] for p in px])
    has_mp = np.ones(len(px), np.uint8); has_mp[::11] = 0
    T_init = oracle.se3_mul(synth.se3_exp([0.003, -0.002, 0.002, 0.001, -0.001, 0.001]), poses[1])
    ctx = make_ctx(hip_lib, width=w, height=h, max_frames=2)
    assert ctx.cells == 20736 and len(px) > 4000
    for s_ in range(2):
        ctx.upload_gray(s_, imgs[s_])
    ctx.build_pyramid(0, 2)
    nm, T, iters = ctx.sparse_align(0, poses[0], 1, T_init, px, depth, has_mp)
    onm, oT, st = oracle.sparse_align(oracle.pyramid(imgs[0], 3), poses[0], oracle.pyramid(imgs[1], 3), T_init, px, depth, has_mp)
    assert nm == onm and iters == list(st.iters_per_level)[:3]
    assert np.allclose(T, oT, rtol=1e-9, atol=1e-11)
    ctx.close()


def test_sparse_align_golden_and_empty(hip_lib):
    g, e = golden("align"), golden("extract")
    ctx = make_ctx(hip_lib, width=320, height=240, max_frames=2)
    for s in range(2):
        ctx.upload_gray(s, e["imgs"][s])
    ctx.build_pyramid(0, 2)
    nm, T, iters = ctx.sparse_align(0, e["poses"][0], 1, g["T_init"], g["px_ref"], g["depth"], g["has_mp"])
    assert nm == int(g["sa_nmeas"]) and iters == list(g["sa_iters"])
    assert np.allclose(T, g["sa_T"], rtol=1e-9, atol=1e-11)
    nm, T, _ = ctx.sparse_align(0, e["poses"][0], 1, g["T_init"], np.zeros((0, 2)), np.zeros(0), np.zeros(0, np.uint8))
    assert nm == 0 and np.array_equal(T, g["T_init"])      # run() returns 0 and leaves the pose (SparseImageAlign.cpp:25-29)
    no_mp = np.zeros(len(g["depth"]), np.uint8)            # no feature has a map point -> nothing visible
    nm, T, _ = ctx.sparse_align(0, e["poses"][0], 1, g["T_init"], g["px_ref"], g["depth"], no_mp)
    assert nm == 0
    # ygz_hip_set_wait_hook: the caller's own host work between the launch and the wait of the NEXT single-frame alignment -- called once, on the
    # calling thread, then cleared; the result does not change; a hook that was set and cleared again is never called
    import ctypes as C
    calls = []
    HOOK = C.CFUNCTYPE(None, C.c_void_p)
    hook = HOOK(lambda user: calls.append(user))
    ctx.lib.ygz_hip_set_wait_hook.argtypes = [C.c_void_p, HOOK, C.c_void_p]
    assert ctx.lib.ygz_hip_set_wait_hook(ctx._ctx, hook, C.c_void_p(7)) == 0
    nm2, T2, it2 = ctx.sparse_align(0, e["poses"][0], 1, g["T_init"], g["px_ref"], g["depth"], g["has_mp"])
    assert calls == [7] and nm2 == int(g["sa_nmeas"]) and it2 == list(g["sa_iters"]) and np.allclose(T2, g["sa_T"], rtol=1e-9, atol=1e-11)
    nm3, T3, _ = ctx.sparse_align(0, e["poses"][0], 1, g["T_init"], g["px_ref"], g["depth"], g["has_mp"])
    assert calls == [7] and np.array_equal(T3, T2)
    assert ctx.lib.ygz_hip_set_wait_hook(ctx._ctx, hook, C.c_void_p(8)) == 0 and ctx.lib.ygz_hip_set_wait_hook(ctx._ctx, HOOK(0), None) == 0
    ctx.sparse_align(0, e["poses"][0], 1, g["T_init"], g["px_ref"], g["depth"], g["has_mp"])
    assert calls == [7]
    ctx.close()


# ------------------------------------------------------------------------------------- L4
def test_klt(hip_lib, oracle):
    for (w, h, n, seed) in ((320, 240, 200, 3), (640, 480, 1000, 9)):
        imgs, poses, depths = _frames(2, w, h, seed=seed, step=0.5)
        k0 = oracle.detect(oracle.pyramid(imgs[0], 3), oracle.default_params(w, h, 3))[:n]
        pts = np.stack([k0["px"], k0["py"]], 1).astype(np.float32)
        rng = np.random.default_rng(4)
        init = pts + rng.uniform(-3, 3, pts.shape).astype(np.float32)
        pts[0] = [2.0, 2.0]; init[0] = [1.0, 1.0]           # window mostly outside: reflect-101 border path
        ctx = make_ctx(hip_lib, width=w, height=h, max_frames=2)
        for s in range(2):
            ctx.upload_gray(s, imgs[s])
        ctx.build_pyramid(0, 2)
        out, st, err = ctx.klt_track(0, 1, pts, init)
        oout, ost, oerr = oracle.klt_track(imgs[0], imgs[1], pts, init)
        assert np.array_equal(st, ost)
        m = ost.astype(bool)
        assert m.mean() > 0.5
        # tracks within 1e-5 relative (north_star); the float normal-equation sums are tree-ordered on the GPU
        assert np.all(np.abs(out[m] - oout[m]).max(1) <= 1e-5 * np.maximum(1.0, np.abs(oout[m]).max(1)))    # relative to the track's magnitude
        # err = sum |J - I| / (32 * win * win) at the final position, an INTEGER sum of 1/32 grey levels scaled by one constant: where the
        # track is bit-equal the error is bit-equal; where the track differs in its last bits a 14-bit bilinear weight may flip by one
        # unit and move a few window pixels by one count: err differs by k / (32 * win * win) with a small integer k.  (The tracker
        # never reads err, Tracker.cpp:100-112.)
        same = m & np.all(out == oout, axis=1)
        assert np.array_equal(err[same], oerr[same])
        quantum = 1.0 / (32 * 21 * 21)                       # the default tracker: 21 x 21 window (Tracker.cpp:97)
        k = (err[m].astype(np.float64) - oerr[m].astype(np.float64)) / quantum
        assert np.all(np.abs(k - np.rint(k)) < 0.05 + 1e-6 * np.abs(oerr[m]) / quantum) and np.abs(k).max() <= 64, (np.abs(k).max(), float(same[m].mean()))     # measured: |k| <= 27, 4 % of the tracks bit-equal
        ctx.close()


def test_lk_framed_copies_written_by_the_pyramid_kernels(hip_lib):
    """The tracker's working images (every level inside a 24-pixel BORDER_REFLECT_101 frame, cv::buildOpticalFlowPyramid's copyMakeBorder) are
    written by the pyramid kernels once the tracker's buffers exist: k_bgr2gray16 / k_pyr_down store the interior AND the frame, k_klt_frame the
    small levels and gray uploads, k_klt_pad only the first time.  Every path against numpy's reflect padding, all five levels, and the tracks
    unchanged whichever kernel wrote the images."""
    rng = np.random.default_rng(11)
    for (w, h, bgr) in ((640, 480, True), (640, 480, False), (1280, 720, True), (328, 250, True), (96, 64, True)):
        ctx = make_ctx(hip_lib, width=w, height=h, max_frames=2)
        img = [rng.integers(0, 256, (h, w, 3) if bgr else (h, w), dtype=np.uint8) for _ in range(2)]
        up = ctx.upload_bgr if bgr else ctx.upload_gray
        for s_ in range(2):
            up(s_, img[s_])
        ctx.build_pyramid(0, 2, from_bgr=bgr)
        assert ctx.download_framed_level(0, 0) is None                 # the tracker has not run: no framed copies yet
        n = 50
        pts = np.stack([rng.uniform(0, w, n), rng.uniform(0, h, n)], 1).astype(np.float32)
        a = ctx.klt_track(0, 1, pts, pts)                              # k_klt_pad builds them (and allocates the buffers)
        levels = 5
        seen = set()
        def check(tag):
            for s_ in range(2):
                for L in range(levels):
                    try:
                        f = ctx.download_framed_level(s_, L)
                    except hip_lib.YgzHipError:                          # a level this context never allocated (small frames)
                        break
                    if f is None:                                       # a level the tracker does not use at this size
                        continue
                    seen.add((tag, L))
                    ref = np.pad(ctx.download_level(s_, L), 24, mode="reflect")
                    assert np.array_equal(f, ref), (tag, w, h, bgr, s_, L, np.argwhere(f != ref)[:4])
        check("k_klt_pad")
        for s_ in range(2):                                             # new content: the pyramid kernels write the framed copies themselves
            img[s_] = rng.integers(0, 256, img[s_].shape, dtype=np.uint8)
            up(s_, img[s_])
        ctx.build_pyramid(0, 2, from_bgr=bgr)
        check("pyramid kernels")
        if w >= 640:                                                    # all five levels, both ways
            assert seen == {(t, L) for t in ("k_klt_pad", "pyramid kernels") for L in range(5)}, seen
        b = ctx.klt_track(0, 1, pts, pts)
        ctx.close()
        ctx = make_ctx(hip_lib, width=w, height=h, max_frames=2)        # the same frames through k_klt_pad: identical tracks
        for s_ in range(2):
            up = ctx.upload_bgr if bgr else ctx.upload_gray
            up(s_, img[s_])
        ctx.build_pyramid(0, 2, from_bgr=bgr)
        c = ctx.klt_track(0, 1, pts, pts)
        for x, y in zip(b, c):
            assert np.array_equal(x, y, equal_nan=True)
        ctx.close()


def test_klt_golden(hip_lib):
    g, e = golden("align"), golden("extract")
    ctx = make_ctx(hip_lib, width=320, height=240, max_frames=2)
    for s in range(2):
        ctx.upload_gray(s, e["imgs"][s])
    ctx.build_pyramid(0, 2)
    out, st, err = ctx.klt_track(0, 1, g["px_ref"].astype(np.float32), g["klt_init"])
    assert np.array_equal(st, g["klt_status"])
    m = st.astype(bool)
    assert np.all(np.abs(out[m] - g["klt_pts"][m]).max(1) <= 1e-5 * np.maximum(1.0, np.abs(g["klt_pts"][m]).max(1)))
    ctx.close()


# ------------------------------------------------------------------------------------- B1-B5
def _ba_close(g, r, tol=1e-5):
    for k in ("err", "chi2_edge", "Hll", "bl", "Hpl", "Hpp", "bp"):
        a, b = np.asarray(g[k]), np.asarray(r[k])
        scale = max(1.0, np.abs(b).max())
        assert np.all(np.abs(a - b) <= 1e-9 * scale), (k, np.abs(a - b).max(), scale)      # far inside the 1e-5 bar
    assert abs(g["chi2"] - r["chi2"]) <= 1e-9 * max(1.0, abs(r["chi2"]))


def test_ba_golden_known_answer(hip_lib):
    ctx = make_ctx(hip_lib, max_frames=1)
    for name in ("ba_exact", "ba_noisy"):
        g = golden(name)
        r = ctx.ba_linearize(g["poses"], g["fixed"], g["points"], g["edge_pose"], g["edge_point"], g["obs"])
        _ba_close(r, {k[2:]: (float(g[k]) if k == "o_chi2" else g[k]) for k in g.files if k.startswith("o_")})
        if name == "ba_exact":                               # test/test_local_ba.cpp fixture: residuals vanish
            assert np.abs(r["err"]).max() < 1e-9
    ctx.close()


def test_ba_window_10x2000(hip_lib, oracle):
    f = synth.ba_window(10, 2000, seed=7)
    assert len(f["obs"]) > 10000
    f["obs"][::97] += 30.0                                   # outliers beyond the Huber delta
    ctx = make_ctx(hip_lib, max_frames=1)
    g = ctx.ba_linearize(f["poses"], f["fixed"], f["points"], f["edge_pose"], f["edge_point"], f["obs"])
    r = oracle.ba_linearize(f["poses"], f["fixed"], f["points"], f["edge_pose"], f["edge_point"], f["obs"])
    _ba_close(g, r)
    # unsorted edge order gives the same blocks (CSR is built by the ABI)
    perm = np.random.default_rng(0).permutation(len(f["obs"]))
    g2 = ctx.ba_linearize(f["poses"], f["fixed"], f["points"], f["edge_pose"][perm], f["edge_point"][perm], f["obs"][perm])
    assert np.allclose(g2["Hll"], g["Hll"], rtol=1e-12, atol=1e-9) and np.allclose(g2["err"], g["err"][perm])
    # resident form, 3 LM-like state updates without re-uploading the structure
    K, P, E = ctx.ba_upload(0, f["poses"], f["fixed"], f["points"], f["edge_pose"], f["edge_point"], f["obs"])
    rng = np.random.default_rng(1)
    poses, pts = f["poses"].copy(), f["points"].copy()
    for _ in range(3):
        poses[1:] += rng.normal(0, 1e-3, poses[1:].shape); pts += rng.normal(0, 1e-3, pts.shape)
        ctx.ba_set_state(0, poses, pts)
        ctx.ba_linearize_resident(0, 1)
        _ba_close(ctx.ba_download(0, K, P, E), oracle.ba_linearize(poses, f["fixed"], pts, f["edge_pose"], f["edge_point"], f["obs"]))
    # ygz_hip_ba_set_enable: edges switched off in the resident graph (what an inlier test between rounds does) == the oracle on the edges left
    en = (rng.random(E) < 0.8).astype(np.uint8)
    ctx.ba_set_enable(0, en)
    ctx.ba_linearize_resident(0, 1)
    m = en.astype(bool)
    gd = ctx.ba_download(0, K, P, E)
    rs = oracle.ba_linearize(poses, f["fixed"], pts, f["edge_pose"][m], f["edge_point"][m], f["obs"][m])
    for k in ("Hpp", "bp", "Hll", "bl"):
        assert np.all(np.abs(gd[k] - rs[k]) <= 1e-9 * max(1.0, np.abs(rs[k]).max())), k
    assert abs(gd["chi2"] - rs["chi2"]) <= 1e-9 * rs["chi2"]
    for k in ("Hpl", "err", "chi2_edge"):
        assert np.all(np.abs(gd[k][m] - rs[k]) <= 1e-9 * max(1.0, np.abs(rs[k]).max())), k
    ctx.ba_set_enable(0, np.ones(E, np.uint8))               # ... and on again
    ctx.ba_linearize_resident(0, 1)
    _ba_close(ctx.ba_download(0, K, P, E), oracle.ba_linearize(poses, f["fixed"], pts, f["edge_pose"], f["edge_point"], f["obs"]))
    with pytest.raises(hip_lib.YgzHipError):
        ctx.ba_set_enable(5, en)                             # no such window
    # Huber off
    g3 = ctx.ba_linearize(f["poses"], f["fixed"], f["points"], f["edge_pose"], f["edge_point"], f["obs"], huber_delta=0.0)
    assert abs(g3["chi2"] - g3["chi2_edge"].sum()) < 1e-6 * g3["chi2"]
    ctx.close()


def test_ba_normalised_plane_formulation(hip_lib, oracle):
    f = synth.ba_window(6, 300, seed=3)
    cam = oracle.camera()
    obs_n = np.stack([(f["obs"][:, 0] - cam.cx) / cam.fx, (f["obs"][:, 1] - cam.cy) / cam.fy], 1)
    poses_tr = np.concatenate([f["poses"][:, 3:], f["poses"][:, :3]], 1)
    ctx = make_ctx(hip_lib, max_frames=1)
    g = ctx.ba_linearize(poses_tr, f["fixed"], f["points"], f["edge_pose"], f["edge_point"], obs_n, huber_delta=0.0, formulation=1)
    Hll = np.zeros_like(g["Hll"]); Hpp = np.zeros_like(g["Hpp"])
    for e in range(len(obs_n)):
        ip, il = f["edge_pose"][e], f["edge_point"][e]
        er, Jp, Jx = oracle.ba_edge_norm(poses_tr[ip], f["points"][il], obs_n[e])
        assert np.allclose(g["err"][e], er, atol=1e-12)
        Hll[il] += Jp.T @ Jp
        if not f["fixed"][ip]:
            Hpp[ip] += Jx.T @ Jx
    assert np.allclose(g["Hll"], Hll, rtol=1e-10, atol=1e-12) and np.allclose(g["Hpp"], Hpp, rtol=1e-10, atol=1e-12)
    ctx.close()


# ------------------------------------------------------------------------------------- resident batched path
@pytest.mark.parametrize("overlap", [False, True])
def test_resident_batched_tracking(hip_lib, oracle, overlap):
    """detect -> (device-side) track sets -> KLT / FindDirectProjection / SparseImgAlign for 3 pairs in one launch each;
    overlap=True runs sparse alignment on a side stream concurrently with KLT / direct projection"""
    imgs, poses, depths = _frames(4, 640, 480, seed=11, step=0.3)
    ctx = make_ctx(hip_lib, max_frames=4)
    ctx.set_overlap(overlap)
    for s in range(4):
        ctx.upload_gray(s, imgs[s])
    ctx.build_pyramid(0, 4); ctx.detect(0, 4)
    kps = [ctx.get_keypoints(s) for s in range(4)]
    deps, mps = [], []
    for s in range(4):
        d = np.array([depths[s][int(p[1]), int(p[0])] for p in kps[s]["px"]])
        d[::50] = -1.0                                           # features without depth
        m = np.ones(len(d), np.uint8); m[::5] = 0
        deps.append(d); mps.append(m)
        ctx.set_keypoint_depths(s, d, m)
    cur, ref = [1, 2, 3], [0, 1, 2]
    lv = [oracle.pyramid(imgs[s], 3) for s in range(4)]
    cam = oracle.camera()
    for predict in (False, True):
        ctx.track_begin(cur, ref, poses[cur], poses[ref], predict=predict)
        ctx.track_sparse_align(); ctx.track_klt(); ctx.track_direct()
        for p in range(3):
            c, r = cur[p], ref[p]
            px = kps[r]["px"]
            # KLT from the reference pixels
            pts = px.astype(np.float32)
            out, st, err = ctx.track_get_klt(p)
            oout, ost, oerr = oracle.klt_track(imgs[r], imgs[c], pts, pts)
            assert np.array_equal(st, ost)
            m = ost.astype(bool)
            assert np.all(np.abs(out[m] - oout[m]).max(1) <= 1e-5 * np.maximum(1.0, np.abs(oout[m]).max(1)))    # relative to the track's magnitude
            # direct projection
            ok, pxo, sl = ctx.track_get_direct(p)
            Tcr = oracle.se3_mul(poses[c], oracle.se3_inv(poses[r]))
            for i in range(0, len(px), 2):
                start = px[i].copy()
                if predict and deps[r][i] > 0:
                    pr = np.array([(px[i, 0] - np.float64(cam.cx)) * deps[r][i] / np.float64(cam.fx),
                                   (px[i, 1] - np.float64(cam.cy)) * deps[r][i] / np.float64(cam.fy), deps[r][i]])
                    pc = oracle.se3_act(Tcr, pr)
                    start = np.array([np.float64(cam.fx) * pc[0] / pc[2] + np.float64(cam.cx), np.float64(cam.fy) * pc[1] / pc[2] + np.float64(cam.cy)])
                o_ok, o_px, o_sl = oracle.find_direct_projection(lv[r], poses[r], lv[c], poses[c], px[i], deps[r][i], int(kps[r]["level"][i]), start)
                assert ok[i] == o_ok, (p, i)
                if deps[r][i] >= 0:
                    assert sl[i] == o_sl and np.array_equal(pxo[i], o_px, equal_nan=True), (p, i)
            # sparse alignment starting from the reference pose
            nm, T, iters = ctx.track_get_pose(p)
            onm, oT, st_ = oracle.sparse_align(lv[r], poses[r], lv[c], poses[r], px, deps[r], mps[r])
            assert nm == onm and iters == list(st_.iters_per_level)[:3]
            assert np.allclose(T, oT, rtol=1e-9, atol=1e-11)
    # reload + re-run gives identical results (no stale state between steps)
    ctx.track_reload(True); ctx.track_klt(); ctx.track_direct(); ctx.track_sparse_align()
    nm2, T2, _ = ctx.track_get_pose(2)
    assert nm2 == nm and np.array_equal(T2, T)
    # LK's working images built ahead, beside the extractor (ygz_hip_track_klt_prepare): the same tracks bit for bit
    klt_ref = [ctx.track_get_klt(p) for p in range(3)]
    ctx.build_pyramid(0, 4); ctx.track_klt_prepare(); ctx.detect(0, 4)
    ctx.track_reload(True); ctx.track_sparse_align(); ctx.track_direct(); ctx.track_klt()
    for p in range(3):
        for a_, b_ in zip(ctx.track_get_klt(p), klt_ref[p]):
            assert np.array_equal(a_, b_, equal_nan=True), p
    ctx.build_pyramid(0, 4); ctx.track_klt_prepare(); ctx.build_pyramid(0, 4)      # prepared, then overwritten: rebuilt by the LK call
    ctx.detect(0, 4); ctx.track_reload(True); ctx.track_klt()
    for p in range(3):
        for a_, b_ in zip(ctx.track_get_klt(p), klt_ref[p]):
            assert np.array_equal(a_, b_, equal_nan=True), p
    ctx.close()


def test_ba_repeated_point_pose_edges(hip_lib, oracle):
    """the same (map point, free pose) pair observed twice -- two features of one frame sharing a map point, as ba::OptimizeCurrent can
    build (BA.cpp:91-186): every residual block counts, in Hpp / bp too.  Linearisation vs the oracle, the LM loop (falls back to the
    host-side reduced system), the ceres solver, and the resident LM kernel refusing such a window."""
    f = synth.ba_window(6, 300, seed=21)
    rng = np.random.default_rng(4)
    dup = rng.choice(len(f["obs"]), 90, replace=False)
    ep = np.concatenate([f["edge_pose"], f["edge_pose"][dup], f["edge_pose"][dup[:10]]]).astype(np.int32)        # some pairs three times
    el = np.concatenate([f["edge_point"], f["edge_point"][dup], f["edge_point"][dup[:10]]]).astype(np.int32)
    obs = np.concatenate([f["obs"], f["obs"][dup] + rng.normal(0, 1.5, (90, 2)), f["obs"][dup[:10]] + rng.normal(0, 1.5, (10, 2))])
    perm = rng.permutation(len(ep))                                                                              # unsorted edge list
    ep, el, obs = ep[perm], el[perm], obs[perm]
    ctx = make_ctx(hip_lib, max_frames=2)
    for form in (0, 1):
        g = ctx.ba_linearize(f["poses"], f["fixed"], f["points"], ep, el, obs, formulation=form)
        if form == 0:
            r = oracle.ba_linearize(f["poses"], f["fixed"], f["points"], ep, el, obs)
            for k in ("err", "chi2_edge", "Hpp", "bp", "Hll", "bl", "Hpl"):
                assert np.allclose(g[k], r[k], rtol=1e-9, atol=1e-7), k
            assert abs(g["chi2"] - r["chi2"]) <= 1e-9 * r["chi2"]
        # the blocks of a free pose are the sums over ALL its edges: rebuild Hpp from single-edge problems
        k = 3
        sel = np.nonzero(ep == k)[0]
        Hk = np.zeros(36); bk = np.zeros(6)
        for e in sel[:40]:
            one = ctx.ba_linearize(f["poses"], f["fixed"], f["points"], ep[e:e + 1], el[e:e + 1], obs[e:e + 1], formulation=form)
            Hk += one["Hpp"][k].ravel(); bk += one["bp"][k]
        part = ctx.ba_linearize(f["poses"], f["fixed"], f["points"], ep[sel[:40]], el[sel[:40]], obs[sel[:40]], formulation=form)
        assert np.allclose(part["Hpp"][k].ravel(), Hk, rtol=1e-10, atol=1e-6) and np.allclose(part["bp"][k], bk, rtol=1e-10, atol=1e-6)
    ctx.ba_upload(0, f["poses"], f["fixed"], f["points"], ep, el, obs)
    with pytest.raises(hip_lib.YgzHipError) as ei:
        ctx.ba_optimize_resident(0, 1, 5)
    assert ei.value.code == hip_lib.E_INVALID
    po, pt, st = ctx.ba_optimize(f["poses"], f["fixed"], f["points"], ep, el, obs, iterations=10)
    opo, opt_, ost = oracle.g2o_lm(f["poses"], f["fixed"], f["points"], ep, el, obs, max_iterations=10)
    assert st.iterations == ost["iterations"] and abs(st.chi2_final - ost["chi2_final"]) <= 1e-8 * ost["chi2_final"]
    assert np.allclose(po, opo, rtol=1e-6, atol=1e-8) and np.allclose(pt, opt_, rtol=1e-6, atol=1e-8)
    c = fixtures.ba_to_ceres(dict(f, obs=obs, edge_pose=ep, edge_point=el))
    gpo, gpt, gsm = ctx.ba_solve_ceres(c["poses"], c["fixed"], c["points"], ep, el, c["obs_n"])
    cpo, cpt, csm = oracle.ceres_solve(c["poses"], c["fixed"], c["points"], ep, el, c["obs_n"])
    assert gsm["iterations"] == csm["iterations"] and gsm["termination"] == csm["termination"]
    assert abs(gsm["final_cost"] - csm["final_cost"]) <= 1e-6 * csm["final_cost"]
    ctx.close()


def test_ba_batched_windows(hip_lib, oracle):
    ctx = make_ctx(hip_lib, max_frames=1)
    fs = [synth.ba_window(10, 500, seed=20 + i) for i in range(3)] + [synth.ba_window(6, 200, seed=30)]
    dims = [ctx.ba_upload(w, f["poses"], f["fixed"], f["points"], f["edge_pose"], f["edge_point"], f["obs"]) for w, f in enumerate(fs)]
    ctx.ba_linearize_resident(0, 4)                       # ragged windows in one set of launches
    for w, f in enumerate(fs):
        _ba_close(ctx.ba_download(w, *dims[w]), oracle.ba_linearize(f["poses"], f["fixed"], f["points"], f["edge_pose"], f["edge_point"], f["obs"]))
    ctx.close()


# ------------------------------------------------------------------------------------- B3 / B6 / B7: the ceres side
def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_ba_ceres_formulation_blocks(hip_lib, oracle):
    """formulation 2 = the auto-differentiated ceres functors (Ceres/CeresReprojectionError*.h), closed form on the GPU vs
    Jets in the oracle, with constant poses / points, disabled edges and per-edge Huber widths.  Bar 1e-5 relative; met 1e-9."""
    c = fixtures.ba_to_ceres(synth.ba_window(8, 600, seed=4))
    c["poses"][0] = 0.0                                            # keyframe 0 at the identity: the first-order rotation branch
    E, P = len(c["obs_n"]), len(c["points"])
    rng = np.random.default_rng(0)
    huber = np.where(rng.random(E) < 0.5, 0.002, 0.0)
    enable = (rng.random(E) < 0.9).astype(np.uint8)
    pfix = (rng.random(P) < 0.1).astype(np.uint8)
    o = oracle.ceres_linearize(c["poses"], c["fixed"], c["points"], c["edge_pose"], c["edge_point"], c["obs_n"],
                               point_fixed=pfix, edge_huber=huber, edge_enable=enable)
    ctx = make_ctx(hip_lib, max_frames=1)
    g = ctx.ba_linearize(c["poses"], c["fixed"], c["points"], c["edge_pose"], c["edge_point"], c["obs_n"], huber_delta=0.0,
                         formulation=2, point_fixed=pfix, edge_huber=huber, edge_enable=enable)
    assert g["n_behind"] == 0
    assert abs(0.5 * g["chi2"] - o["cost"]) <= 1e-12 * o["cost"]
    for k in ("Hpp", "bp", "Hll", "bl", "Hpl"):
        assert _rel(g[k], o[k]) < 1e-9, k
    assert not g["Hll"][pfix == 1].any() and not g["Hpl"][enable == 0].any() and not g["err"][enable == 0].any()
    # raw residuals of enabled edges (the oracle's `res` carries the sqrt(rho') of the corrector)
    raw = np.array([oracle.ceres_edge(c["poses"][c["edge_pose"][e]], c["points"][c["edge_point"][e]], c["obs_n"][e])[0]
                    for e in range(0, E, 37)])
    assert np.allclose(g["err"][::37][enable[::37] == 1], raw[enable[::37] == 1], atol=1e-13)
    # behind-the-camera count (what makes the PoseOnly functor fail)
    pts = c["points"].copy(); pts[:5, 2] = -4.0
    g2 = ctx.ba_linearize(c["poses"], c["fixed"], pts, c["edge_pose"], c["edge_point"], c["obs_n"], huber_delta=0.0, formulation=2)
    assert g2["n_behind"] == int(np.isin(c["edge_point"], np.arange(5)).sum())
    ctx.close()


def test_ba_solve_ceres_local_ba(hip_lib, oracle):
    """ba::LocalBA (BA.cpp:324-384): trust-region LM around the GPU linearisation vs the oracle's restatement: same
    accept/reject sequence, same termination, final cost and state within 1e-6 relative (bar 1e-5)."""
    for fx in (fixtures.ba_to_ceres(synth.ba_window(6, 300, seed=5)), fixtures.ba_to_ceres(fixtures.ba_fixture_test_local_ba(noise=True))):
        po, pt, so = oracle.ceres_solve(fx["poses"], fx["fixed"], fx["points"], fx["edge_pose"], fx["edge_point"], fx["obs_n"])
        ctx = make_ctx(hip_lib, max_frames=1)
        pg, tg, sg = ctx.ba_solve_ceres(fx["poses"], fx["fixed"], fx["points"], fx["edge_pose"], fx["edge_point"], fx["obs_n"])
        ctx.close()
        assert (sg["iterations"], sg["successful_steps"], sg["unsuccessful_steps"], sg["termination"]) == \
               (so["iterations"], so["successful_steps"], so["unsuccessful_steps"], so["termination"])
        assert abs(sg["initial_cost"] - so["initial_cost"]) <= 1e-12 * so["initial_cost"]
        assert abs(sg["final_cost"] - so["final_cost"]) <= 1e-6 * so["final_cost"]
        assert _rel(pg, po) < 1e-6 and _rel(tg, pt) < 1e-6
        assert sg["final_cost"] < 0.05 * sg["initial_cost"]


def test_ba_solve_ceres_variants(hip_lib, oracle):
    """OptimizeCurrentPointOnly (every pose constant), OptimizeCurrent-like (Huber 0.1 on every edge, one free pose) and a
    pose-only problem with the behind-camera failure rule -- the other ceres call sites of BA.cpp on the same entry point."""
    fx = fixtures.ba_to_ceres(synth.ba_window(5, 200, seed=9))
    ctx = make_ctx(hip_lib, max_frames=1)
    allfix = np.ones(len(fx["poses"]), np.uint8)
    po, pt, so = oracle.ceres_solve(fx["poses"], allfix, fx["points"], fx["edge_pose"], fx["edge_point"], fx["obs_n"])
    pg, tg, sg = ctx.ba_solve_ceres(fx["poses"], allfix, fx["points"], fx["edge_pose"], fx["edge_point"], fx["obs_n"])
    assert sg["termination"] == so["termination"] and sg["iterations"] == so["iterations"]
    assert np.array_equal(pg, fx["poses"]) and _rel(tg, pt) < 1e-6 and sg["final_cost"] < so["initial_cost"]
    onefree = np.ones(len(fx["poses"]), np.uint8); onefree[-1] = 0
    hub = np.full(len(fx["obs_n"]), 0.1)
    po, pt, so = oracle.ceres_solve(fx["poses"], onefree, fx["points"], fx["edge_pose"], fx["edge_point"], fx["obs_n"], edge_huber=hub)
    pg, tg, sg = ctx.ba_solve_ceres(fx["poses"], onefree, fx["points"], fx["edge_pose"], fx["edge_point"], fx["obs_n"], edge_huber=hub)
    assert sg["termination"] == so["termination"] and sg["iterations"] == so["iterations"]
    assert _rel(pg, po) < 1e-6 and _rel(tg, pt) < 1e-6
    # pose only, a point behind the camera at the start: ceres gives up at iteration zero, state untouched
    f = fixtures.pose_only_fixture(n=60, seed=8, outlier_frac=0.0)
    n = len(f["px"])
    obs_n = np.stack([(f["px"][:, 0] - synth.CX) / synth.FX, (f["px"][:, 1] - synth.CY) / synth.FY], axis=1)
    pw = f["pw"].copy(); pw[7, 2] = -3.0
    args = (f["entry"][None], None, pw, np.zeros(n, np.int32), np.arange(n, dtype=np.int32), obs_n)
    po, _, so = oracle.ceres_solve(*args, point_fixed=np.ones(n, np.uint8), fail_behind=True)
    pg, _, sg = ctx.ba_solve_ceres(*args, point_fixed=np.ones(n, np.uint8), options=ctx.ceres_options(fail_behind_camera=1))
    assert so["termination"] == 5 and sg["termination"] == 5 and np.array_equal(pg[0], f["entry"]) and np.array_equal(po[0], f["entry"])
    ctx.close()


def test_ba_solve_ceres_dogleg(hip_lib, oracle):
    """options.trust_region_strategy_type = DOGLEG (ba::TwoViewBACeres, BA.cpp:58-62): ygz_hip_ba_solve_ceres with the DoglegStrategy (host loop around the
    GPU linearisations) against the oracle's restatement -- same accepted / rejected steps, iteration count and termination, final cost and state 1e-6 --
    on a local-BA window, a small window and a TwoViewBACeres-shaped problem (one constant pose, HuberLoss(0.1) on some residual blocks)"""
    ctx = make_ctx(hip_lib, max_frames=1)
    gopt, oopt = ctx.ceres_options(trust_region_strategy=1), oracle.ceres_options(trust_region_strategy=1)
    rng = np.random.default_rng(3)
    for w, with_huber in ((synth.ba_window(6, 300, seed=5), False), (synth.ba_window(4, 60, seed=9), False), (synth.ba_window(2, 200, seed=4), True)):
        c = fixtures.ba_to_ceres(w)
        huber = np.where(rng.random(len(c["obs_n"])) < 0.3, 0.1, 0.0) if with_huber else None
        pg, xg, sg = ctx.ba_solve_ceres(c["poses"], c["fixed"], c["points"], c["edge_pose"], c["edge_point"], c["obs_n"], edge_huber=huber, options=gopt)
        po, xo, so = oracle.ceres_solve(c["poses"], c["fixed"], c["points"], c["edge_pose"], c["edge_point"], c["obs_n"], edge_huber=huber, options=oopt)
        assert ctx.ba_last_path()[0] is False                                  # the resident kernel has the LM strategy only
        assert (sg["iterations"], sg["successful_steps"], sg["unsuccessful_steps"], sg["termination"]) == \
               (so["iterations"], so["successful_steps"], so["unsuccessful_steps"], so["termination"]), (sg, so)
        assert abs(sg["final_cost"] - so["final_cost"]) <= 1e-6 * so["final_cost"] + 1e-14 and so["successful_steps"] >= 2
        # the state: a window with one constant pose keeps the monocular scale as a free direction, and the Gauss-Newton system of the dogleg carries
        # only mu_ = 1e-8 of regularisation along it (LM damps it with 1 / radius): rounding moves the state ALONG that direction (1e-4 relative
        # measured) without moving the cost (1e-6 above)
        assert np.allclose(pg, po, rtol=1e-3, atol=1e-6) and np.allclose(xg, xo, rtol=1e-3, atol=1e-5)
        assert np.isclose(sg["final_radius"], so["final_radius"], rtol=1e-3)
    ctx.close()


def test_ba_solve_ceres_resident_windows(hip_lib, oracle, monkeypatch):
    """SURVEY 8f-1, second half: the ceres trust-region loop resident on the GPU (k_ba_ceres), five different windows in ONE launch --
    ba::LocalBA problems of two sizes, all poses constant (OptimizeCurrentPointOnly), HuberLoss(0.1) with one free pose
    (OptimizeCurrent), and a pose-only window that fails at iteration zero (point behind the camera) -- against the oracle's
    ceres::Solve restatement and against the host-loop form of the same entry point (YGZ_BA_HOST_LOOP=1)."""
    fa = fixtures.ba_to_ceres(synth.ba_window(6, 300, seed=5)); fb = fixtures.ba_to_ceres(fixtures.ba_fixture_test_local_ba(noise=True))
    fc = fixtures.ba_to_ceres(synth.ba_window(5, 200, seed=9))
    onefree = np.ones(len(fc["poses"]), np.uint8); onefree[-1] = 0
    f = fixtures.pose_only_fixture(n=60, seed=8, outlier_frac=0.0)
    n = len(f["px"])
    obs_po = np.stack([(f["px"][:, 0] - synth.CX) / synth.FX, (f["px"][:, 1] - synth.CY) / synth.FY], axis=1)
    pw = f["pw"].copy(); pw[7, 2] = -3.0
    wins = [dict(poses=fa["poses"], fixed=fa["fixed"], points=fa["points"], ep=fa["edge_pose"], el=fa["edge_point"], obs=fa["obs_n"]),
            dict(poses=fb["poses"], fixed=fb["fixed"], points=fb["points"], ep=fb["edge_pose"], el=fb["edge_point"], obs=fb["obs_n"]),
            dict(poses=fc["poses"], fixed=np.ones(len(fc["poses"]), np.uint8), points=fc["points"], ep=fc["edge_pose"], el=fc["edge_point"], obs=fc["obs_n"]),
            dict(poses=fc["poses"], fixed=onefree, points=fc["points"], ep=fc["edge_pose"], el=fc["edge_point"], obs=fc["obs_n"], huber=np.full(len(fc["obs_n"]), 0.1))]
    ctx = make_ctx(hip_lib, max_frames=1)
    for w, d in enumerate(wins):
        ctx.ba_upload(w, d["poses"], d["fixed"], d["points"], d["ep"], d["el"], d["obs"], huber_delta=0.0, formulation=2, edge_huber=d.get("huber"))
    sums = ctx.ba_solve_ceres_resident(0, len(wins))
    for w, d in enumerate(wins):
        po, pt, so = oracle.ceres_solve(d["poses"], d["fixed"], d["points"], d["ep"], d["el"], d["obs"], edge_huber=d.get("huber"))
        sg = sums[w]
        assert (sg["iterations"], sg["successful_steps"], sg["unsuccessful_steps"], sg["termination"]) == \
               (so["iterations"], so["successful_steps"], so["unsuccessful_steps"], so["termination"]), w
        assert abs(sg["initial_cost"] - so["initial_cost"]) <= 1e-11 * so["initial_cost"] and abs(sg["final_cost"] - so["final_cost"]) <= 1e-6 * so["final_cost"]
        pg, tg = ctx.ba_get_state(w, len(d["poses"]), len(d["points"]))
        assert _rel(pg, po) < 1e-6 and _rel(tg, pt) < 1e-6, w
        fixed = d["fixed"].astype(bool)
        assert np.array_equal(pg[fixed], d["poses"][fixed])                           # constant poses are not touched
    # the failure at iteration zero (PoseOnly functor, p_z < 0) and the host-loop form of the same problems
    ctx.ba_upload(0, f["entry"][None], None, pw, np.zeros(n, np.int32), np.arange(n, dtype=np.int32), obs_po, huber_delta=0.0, formulation=2,
                  point_fixed=np.ones(n, np.uint8))
    sg = ctx.ba_solve_ceres_resident(0, 1, options=ctx.ceres_options(fail_behind_camera=1))[0]
    pg, _ = ctx.ba_get_state(0, 1, n)
    assert sg["termination"] == 5 and np.array_equal(pg[0], f["entry"])
    d = wins[0]
    r_po, r_pt, r_s = ctx.ba_solve_ceres(d["poses"], d["fixed"], d["points"], d["ep"], d["el"], d["obs"])
    monkeypatch.setenv("YGZ_BA_HOST_LOOP", "1")
    h_po, h_pt, h_s = ctx.ba_solve_ceres(d["poses"], d["fixed"], d["points"], d["ep"], d["el"], d["obs"])
    assert (r_s["iterations"], r_s["termination"]) == (h_s["iterations"], h_s["termination"]) and _rel(r_po, h_po) < 1e-8 and _rel(r_pt, h_pt) < 1e-8
    ctx.close()


def test_optimize_pose_only_batch(hip_lib, oracle):
    """ba::OptimizeCurrentPoseOnly for a batch of frames in one launch (one workgroup per frame, four rounds on the device)
    vs the oracle frame by frame: flags, inlier counts and rounds equal; pose within 1e-7 relative (bar 1e-5), depth 1e-9."""
    frames = [fixtures.pose_only_fixture(n=400, seed=3), fixtures.pose_only_fixture(n=1000, seed=5, outlier_frac=0.3),
              fixtures.pose_only_fixture(n=37, seed=6, outlier_frac=0.0), fixtures.pose_only_fixture(n=12, seed=4, outlier_frac=0.0),
              dict(entry=np.zeros(6), px=np.zeros((0, 2)), pw=np.zeros((0, 3))), fixtures.pose_only_fixture(n=257, seed=7)]
    frames[3]["entry"] = frames[3]["entry"] + np.array([0.5, 0.5, 0, 0, 0, 0])        # < 10 inliers in round 1: no commit
    frames[5]["pw"][11, 2] = -2.0        # behind the camera: the first solve fails, the re-classification disables the point
    off = np.concatenate([[0], np.cumsum([len(f["px"]) for f in frames])]).astype(np.int32)
    ctx = make_ctx(hip_lib, max_frames=1)
    po, bad, dep, inl, rounds = ctx.optimize_pose_only(off, np.concatenate([f["px"] for f in frames]),
                                                       np.concatenate([f["pw"] for f in frames]), np.stack([f["entry"] for f in frames]))
    ctx.close()
    for i, f in enumerate(frames):
        o_pose, o_bad, o_dep, o_inl, o_rounds = oracle.optimize_current_pose_only(f["entry"], f["px"], f["pw"])
        s = slice(off[i], off[i + 1])
        assert rounds[i] == o_rounds and inl[i] == o_inl, i
        assert np.array_equal(bad[s], o_bad), i
        assert np.abs(po[i] - o_pose).max() <= 1e-7 * max(1.0, np.abs(o_pose).max()), i
        m = o_bad == 0
        assert np.allclose(dep[s][m], o_dep[m], rtol=1e-9) and np.all(np.isnan(dep[s][~m]) == np.isnan(o_dep[~m])), i
    assert rounds[3] == 1 and np.array_equal(po[3], frames[3]["entry"]) and rounds[4] == 1 and inl[4] == 0
    assert rounds[5] == 4 and bad[off[5] + 11] == 1 and not np.array_equal(po[5], frames[5]["entry"])


# ------------------------------------------------------------------------------------- config 5 geometry: 1280x720
def test_pipeline_1280x720(hip_lib, oracle):
    """SURVEY 8d config 5 frame size (128 x 72 = 9216 grid cells): every stage of one frame pair against the oracle."""
    W, H = 1280, 720
    imgs, poses, depths = _frames(2, W, H, seed=17, step=0.3)
    ctx = make_ctx(hip_lib, width=W, height=H, max_frames=2)
    assert ctx.cells == 9216
    for s in range(2):
        ctx.upload_gray(s, imgs[s])
    ctx.build_pyramid(0, 2); ctx.detect(0, 2)
    lv = [oracle.pyramid(imgs[s], 3) for s in range(2)]
    kps = [ctx.get_keypoints(s) for s in range(2)]
    for s in range(2):
        for L in range(3):
            assert np.array_equal(ctx.download_level(s, L), lv[s][L])
        _kp_check(kps[s], oracle.detect(lv[s]))
    assert len(kps[0]["level"]) > 2000
    # brute-force cross-checked matching of the two descriptor sets
    idx, dist = ctx.hamming_match(kps[0]["desc"], kps[1]["desc"], cross_check=1)
    oi, od, _ = oracle.bf_match(kps[0]["desc"], kps[1]["desc"], 1)
    assert np.array_equal(idx, oi) and np.array_equal(dist[oi >= 0], od[oi >= 0])
    # tracking stages on the resident pair
    d = np.array([depths[0][int(p[1]), int(p[0])] for p in kps[0]["px"]])
    m = np.ones(len(d), np.uint8)
    ctx.set_keypoint_depths(0, d, m)
    ctx.track_begin([1], [0], poses[[1]], poses[[0]], predict=False)
    ctx.track_klt(); ctx.track_direct(); ctx.track_sparse_align()
    px = kps[0]["px"]
    sel = np.arange(0, len(px), 5)                      # the oracle's KLT takes ~2 ms per point
    out, st, err = ctx.track_get_klt(0)
    pts = px.astype(np.float32)
    oout, ost, _ = oracle.klt_track(imgs[0], imgs[1], pts[sel], pts[sel])
    assert np.array_equal(st[sel], ost)
    mk = ost.astype(bool)
    assert np.all(np.abs(out[sel][mk] - oout[mk]).max(1) <= 1e-5 * np.maximum(1.0, np.abs(oout[mk]).max(1)))
    ok, pxo, sl = ctx.track_get_direct(0)
    for i in range(0, len(px), 25):
        o_ok, o_px, o_sl = oracle.find_direct_projection(lv[0], poses[0], lv[1], poses[1], px[i], d[i], int(kps[0]["level"][i]), px[i])
        assert ok[i] == o_ok and sl[i] == o_sl and np.array_equal(pxo[i], o_px, equal_nan=True), i
    nm, T, iters = ctx.track_get_pose(0)
    onm, oT, st_ = oracle.sparse_align(lv[0], poses[0], lv[1], poses[0], px, d, m)
    assert nm == onm and iters == list(st_.iters_per_level)[:3]
    assert np.allclose(T, oT, rtol=1e-9, atol=1e-11)
    ctx.close()


# ------------------------------------------------------------------------------------- LM loop resident on the GPU
def test_ba_optimize_resident_windows(hip_lib, oracle):
    """ba::LocalBAG2O's Levenberg-Marquardt loop (g2o LM + Schur + Cholesky) as ONE kernel, several windows per launch, vs the
    oracle's restatement (yo_g2o_lm) and the host-loop form: same iteration / trial counts, chi2 and state within 1e-6 relative
    (bar 1e-5; the block reductions sum in a different order than the scalar loops)."""
    wins = [fixtures.ba_fixture_test_local_ba(noise=True, seed=5), synth.ba_window(6, 300, seed=5), synth.ba_window(10, 2000, seed=7),
            synth.ba_window(4, 50, seed=9), synth.ba_window(8, 700, seed=3, sort_by_point=False)]
    ctx = make_ctx(hip_lib, max_frames=1)
    for i, w in enumerate(wins):
        ctx.ba_upload(i, w["poses"], w["fixed"], w["points"], w["edge_pose"], w["edge_point"], w["obs"])
    assert ctx.ba_light_barrier() == -1                      # no team launch yet
    stats = ctx.ba_optimize_resident(0, len(wins), iterations=20)
    # the same-XCD barrier (no L2 write-back) is only used after its message-passing self-test passed on this device: it does on an MI355X in SPX mode
    assert ctx.ba_light_barrier() == (int(os.environ["YGZ_LM_XCD_BARRIER"] != "0") if "YGZ_LM_XCD_BARRIER" in os.environ else 1)
    for i, w in enumerate(wins):
        po, pt, so = oracle.g2o_lm(w["poses"], w["fixed"], w["points"], w["edge_pose"], w["edge_point"], w["obs"], max_iterations=20)
        pg, tg = ctx.ba_get_state(i, len(w["poses"]), len(w["points"]))
        st = stats[i]
        assert 3 <= st.iterations <= 20 and st.lm_trials >= st.iterations, i
        assert abs(st.chi2_initial - so["chi2_initial"]) <= 1e-10 * so["chi2_initial"], i
        assert abs(st.chi2_final - so["chi2_final"]) <= 1e-9 * so["chi2_final"], i
        assert _rel(pg, po) < 1e-6 and _rel(tg, pt) < 1e-6, i
        assert st.chi2_final < 0.1 * st.chi2_initial and np.array_equal(pg[0], w["poses"][0]), i     # keyframe 0 fixed
        # the state left in HBM reproduces the reported minimum
        back = oracle.ba_linearize(pg, w["fixed"], tg, w["edge_pose"], w["edge_point"], w["obs"])
        assert abs(back["chi2"] - st.chi2_final) <= 1e-9 * st.chi2_final, i
    # host-loop form (reduced system solved on the CPU) on one of them
    w = wins[1]
    os.environ["YGZ_BA_HOST_LOOP"] = "1"
    try:
        ph, th, sh = ctx.ba_optimize(w["poses"], w["fixed"], w["points"], w["edge_pose"], w["edge_point"], w["obs"])
    finally:
        del os.environ["YGZ_BA_HOST_LOOP"]
    assert abs(sh.chi2_final - stats[1].chi2_final) <= 1e-9 * sh.chi2_final
    assert _rel(ph, ctx.ba_get_state(1, len(w["poses"]), len(w["points"]))[0]) < 1e-6
    # the caller can tell which loop ran (the host loop is ~10x slower): forced here, the resident kernel by default
    assert ctx.ba_last_path() == (False, ["YGZ_BA_HOST_LOOP=1"])
    ctx.ba_optimize(w["poses"], w["fixed"], w["points"], w["edge_pose"], w["edge_point"], w["obs"])
    assert ctx.ba_last_path() == (True, [])
    # a window that repeats a (point, pose) edge cannot run on the resident kernel: reported, not silent
    e = int(np.nonzero(np.asarray(w["fixed"])[w["edge_pose"]] == 0)[0][0])          # an edge to a FREE pose, once more
    ep = np.concatenate([w["edge_pose"], w["edge_pose"][e:e + 1]]); el = np.concatenate([w["edge_point"], w["edge_point"][e:e + 1]])
    ob = np.concatenate([w["obs"], w["obs"][e:e + 1] + 0.25])
    ctx.ba_optimize(w["poses"], w["fixed"], w["points"], ep, el, ob, iterations=2)
    assert ctx.ba_last_path() == (False, ["repeated (point, pose) edges"])
    ctx.close()


def test_ba_optimize_resident_many_free_poses(hip_lib, oracle):
    """The reduced system of the resident LM lives in dynamic LDS sized by the window: 19 keyframes (18 free poses, a 108 x 108 system) beside a small
    window in the same launch, against the oracle's g2o LM; 22 keyframes go to the host loop and say so."""
    wins = [synth.ba_window(19, 1200, seed=11), synth.ba_window(5, 120, seed=12), synth.ba_window(12, 800, seed=13)]
    ctx = make_ctx(hip_lib, max_frames=1)
    for i, w in enumerate(wins):
        ctx.ba_upload(i, w["poses"], w["fixed"], w["points"], w["edge_pose"], w["edge_point"], w["obs"])
    stats = ctx.ba_optimize_resident(0, len(wins), iterations=20)
    for i, w in enumerate(wins):
        po, pt, so = oracle.g2o_lm(w["poses"], w["fixed"], w["points"], w["edge_pose"], w["edge_point"], w["obs"], max_iterations=20)
        pg, tg = ctx.ba_get_state(i, len(w["poses"]), len(w["points"]))
        st = stats[i]
        assert abs(st.chi2_initial - so["chi2_initial"]) <= 1e-10 * so["chi2_initial"], i
        assert abs(st.chi2_final - so["chi2_final"]) <= 1e-9 * so["chi2_final"], i
        assert _rel(pg, po) < 1e-6 and _rel(tg, pt) < 1e-6, i
        back = oracle.ba_linearize(pg, w["fixed"], tg, w["edge_pose"], w["edge_point"], w["obs"])
        assert abs(back["chi2"] - st.chi2_final) <= 1e-9 * st.chi2_final, i
        # the Hpp / bp blocks of EVERY free pose are those of the last linearisation point (poses beyond the ninth were not written before round 5)
    w = wins[0]
    ctx.ba_optimize(w["poses"], w["fixed"], w["points"], w["edge_pose"], w["edge_point"], w["obs"], iterations=3)
    assert ctx.ba_last_path() == (True, [])
    big = synth.ba_window(22, 600, seed=14)
    ctx.ba_optimize(big["poses"], big["fixed"], big["points"], big["edge_pose"], big["edge_point"], big["obs"], iterations=2)
    ok_, why = ctx.ba_last_path()
    assert not ok_ and any("free poses" in x for x in why), why
    ctx.close()


def test_ba_optimize_chi2_one_transfer(hip_lib):
    """ygz_hip_ba_optimize_chi2 (what ba::LocalBAG2O calls, BA.cpp:497-515): upload, resident LM, one more linearisation and the unpacking of its
    per-edge chi2 are queued back to back and statistics + state + chi2 return in one transfer.  Must equal the step-by-step calls bit for bit:
    ygz_hip_ba_optimize for the state and the statistics, ygz_hip_ba_linearize at that state for the chi2 -- also for a window that is uploaded
    into the slot of a LARGER one (the slot's allocation is kept) and for repeated (point, pose) edges (host loop)."""
    ctx = make_ctx(hip_lib, max_frames=1)
    for (K, P, seed) in ((8, 1500, 31), (5, 300, 32), (8, 1500, 31)):
        w = synth.ba_window(K, P, seed=seed)
        a = (w["poses"], w["fixed"], w["points"], w["edge_pose"], w["edge_point"], w["obs"])
        po, pt, st = ctx.ba_optimize(*a, iterations=12)
        assert ctx.ba_last_path()[0]
        po2, pt2, st2, chi = ctx.ba_optimize_chi2(*a, iterations=12)
        assert ctx.ba_last_path()[0]
        assert (st.iterations, st.lm_trials, st.chi2_initial, st.chi2_final) == (st2.iterations, st2.lm_trials, st2.chi2_initial, st2.chi2_final)
        assert po.tobytes() == po2.tobytes() and pt.tobytes() == pt2.tobytes()
        lin = ctx.ba_linearize(po, w["fixed"], pt, w["edge_pose"], w["edge_point"], w["obs"])
        assert lin["chi2_edge"].tobytes() == chi.tobytes()
        assert st2.chi2_final < 0.1 * st2.chi2_initial and chi.shape == (len(w["edge_pose"]),)
    # repeated (point, pose) pair: the host loop, through the same entry point
    w = synth.ba_window(4, 80, seed=33)
    ep = np.concatenate([w["edge_pose"], w["edge_pose"][:3]]); el = np.concatenate([w["edge_point"], w["edge_point"][:3]])
    ob = np.concatenate([w["obs"], w["obs"][:3] + 0.25])
    po, pt, st = ctx.ba_optimize(w["poses"], w["fixed"], w["points"], ep, el, ob, iterations=6)
    assert not ctx.ba_last_path()[0]
    po2, pt2, st2, chi = ctx.ba_optimize_chi2(w["poses"], w["fixed"], w["points"], ep, el, ob, iterations=6)
    assert not ctx.ba_last_path()[0] and po.tobytes() == po2.tobytes() and pt.tobytes() == pt2.tobytes()
    lin = ctx.ba_linearize(po, w["fixed"], pt, ep, el, ob)
    assert lin["chi2_edge"].tobytes() == chi.tobytes()
    ctx.close()


def test_ba_optimize_resident_team_size_invariance(hip_lib):
    """k_ba_lm_team gives a window to 1, 2, 4 or 8 cooperating workgroups depending on how many windows the launch holds; the points
    are reduced in 8 fixed parts whatever the team size, so the optimum, the trial sequence and the refined state must be BIT-identical
    (this is what keeps a sharded offline run equal to the unsharded one).  The same window solved alone (8 workgroups), among 10
    (4), among 20 (2) and among 70 (1)."""
    w = synth.ba_window(8, 1500, seed=21)
    others = synth.ba_window(4, 60, seed=22)
    ctx = make_ctx(hip_lib, max_frames=1)
    N = 70
    for i in range(N):
        u = w if i == 0 else others
        ctx.ba_upload(i, u["poses"], u["fixed"], u["points"], u["edge_pose"], u["edge_point"], u["obs"])
    ref = None
    for n_launch in (1, 10, 20, N):
        ctx.ba_set_state(0, w["poses"], w["points"])
        for i in range(1, n_launch):
            ctx.ba_set_state(i, others["poses"], others["points"])
        st = ctx.ba_optimize_resident(0, n_launch, iterations=20)[0]
        pg, tg = ctx.ba_get_state(0, len(w["poses"]), len(w["points"]))
        cur = (st.iterations, st.lm_trials, st.chi2_initial, st.chi2_final, st.lambda_final, pg.tobytes(), tg.tobytes())
        assert st.chi2_final < 0.1 * st.chi2_initial
        if ref is None:
            ref = cur
        assert cur == ref, n_launch
    # the same through the team budget (ygz_hip_ba_set_team_budget): 1 window with 32, 4 and 1 workgroups
    for budget in (0, 4, 1):
        ctx.ba_set_team_budget(budget)
        ctx.ba_set_state(0, w["poses"], w["points"])
        st = ctx.ba_optimize_resident(0, 1, iterations=20)[0]
        pg, tg = ctx.ba_get_state(0, len(w["poses"]), len(w["points"]))
        assert (st.iterations, st.lm_trials, st.chi2_initial, st.chi2_final, st.lambda_final, pg.tobytes(), tg.tobytes()) == ref, budget
    ctx.ba_set_team_budget(0)
    # ... and through the team's PLACEMENT (ygz_hip_ba_set_team_placement, round 5): compact = one XCD per window, the barrier without the L2
    # write-back; spread = four CUs of every XCD, the full barrier -- 1 window (team of 32) and 10 windows (teams of 8)
    for spread in (1, 0):
        ctx.ba_set_team_placement(spread)
        for n_launch in (1, 10):
            ctx.ba_set_state(0, w["poses"], w["points"])
            for i in range(1, n_launch):
                ctx.ba_set_state(i, others["poses"], others["points"])
            st = ctx.ba_optimize_resident(0, n_launch, iterations=20)[0]
            pg, tg = ctx.ba_get_state(0, len(w["poses"]), len(w["points"]))
            assert (st.iterations, st.lm_trials, st.chi2_initial, st.chi2_final, st.lambda_final, pg.tobytes(), tg.tobytes()) == ref, (spread, n_launch)
    ctx.close()


# ------------------------------------------------------------------------------------- M4 / M5: BoW-guided matching
def test_bow_transform_and_guided_matching(hip_lib, oracle):
    """Frame::ComputeBoW (DBoW3 tree descent), Matcher::SearchByBoW and Matcher::SearchForTriangulation on extracted frames
    (slot form, all pairs in one launch) and on host arrays, against the oracle: every index bit-exact."""
    blob = fixtures.synthetic_vocabulary(k=10, L=4, seed=5)
    vo = oracle.vocab_parse(blob)
    imgs, poses, depths = _frames(3, 640, 480, seed=23, step=0.3)
    ctx = make_ctx(hip_lib, max_frames=3)
    assert ctx.vocab_load(blob) == (10, 4, vo.n_nodes, vo.n_words)
    for s in range(3):
        ctx.upload_gray(s, imgs[s])
    ctx.build_pyramid(0, 3); ctx.detect(0, 3)
    kps = [ctx.get_keypoints(s) for s in range(3)]
    for levelsup in (2, 4):                                  # Frame.cpp:199 uses 4 (with L = 4: the root); 2 gives 100 nodes
        ctx.compute_bow(0, 3, levelsup)
        nodes = []
        for s in range(3):
            w, wt, nd = ctx.get_bow(s)
            ow, owt, ond, _, _ = oracle.bow_transform(vo, kps[s]["desc"], levelsup)
            assert np.array_equal(w, ow) and np.array_equal(wt, owt) and np.array_equal(nd, ond), (s, levelsup)
            nodes.append(nd)
        pairs1, pairs2 = [1, 2, 0], [0, 1, 0]
        m, cnt = ctx.search_by_bow_slots(pairs1, pairs2, mode=0, th_low=65, knn_ratio=0.7)
        for p in range(3):
            a, b = pairs1[p], pairs2[p]
            om, oc = oracle.search_by_bow(kps[a]["desc"], nodes[a], kps[b]["desc"], nodes[b], 65, 0.7)
            assert cnt[p] == oc and np.array_equal(m[p, :len(om)], om), (p, levelsup)
        # Matcher::Options::checkOrientation: the rotation histogram of the matches, its three maxima and the count that is left (slot form:
        # the resident keypoints' angles; host form: Feature::_angle as doubles) -- identical to the oracle's
        kept, hist, ind = ctx.bow_orientation_slots(pairs1, pairs2, m)
        for p in range(3):
            a, b = pairs1[p], pairs2[p]
            oc, oh, oi = oracle.bow_orientation(kps[a]["angle"], kps[b]["angle"], m[p, :len(nodes[a])])
            assert kept[p] == oc and np.array_equal(hist[p], oh) and np.array_equal(ind[p], oi), (p, levelsup)
            assert 0 < kept[p] <= cnt[p]
            hk, hh, hi = ctx.bow_orientation(kps[a]["angle"], kps[b]["angle"], m[p, :len(nodes[a])])
            assert (hk, list(hh), list(hi)) == (oc, list(oh), list(oi))
        assert kept[2] == cnt[2] and list(ind[2]) == [0, -1, -1]       # a frame against itself: every rotation is 0
        self_m = m[2, :len(nodes[0])]                          # a frame against itself: whatever matches, matches itself
        assert cnt[2] > 0.9 * (nodes[0] >= 0).sum() and np.all(self_m[self_m >= 0] == np.nonzero(self_m >= 0)[0])
        # SearchForTriangulation with E12 of the true relative pose (E = [t]x R, normalised coordinates)
        Es = []
        for p in range(3):
            T12 = oracle.se3_mul(poses[pairs2[p]], oracle.se3_inv(poses[pairs1[p]]))         # x2 = R x1 + t
            R = synth.quat_to_R(T12[:4]); t = T12[4:]
            tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
            Es.append((tx @ R).T if False else (tx @ R))
        # CheckDistEpipolarLine multiplies pt1 from the left: line = pt1^T E12, so E12 maps frame-1 points to frame-2 lines
        E12 = np.stack([e.T for e in Es])
        mt, ct = ctx.search_by_bow_slots(pairs1, pairs2, mode=1, E12=E12, th_low=65, epipolar_dsqr=1e-4)
        for p in range(2):
            a, b = pairs1[p], pairs2[p]
            om, oc = oracle.search_for_triangulation(kps[a]["desc"], nodes[a], kps[a]["px"], kps[b]["desc"], nodes[b], kps[b]["px"], E12[p], 65, 1e-4)
            assert ct[p] == oc and np.array_equal(mt[p, :len(om)], om), (p, levelsup)
            assert oc > 20
    # host-array forms (what the class surface calls), incl. ragged / empty sets
    d1, d2 = kps[1]["desc"], kps[0]["desc"]
    w, wt, nd1 = ctx.bow_transform(d1, 2); nd2 = ctx.bow_transform(d2, 2)[2]
    assert np.array_equal(nd1, oracle.bow_transform(vo, d1, 2)[2])
    m, c = ctx.search_by_bow(d1, nd1, d2, nd2)
    om, oc = oracle.search_by_bow(d1, nd1, d2, nd2)
    assert c == oc and np.array_equal(m, om)
    m, c = ctx.search_by_bow(d1[:300], nd1[:300], d2[:7], nd2[:7], mode=1, px1=kps[1]["px"][:300], px2=kps[0]["px"][:7], E12=E12[0])
    om, oc = oracle.search_for_triangulation(d1[:300], nd1[:300], kps[1]["px"][:300], d2[:7], nd2[:7], kps[0]["px"][:7], E12[0])
    assert c == oc and np.array_equal(m, om)
    m, c = ctx.search_by_bow(d1[:5], nd1[:5], np.zeros((0, 32), np.uint8), np.zeros(0, np.int32))
    assert c == 0 and np.all(m == -1)
    ctx.close()


def test_depth_from_triangulation_batch(hip_lib, oracle):
    """cvutils::DepthFromTriangulation (CVUtils.h:18-38) for a batch of ray pairs: depths within 1e-12 relative, same accept
    flags (incl. near-parallel rays rejected by the determinant test); exact geometry recovers the true depths."""
    rng = np.random.default_rng(4)
    T21 = oracle.se3_exp(np.array([0.3, -0.1, 0.05, 0.02, -0.03, 0.01]))          # search <- ref
    n = 2000
    p1 = np.stack([rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), rng.uniform(2, 8, n)], 1)
    p2 = np.array([oracle.se3_act(T21, p) for p in p1])
    f1, f2 = p1 / p1[:, 2:3], p2 / p2[:, 2:3]
    f2[::50] = (synth.quat_to_R(T21[:4]) @ f1[::50].T).T * (1 + 1e-9)           # parallel rays: det ~ 0
    ctx = make_ctx(hip_lib, max_frames=1)
    d1, d2, ok = ctx.depth_from_triangulation(T21, f1, f2)
    o1, o2, ook = oracle.depth_from_triangulation(T21, f1, f2)
    assert np.array_equal(ok, ook) and ok.sum() == n - len(f1[::50])
    m = ok == 1
    assert np.allclose(d1[m], o1[m], rtol=1e-12) and np.allclose(d2[m], o2[m], rtol=1e-12)
    assert np.allclose(d1[m], p1[m, 2], rtol=1e-9) and np.allclose(d2[m], p2[m, 2], rtol=1e-9)
    assert ctx.depth_from_triangulation(T21, np.zeros((0, 3)), np.zeros((0, 3)))[2].size == 0
    ctx.close()


# ------------------------------------------------------------------------------------- degenerate inputs through the ABI
def test_degenerate_inputs(hip_lib, oracle):
    """empty / minimal inputs of every batched entry point: status codes instead of crashes, results equal to the oracle's"""
    ctx = make_ctx(hip_lib, max_frames=2)
    black = np.zeros((480, 640), np.uint8)
    for s in range(2):
        ctx.upload_gray(s, black)
    ctx.build_pyramid(0, 2); ctx.detect(0, 2)
    assert len(ctx.get_keypoints(0)["level"]) == 0 and len(oracle.detect(oracle.pyramid(black, 3))) == 0      # no corner in a flat image
    idx, dist = ctx.hamming_match(np.zeros((0, 32), np.uint8), np.zeros((0, 32), np.uint8))
    assert len(idx) == 0
    # tracking stages on a pair without keypoints
    I7 = np.array([[0, 0, 0, 1, 0, 0, 0.0]])
    ctx.track_begin([1], [0], I7, I7, predict=False)
    ctx.track_klt(); ctx.track_direct(); ctx.track_sparse_align()
    out, st, err = ctx.track_get_klt(0)
    assert len(st) == 0
    nm, T, iters = ctx.track_get_pose(0)
    assert nm == 0 and np.allclose(T, I7[0])
    # BA window without a single edge, one pose, one point
    g = ctx.ba_linearize(np.zeros((1, 6)), np.zeros(1, np.uint8), np.array([[0.0, 0.0, 3.0]]), np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((0, 2)))
    assert g["chi2"] == 0.0 and not g["Hpp"].any() and not g["Hll"].any()
    po, pt, stt = ctx.ba_optimize(np.zeros((1, 6)), np.zeros(1, np.uint8), np.array([[0.0, 0.0, 3.0]]), np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((0, 2)))
    assert np.array_equal(pt, [[0.0, 0.0, 3.0]]) and stt.chi2_final == 0.0
    # pose-only BA with no frame / a frame without features; BoW without a vocabulary is a state error, not a crash
    p, bad, dep, inl, rounds = ctx.optimize_pose_only(np.array([0, 0], np.int32), np.zeros((0, 2)), np.zeros((0, 3)), np.zeros((1, 6)))
    assert inl[0] == 0 and rounds[0] == 1
    with pytest.raises(hip_lib.YgzHipError):
        ctx.bow_transform(np.zeros((4, 32), np.uint8))
    ctx.close()
