"""CPU checks of the ceres-side oracle rows (oracle/ceres_ba.c: SURVEY 8a B3, B6, B7) and of the two NLLS drivers.

ceres and g2o are absent from the image and the reference holds no asserting test for these paths ("parity unpinned"):
what can be checked is mathematical -- Jets against central differences, block assembly against numpy, the drivers on
the zero-noise known-answer fixture of test/test_local_ba.cpp:9-37 -- plus the control flow quirks of
ba::OptimizeCurrentPoseOnly as written (BA.cpp:188-264)."""
import numpy as np
import pytest
import fixtures

from ygz_slam_amd import synth


def _fd(f, x, h=1e-6):
    J = []
    for k in range(len(x)):
        d = np.zeros(len(x)); d[k] = h
        J.append((f(x + d) - f(x - d)) / (2 * h))
    return np.stack(J, axis=1)


@pytest.mark.parametrize("aa", [(0.3, -0.2, 0.5), (1e-3, 2e-3, -1e-3), (0.0, 0.0, 0.0), (1e-9, 0.0, 0.0), (2.5, 0.4, -1.0)])
def test_ceres_edge_jets_against_central_differences(oracle, aa):
    pose = np.array([0.1, -0.05, 0.2, *aa])
    pt = np.array([0.4, -0.3, 3.0])
    ob = np.array([0.11, -0.07])
    r, Jp, Jx, z = oracle.ceres_edge(pose, pt, ob)
    # residual against an independent Rodrigues rotation
    th = np.linalg.norm(aa)
    K = np.array([[0, -aa[2], aa[1]], [aa[2], 0, -aa[0]], [-aa[1], aa[0], 0]])
    R = np.eye(3) + K if th * th <= np.finfo(float).eps else np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K
    pc = R @ pt + pose[:3]
    assert np.allclose(r, ob - pc[:2] / pc[2], atol=1e-14) and abs(z - pc[2]) < 1e-14
    assert np.allclose(oracle.ceres_rotate_point(aa, pt), R @ pt, atol=1e-14)
    if th * th > np.finfo(float).eps or th == 0.0:
        assert np.allclose(_fd(lambda x: oracle.ceres_edge(x, pt, ob)[0], pose), Jx, rtol=1e-6, atol=1e-8)
    assert np.allclose(_fd(lambda x: oracle.ceres_edge(pose, x, ob)[0], pt), Jp, rtol=1e-6, atol=1e-8)
    if th == 0.0:       # first-order branch: d(aa x p)/d aa = -[p]x
        A = np.array([[1 / pc[2], 0, -pc[0] / pc[2] ** 2], [0, 1 / pc[2], -pc[1] / pc[2] ** 2]])
        Px = np.array([[0, -pt[2], pt[1]], [pt[2], 0, -pt[0]], [-pt[1], pt[0], 0]])
        assert np.allclose(Jx[:, 3:], A @ Px, atol=1e-14) and np.allclose(Jx[:, :3], -A, atol=1e-14)


def test_ceres_residual_equals_legacy_normalised_plane_residual(oracle):
    """same SE3, two parametrisations: [t; aa] with T = (exp(aa), t) vs the legacy g2o edge's exp([upsilon; omega])"""
    rng = np.random.default_rng(2)
    for _ in range(5):
        om, t = rng.normal(0, 0.3, 3), rng.normal(0, 0.2, 3)
        pt = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(2, 5)])
        ob = rng.normal(0, 0.1, 2)
        r_c = oracle.ceres_edge(np.concatenate([t, om]), pt, ob)[0]
        r_g = oracle.ba_edge_norm(np.concatenate([synth.se3_log_t(om, t), om]), pt, ob)[0]
        assert np.allclose(r_c, r_g, atol=1e-13)


def test_ceres_linearize_blocks_flags_and_loss(oracle):
    fx = fixtures.ba_to_ceres(fixtures.ba_fixture_test_local_ba(noise=True))
    E, K, P = len(fx["obs_n"]), len(fx["poses"]), len(fx["points"])
    rng = np.random.default_rng(0)
    huber = np.where(rng.random(E) < 0.5, 0.002, 0.0)
    enable = (rng.random(E) < 0.9).astype(np.uint8)
    pfix = np.zeros(P, np.uint8); pfix[3] = 1
    r = oracle.ceres_linearize(fx["poses"], fx["fixed"], fx["points"], fx["edge_pose"], fx["edge_point"], fx["obs_n"],
                               point_fixed=pfix, edge_huber=huber, edge_enable=enable)
    assert r["rc"] == 0
    Hpp, bp, Hll, bl = np.zeros((K, 6, 6)), np.zeros((K, 6)), np.zeros((P, 3, 3)), np.zeros((P, 3))
    cost, n_rob = 0.0, 0
    for e in range(E):
        if not enable[e]:
            assert not r["Hpl"][e].any() and not r["Jx"][e].any() and not r["res"][e].any()
            continue
        ip, il = fx["edge_pose"][e], fx["edge_point"][e]
        res, Jp, Jx, _ = oracle.ceres_edge(fx["poses"][ip], fx["points"][il], fx["obs_n"][e])
        s = res @ res
        w = 1.0
        if huber[e] > 0 and s > huber[e] ** 2:
            w = huber[e] / np.sqrt(s); cost += 0.5 * (2 * huber[e] * np.sqrt(s) - huber[e] ** 2); n_rob += 1
        else:
            cost += 0.5 * s
        if pfix[il]: Jp = Jp * 0
        if fx["fixed"][ip]: Jx = Jx * 0
        Hll[il] += w * Jp.T @ Jp; bl[il] -= w * Jp.T @ res
        Hpp[ip] += w * Jx.T @ Jx; bp[ip] -= w * Jx.T @ res
        assert np.allclose(r["Hpl"][e], w * Jx.T @ Jp, rtol=1e-12, atol=1e-12)
    assert n_rob > 5
    assert np.isclose(r["cost"], cost, rtol=1e-13)
    for a, b in ((Hpp, r["Hpp"]), (bp, r["bp"]), (Hll, r["Hll"]), (bl, r["bl"])):
        assert np.allclose(a, b, rtol=1e-12, atol=1e-12)
    assert not r["Hll"][3].any() and not r["Hpp"][0].any()
    # PoseOnly functor: evaluation fails behind the camera
    pts = fx["points"].copy(); pts[0, 2] = -5.0
    bad = oracle.ceres_linearize(fx["poses"], fx["fixed"], pts, fx["edge_pose"], fx["edge_point"], fx["obs_n"], fail_behind=True)
    assert bad["rc"] == -1


def test_schur_solve_against_dense(oracle):
    import ctypes as C
    from oracle.pyoracle import _f64, _u8, _p
    fx = fixtures.ba_to_ceres(fixtures.ba_fixture_test_local_ba(noise=True))
    r = oracle.ceres_linearize(fx["poses"], fx["fixed"], fx["points"], fx["edge_pose"], fx["edge_point"], fx["obs_n"])
    K, P, E = len(fx["poses"]), len(fx["points"]), len(fx["obs_n"])
    rng = np.random.default_rng(1)
    dp, dl = rng.uniform(0.1, 1, (K, 6)), rng.uniform(0.1, 1, (P, 3))
    pfree = (1 - fx["fixed"]).astype(np.uint8); lfree = np.ones(P, np.uint8); lfree[5] = 0
    xp, xl = np.empty((K, 6)), np.empty((P, 3))
    ep, el = np.ascontiguousarray(fx["edge_pose"], np.int32), np.ascontiguousarray(fx["edge_point"], np.int32)
    ok = oracle.lib.yo_ba_schur_solve(K, P, E, _p(ep, C.c_int32), _p(el, C.c_int32), _u8(pfree), _u8(lfree), _f64(r["Hpp"]),
                                      _f64(r["Hll"]), _f64(r["Hpl"]), _f64(r["bp"]), _f64(r["bl"]), _f64(dp), _f64(dl),
                                      _f64(xp), _f64(xl))
    assert ok == 1
    n = 6 * K + 3 * P
    H, b = np.zeros((n, n)), np.zeros(n)
    for k in range(K):
        H[6 * k:6 * k + 6, 6 * k:6 * k + 6] = r["Hpp"][k] + np.diag(dp[k]); b[6 * k:6 * k + 6] = r["bp"][k]
    for l in range(P):
        o = 6 * K + 3 * l
        H[o:o + 3, o:o + 3] = r["Hll"][l] + np.diag(dl[l]); b[o:o + 3] = r["bl"][l]
    for e in range(E):
        k, l = ep[e], el[e]
        if pfree[k] and lfree[l]:
            H[6 * k:6 * k + 6, 6 * K + 3 * l:6 * K + 3 * l + 3] += r["Hpl"][e]
            H[6 * K + 3 * l:6 * K + 3 * l + 3, 6 * k:6 * k + 6] += r["Hpl"][e].T
    keep = np.concatenate([np.repeat(pfree, 6), np.repeat(lfree, 3)]).astype(bool)
    x = np.zeros(n); x[keep] = np.linalg.solve(H[np.ix_(keep, keep)], b[keep])
    assert np.allclose(np.concatenate([xp.ravel(), xl.ravel()]), x, rtol=1e-9, atol=1e-12)


def test_ceres_solve_local_ba_zero_noise_converges(oracle):
    """ba::LocalBA (BA.cpp:324-384) on the reference's own fixture: exact observations, perturbed estimate -> cost 0."""
    fx = fixtures.ba_fixture_test_local_ba(noise=True)
    exact = fixtures.ba_fixture_test_local_ba(noise=False)
    fx["obs"] = exact["obs"]
    c = fixtures.ba_to_ceres(fx)
    poses, points, s = oracle.ceres_solve(c["poses"], c["fixed"], c["points"], c["edge_pose"], c["edge_point"], c["obs_n"])
    assert s["rc"] == 0 and s["initial_cost"] > 1e-3
    assert s["final_cost"] < 1e-12 * s["initial_cost"] + 1e-16
    assert s["successful_steps"] >= 3 and s["iterations"] <= 50
    assert np.array_equal(poses[0], c["poses"][0])                      # keyframe 0 is a constant (PointOnly functor)
    # gauge: scale is free; every reprojection must be exact
    r = oracle.ceres_linearize(poses, c["fixed"], points, c["edge_pose"], c["edge_point"], c["obs_n"])
    assert np.abs(r["res"]).max() < 1e-7


def test_ceres_solve_noisy_window_decreases_and_terminates(oracle):
    c = fixtures.ba_to_ceres(synth.ba_window(K=6, P=300, seed=5))
    poses, points, s = oracle.ceres_solve(c["poses"], c["fixed"], c["points"], c["edge_pose"], c["edge_point"], c["obs_n"])
    assert s["rc"] == 0 and s["final_cost"] < 0.05 * s["initial_cost"]
    assert s["termination"] in (0, 1, 2)                                  # a convergence criterion, not the iteration cap
    # first-order optimality of the returned point
    r = oracle.ceres_linearize(poses, c["fixed"], points, c["edge_pose"], c["edge_point"], c["obs_n"])
    assert np.abs(r["bp"]).max() < 1e-3 * np.abs(oracle.ceres_linearize(c["poses"], c["fixed"], c["points"], c["edge_pose"],
                                                                       c["edge_point"], c["obs_n"])["bp"]).max()


def test_ceres_dogleg_strategy(oracle):
    """options.trust_region_strategy_type = DOGLEG (what ba::TwoViewBACeres asks for, BA.cpp:60): the restated DoglegStrategy against the
    Levenberg-Marquardt one on the same problems -- another path through the same cost landscape: both converge (no iteration cap), to the same cost
    and a stationary point; on the zero-noise fixture to cost 0; with a first full Gauss-Newton step when it lies inside the initial radius of 1e4."""
    fx = fixtures.ba_fixture_test_local_ba(noise=True)
    fx["obs"] = fixtures.ba_fixture_test_local_ba(noise=False)["obs"]
    c = fixtures.ba_to_ceres(fx)
    dog = oracle.ceres_options(trust_region_strategy=1)
    pd, xd, sd = oracle.ceres_solve(c["poses"], c["fixed"], c["points"], c["edge_pose"], c["edge_point"], c["obs_n"], options=dog)
    assert sd["rc"] == 0 and sd["final_cost"] < 1e-12 * sd["initial_cost"] + 1e-16 and sd["successful_steps"] >= 3 and sd["iterations"] <= 50
    assert np.array_equal(pd[0], c["poses"][0])
    for w in (synth.ba_window(K=6, P=300, seed=5), synth.ba_window(K=4, P=60, seed=9), fixtures.ba_fixture_test_local_ba(noise=True, seed=5)):
        c = fixtures.ba_to_ceres(w)
        args = (c["poses"], c["fixed"], c["points"], c["edge_pose"], c["edge_point"], c["obs_n"])
        pl, xl, sl = oracle.ceres_solve(*args)
        pd, xd, sd = oracle.ceres_solve(*args, options=dog)
        assert sd["rc"] == 0 and sd["termination"] in (0, 1, 2) and sd["successful_steps"] >= 2
        assert abs(sd["final_cost"] - sl["final_cost"]) <= 1e-4 * sl["final_cost"] + 1e-14, (sd["final_cost"], sl["final_cost"])
        r0 = oracle.ceres_linearize(*args)
        r = oracle.ceres_linearize(pd, c["fixed"], xd, c["edge_pose"], c["edge_point"], c["obs_n"])
        assert np.abs(r["bp"]).max() < 1e-3 * np.abs(r0["bp"]).max()                  # first-order optimality
        assert sd["final_radius"] > 0 and np.isfinite(sd["final_radius"])
    # robustified edges and constant points (the shape of TwoViewBACeres: HuberLoss(0.1) on some residual blocks)
    w = synth.ba_window(K=5, P=120, seed=2)
    c = fixtures.ba_to_ceres(w)
    rng = np.random.default_rng(1)
    huber = np.where(rng.random(len(c["obs_n"])) < 0.3, 0.1, 0.0)
    pfix = (rng.random(len(c["points"])) < 0.1).astype(np.uint8)
    kw = dict(point_fixed=pfix, edge_huber=huber)
    pl, xl, sl = oracle.ceres_solve(c["poses"], c["fixed"], c["points"], c["edge_pose"], c["edge_point"], c["obs_n"], **kw)
    pd, xd, sd = oracle.ceres_solve(c["poses"], c["fixed"], c["points"], c["edge_pose"], c["edge_point"], c["obs_n"], options=dog, **kw)
    assert sd["rc"] == 0 and abs(sd["final_cost"] - sl["final_cost"]) <= 1e-4 * sl["final_cost"]
    assert np.array_equal(xd[pfix.astype(bool)], c["points"][pfix.astype(bool)])


def test_g2o_lm_restatement(oracle):
    fx = fixtures.ba_fixture_test_local_ba(noise=True)
    fx["obs"] = fixtures.ba_fixture_test_local_ba(noise=False)["obs"]
    poses, points, st = oracle.g2o_lm(fx["poses"], fx["fixed"], fx["points"], fx["edge_pose"], fx["edge_point"], fx["obs"],
                                      max_iterations=20)
    assert st["chi2_initial"] > 1.0 and st["chi2_final"] < 1e-8 * st["chi2_initial"]
    assert st["iterations"] <= 20 and st["lm_trials"] >= st["iterations"]
    w = synth.ba_window(K=6, P=300, seed=5)
    poses, points, st = oracle.g2o_lm(w["poses"], w["fixed"], w["points"], w["edge_pose"], w["edge_point"], w["obs"])
    assert st["chi2_final"] < 0.1 * st["chi2_initial"]
    r = oracle.ba_linearize(poses, w["fixed"], points, w["edge_pose"], w["edge_point"], w["obs"])
    assert np.isclose(r["chi2"], st["chi2_final"], rtol=1e-12)


def test_optimize_current_pose_only(oracle):
    f = fixtures.pose_only_fixture(n=400, seed=3)
    pose, bad, depth, inl, rounds = oracle.optimize_current_pose_only(f["entry"], f["px"], f["pw"])
    assert rounds == 4 and inl == int((bad == 0).sum())
    assert np.abs(pose[:3] - f["true"][:3]).max() < 5e-3 and np.abs(pose[3:] - f["true"][3:]).max() < 2e-3
    # the gross outliers are flagged, the inliers kept (threshold chi2Mono = 5.991 px^2, BA.cpp:195)
    assert bad[f["outlier"]].all() and bad[~f["outlier"]].mean() < 0.05
    assert np.all(np.isfinite(depth[bad == 0])) and np.all(depth[bad == 0] > 1.5)
    # control flow as written (BA.cpp:227-255), replayed with the generic solver: every round restarts from the ENTRY pose,
    # the inlier test of a round runs with the _TCW committed by the previous round
    n = len(f["px"])
    obs_n = np.stack([(f["px"][:, 0] - synth.CX) / synth.FX, (f["px"][:, 1] - synth.CY) / synth.FY], axis=1)
    enable, tcw = np.ones(n, np.uint8), f["entry"].copy()
    for _ in range(4):
        sol, _, _ = oracle.ceres_solve(f["entry"][None], None, f["pw"], np.zeros(n, np.int32), np.arange(n, dtype=np.int32), obs_n,
                                       point_fixed=np.ones(n, np.uint8), edge_enable=enable, fail_behind=True)
        T = np.concatenate([synth.se3_exp(np.concatenate([np.zeros(3), tcw[3:]]))[:4], tcw[:3]])
        uv, z = synth.project(T, f["pw"])
        e2 = ((uv - f["px"]) ** 2).sum(axis=1)
        enable = (e2 <= np.float32(5.991)).astype(np.uint8)
        tcw = sol[0].copy()
    assert np.array_equal(pose, tcw) and np.array_equal(bad, 1 - enable)
    # fewer than 10 inliers in round 1 -> break before the pose is committed: _TCW unchanged
    g = fixtures.pose_only_fixture(n=12, seed=4, outlier_frac=0.0)
    far = g["entry"] + np.array([0.5, 0.5, 0, 0, 0, 0])
    pose2, bad2, _, inl2, rounds2 = oracle.optimize_current_pose_only(far, g["px"], g["pw"])
    assert rounds2 == 1 and inl2 < 10 and np.array_equal(pose2, far)
    # empty frame
    pose3, bad3, _, inl3, rounds3 = oracle.optimize_current_pose_only(f["entry"], np.zeros((0, 2)), np.zeros((0, 3)))
    assert inl3 == 0 and rounds3 == 1 and np.array_equal(pose3, f["entry"])
