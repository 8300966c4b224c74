"""CPU tests of the oracle's map-building restatements (oracle/mapping.c): the triangulation loop of LocalMapping::CreateNewMapPoints
(src/Module/LocalMapping.cpp:416-495) and the legacy SVO depth filter (src/optimizer.cpp:537-735, src/utils.cpp:330-661), checked
against the ground truth of a rendered sequence -- properties the reference's own code must have, independent of any GPU run."""
import numpy as np
from ygz_slam_amd import synth, offline


def _seq_and_keypoints(oracle, n=4, step=0.5, seed=6, w=640, h=480):
    seq = synth.Sequence(n, w, h, seed=seed, step=step)
    lv = [oracle.pyramid(oracle.bgr2gray(seq.frame(f)), 3) for f in range(n)]
    k0 = oracle.detect(lv[0], oracle.default_params(w, h, 3))
    px = np.stack([k0["px"], k0["py"]], 1).astype(np.float64)
    z = seq.depth(0)[px[:, 1].astype(int), px[:, 0].astype(int)]
    return seq, lv, k0, px, z


def test_create_map_points_recovers_depth(oracle):
    seq, lv, k0, px1, z = _seq_and_keypoints(oracle, n=6, step=0.6, seed=9)      # frames 0 and 5: ~40 px of image motion (the parallax test compares
                                                                                  # camera-frame rays, i.e. it measures image motion)
    cam = oracle.camera()
    fx, fy, cx, cy = float(cam.fx), float(cam.fy), float(cam.cx), float(cam.cy)
    T1, T2 = seq.poses[0], seq.poses[5]
    pc1 = np.stack([(px1[:, 0] - cx) * z / fx, (px1[:, 1] - cy) * z / fy, z], 1)
    pc2 = offline.se3_act(offline.se3_mul(T2, offline.se3_inv(T1)), pc1)
    px2 = np.stack([fx * pc2[:, 0] / pc2[:, 2] + cx, fy * pc2[:, 1] / pc2[:, 2] + cy], 1) + np.random.default_rng(0).normal(0, 0.5, (len(z), 2))
    r = oracle.create_map_points(lv[0], T1, lv[5], T2, px1, k0["level"], px2)
    ok = r["code"] == 0
    assert ok.sum() > 0.4 * len(z) and r["created"] == ok.sum()
    assert np.median(np.abs(r["depth1"][ok] - z[ok]) / z[ok]) < 0.05
    # the map point is the back-projection of feature 1 at the triangulated depth
    pw = offline.se3_act(offline.se3_inv(T1), pc1[ok] / z[ok, None] * r["depth1"][ok, None])
    assert np.allclose(r["pos_world"][ok], pw, atol=1e-9)
    # refined pixels moved towards the true projection
    true2 = px2[ok] * 0 + np.stack([fx * pc2[ok, 0] / pc2[ok, 2] + cx, fy * pc2[ok, 1] / pc2[ok, 2] + cy], 1)
    assert np.median(np.linalg.norm(r["px2"][ok] - true2, axis=1)) < np.median(np.linalg.norm(px2[ok] - true2, axis=1))
    # identical rays are rejected by the parallax test
    same = oracle.create_map_points(lv[0], T1, lv[0], T1, px1[:20], k0["level"][:20], px1[:20])
    assert np.all(same["code"] == 1)


def test_depth_filter_converges_towards_the_rendered_depth(oracle):
    seq, lv, k0, px, z = _seq_and_keypoints(oracle, n=6, step=0.45, seed=4)
    n = len(z)
    rng = np.random.default_rng(1)
    zr = np.full(n, 1.0 / (float(z.min()) * 0.6), np.float32)
    seeds = dict(kp=px.astype(np.float32), octave=k0["level"], ref=np.zeros(n, np.int32), frame_id=np.zeros(n, np.uint64),
                 a=np.full(n, 10, np.float32), b=np.full(n, 10, np.float32), mu=(1.0 / (z * rng.uniform(0.75, 1.3, n))).astype(np.float32),
                 z_range=zr, sigma2=(zr * zr / 36).astype(np.float32))
    err0 = np.abs(1.0 / seeds["mu"] - z)
    s0 = seeds["sigma2"].copy()
    upd = np.zeros(n, int)
    for f in (1, 2, 3, 4, 5):
        r = oracle.depth_filter_update(lv[f], seq.poses[f], [lv[0]], [seq.poses[0]], seeds, batch_counter=1, conv_thresh=1e9)   # never erase: keep indices
        assert not np.isin(r["state"], (4, 5, 6)).any()
        for k in ("a", "b", "mu", "sigma2"):
            seeds[k] = r[k]
        upd += r["state"] == 0
        m = r["state"] == 0
        assert np.median(np.abs(r["z"][m] - z[m]) / z[m]) < 0.05                  # the epipolar match triangulates to the right depth
    often = upd >= 3
    assert often.sum() > 0.5 * n
    assert np.median(np.abs(1.0 / seeds["mu"][often] - z[often])) < 0.6 * np.median(err0[often])     # Beta(10, 10) prior: the filter trusts slowly
    assert np.all(seeds["sigma2"][often] < s0[often])                             # the variance of a repeatedly updated seed shrinks
    # a seed older than max_n_kfs keyframes is erased; an unsigned frame id larger than the batch counter wraps and is erased too
    old = dict(seeds); old["frame_id"] = np.full(n, 7, np.uint64)
    r = oracle.depth_filter_update(lv[1], seq.poses[1], [lv[0]], [seq.poses[0]], old, batch_counter=20)
    assert np.all(r["state"] == 4)
    r = oracle.depth_filter_update(lv[1], seq.poses[1], [lv[0]], [seq.poses[0]], old, batch_counter=1)
    assert np.all(r["state"] == 4)


def test_shared_exponential_against_libm_and_the_seed_update(oracle):
    """include/ygz_exp.h (the exponential DepthFilter::UpdateSeed uses on the host AND on the device, so that both round alike) against what the
    unmodified reference calls, glibc's expf: (i) never more than 1 ulp apart over the range the update uses, and equal to the correctly rounded
    float exponential (double exp rounded once) almost everywhere; (ii) the drift of a, b, mu, sigma2 when one update step runs with expf
    instead: recorded bounds.  (iii) both builds forbid FMA contraction, which the bit-identity of the shared form rests on."""
    import ctypes as C, os, re
    lib = oracle.lib
    lib.yo_expf_shared.restype = C.c_float; lib.yo_expf_shared.argtypes = [C.c_float]
    libm = C.CDLL("libm.so.6"); libm.expf.restype = C.c_float; libm.expf.argtypes = [C.c_float]
    rng = np.random.default_rng(7)
    xs = np.concatenate([-rng.uniform(0, 100, 60000), -10.0 ** rng.uniform(-8, 2, 40000), [0.0, -0.0, -1e-30, -87.3, -103.9]]).astype(np.float32)
    ours = np.array([lib.yo_expf_shared(float(x)) for x in xs], np.float32)
    theirs = np.array([libm.expf(float(x)) for x in xs], np.float32)
    cr = np.exp(xs.astype(np.float64)).astype(np.float32)                      # correctly rounded unless the double lands on a rounding boundary
    ulp = np.abs(ours.view(np.int32).astype(np.int64) - theirs.view(np.int32).astype(np.int64))
    assert ulp.max() <= 1, ulp.max()
    assert (ours == cr).mean() > 0.9999 and (ulp == 0).mean() > 0.99, ((ours == cr).mean(), (ulp == 0).mean())
    # (ii) one seed update with either exponential
    lib.yo_update_seed.argtypes = [C.c_float, C.c_float] + [C.POINTER(C.c_float)] * 3 + [C.c_float, C.POINTER(C.c_float)]
    worst = np.zeros(4)
    for _ in range(4000):
        mu0, s20 = float(rng.uniform(0.2, 2.0)), float(rng.uniform(1e-4, 0.2))
        x = float(mu0 + rng.normal(0, 2.0 * np.sqrt(s20))); tau2 = float(rng.uniform(1e-5, 1e-2)); zr = float(rng.uniform(1.0, 4.0))
        a0, b0 = float(rng.uniform(5, 40)), float(rng.uniform(5, 40))
        res = []
        for mode in (0, 1):
            lib.yo_set_exp_libm(mode)
            a, b, mu, s2 = C.c_float(a0), C.c_float(b0), C.c_float(mu0), C.c_float(s20)
            lib.yo_update_seed(x, tau2, C.byref(a), C.byref(b), C.byref(mu), zr, C.byref(s2))
            res.append(np.array([a.value, b.value, mu.value, s2.value], np.float64))
        lib.yo_set_exp_libm(0)
        if np.all(np.isfinite(res[0])) and np.all(np.isfinite(res[1])):
            worst = np.maximum(worst, np.abs(res[0] - res[1]) / np.maximum(np.abs(res[1]), 1e-30))
    # the Beta parameters feel one ulp of the pdf through a cancellation (e - f, f - e / f); mu and sigma2 do not
    assert worst[0] < 5e-3 and worst[1] < 5e-3 and worst[2] < 1e-5 and worst[3] < 1e-4, worst
    # (iii)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert "-ffp-contract=off" in re.search(r"^CFLAGS\s*\?=.*$", open(os.path.join(root, "oracle", "Makefile")).read(), re.M).group(0)
    assert "-ffp-contract=off" in re.search(r"^FLAGS\s*:=.*$", open(os.path.join(root, "ygz_slam_amd", "csrc", "Makefile")).read(), re.M).group(0)
