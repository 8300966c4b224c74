"""Fixtures that only the tests (and the probes under tools/) use: the transcribed inputs of the reference's synthetic test
(test/test_local_ba.cpp:9-37), its ceres-side re-parametrisation, pose-only BA frames, random descriptor sets and synthetic
vocabularies in DBoW3's binary format.  They build on the generators of ygz_slam_amd/synth.py (camera, SE3 helpers)."""
import numpy as np
from ygz_slam_amd.synth import (FX, FY, CX, CY, make_texture, trajectory, render, se3_exp, se3_log_t, project, quat_to_R, ba_window)  # noqa: F401


def frame_sequence(n, w=640, h=480, seed=1, noise_sigma=1.0, step=0.02):
    tex, margin = make_texture(seed, w, h)
    poses = trajectory(n, seed + 10, step)
    imgs, depths = [], []
    for i in range(n):
        im, d = render(tex, margin, poses[i], w, h, noise_sigma, seed * 1000 + i)
        imgs.append(im)
        depths.append(d)
    return np.stack(imgs), poses, np.stack(depths)


def random_descriptors(n, seed=42):
    return np.random.default_rng(seed).integers(0, 256, (n, 32), dtype=np.uint8)


TEST_LOCAL_BA_POSES = [  # test/test_local_ba.cpp:9-18  (omega, t)
    ((0, 0, 0), (0, 0, 0)), ((0.1, 0, 0), (0, 0, 0)), ((0, 0.1, 0), (0, 0, 0)), ((0, 0, 0.1), (0, 0, 0)),
    ((0, 0, 0), (0.1, 0, 0)), ((0, 0, 0), (0, 0.1, 0)), ((0, 0, 0), (0, 0, 0.1)), ((0, 0, 0), (0.1, 0.1, 0.1)),
]
TEST_LOCAL_BA_POINTS = [(x, y, z) for z in (2, 3, 4, 5) for (x, y) in ((0, 0), (0, 1), (1, 0), (1, 1))]  # :20-37


def ba_fixture_test_local_ba(noise=True, seed=7):
    """8 keyframes x 16 points x 128 observations, as test/test_local_ba.cpp:39-101 builds
    them (the reference draws its noise from cv::RNG, which is not reproducible here; the
    zero-noise variant is the closed-form known-answer case: residuals must vanish)."""
    rng = np.random.default_rng(seed)
    # keyframe_poses[i] = SE3(SO3::exp(omega), t): the translation is t itself (test_local_ba.cpp:9-18)
    true_poses = np.array([np.concatenate([se3_exp(np.concatenate([np.zeros(3), om]))[:4], np.asarray(t, float)])
                           for om, t in TEST_LOCAL_BA_POSES])
    pts = np.array(TEST_LOCAL_BA_POINTS, np.float64)
    # vertex estimate order is [omega; t] of log(T)  (BA.cpp:407-409)
    poses = np.array([np.concatenate([np.asarray(om, float), se3_log_t(np.asarray(om, float), np.asarray(t, float))])
                      for om, t in TEST_LOCAL_BA_POSES])
    ep, el, obs = [], [], []
    for i in range(len(pts)):
        for j in range(len(true_poses)):
            uv, _ = project(true_poses[j], pts[i:i + 1])
            ep.append(j)
            el.append(i)
            obs.append(uv[0])
    obs = np.array(obs)
    est_poses, est_pts = poses.copy(), pts.copy()
    if noise:
        est_poses[1:] += rng.normal(0, 0.1, est_poses[1:].shape)        # :58-64
        est_pts += rng.normal(0, 0.1, est_pts.shape)                    # :79-82
        obs = obs + rng.normal(0, 1.0, obs.shape)                       # :94
    fixed = np.zeros(len(poses), np.uint8)
    fixed[0] = 1                                                        # BA.cpp:404-405
    return dict(poses=est_poses, fixed=fixed, points=est_pts, edge_pose=np.array(ep, np.int32),
                edge_point=np.array(el, np.int32), obs=obs, true_poses=poses, true_points=pts)


def ba_to_ceres(fx):
    """The same window in the ceres-side parametrisation (BA.cpp:96-99,336-362): pose = [t; angle-axis] of T_cw,
    observation in normalised image coordinates (Camera::Pixel2Camera2D with the float intrinsics)."""
    def conv(poses):
        out = np.empty_like(poses)
        for k, p in enumerate(poses):
            T = se3_exp(np.concatenate([p[3:], p[:3]]))      # [omega; upsilon] -> Sophus [upsilon; omega]
            out[k, :3], out[k, 3:] = T[4:], p[:3]
        return out
    obs_n = np.stack([(fx["obs"][:, 0] - CX) / FX, (fx["obs"][:, 1] - CY) / FY], axis=1)
    d = dict(fx)
    d.update(poses=conv(fx["poses"]), obs_n=obs_n, true_poses=conv(fx["true_poses"]))
    return d


def pose_only_fixture(n=400, seed=3, outlier_frac=0.1, sigma_px=0.5, w=640, h=480):
    """One frame for ba::OptimizeCurrentPoseOnly: n map points seen at pixel noise sigma_px, a fraction of gross
    outliers, an entry pose a few millimetres off (the first inlier test runs with the ENTRY pose, BA.cpp:233)."""
    rng = np.random.default_rng(seed)
    true = np.concatenate([rng.normal(0, 0.05, 3), rng.normal(0, 0.03, 3)])           # [t; aa]
    T = np.concatenate([se3_exp(np.concatenate([np.zeros(3), true[3:]]))[:4], true[:3]])
    pw = np.stack([rng.uniform(-2, 2, 4 * n), rng.uniform(-1.5, 1.5, 4 * n), rng.uniform(2, 6, 4 * n)], axis=1)
    uv, z = project(T, pw)
    ok = (z > 0.1) & (uv[:, 0] >= 0) & (uv[:, 0] < w) & (uv[:, 1] >= 0) & (uv[:, 1] < h)
    pw, uv = pw[ok][:n], uv[ok][:n]
    px = uv + rng.normal(0, sigma_px, uv.shape)
    out = rng.random(len(px)) < outlier_frac
    px[out] += rng.uniform(8, 40, (out.sum(), 2)) * rng.choice([-1, 1], (out.sum(), 2))
    entry = true + np.concatenate([rng.normal(0, 0.002, 3), rng.normal(0, 0.0005, 3)])    # sub-pixel, as after sparse alignment
    return dict(true=true, entry=entry, px=px, pw=pw, outlier=out)


def synthetic_vocabulary(k=10, L=3, seed=5, stop_frac=0.05):
    """A random vocabulary tree in DBoW3's binary format (Vocabulary::loadFromBinaryFile: header nb_nodes, size_node, k, L,
    scoring, weighting; per node int parent, 32 descriptor bytes, float weight, byte is_leaf).  Children of a node are
    perturbed copies of it so that descents are meaningful; a fraction of the words is 'stopped' (weight 0).  Node ids are
    assigned level by level (breadth first), children of one parent consecutively."""
    import struct
    rng = np.random.default_rng(seed)
    rec = np.dtype([("parent", "<i4"), ("desc", "u1", (32,)), ("weight", "<f4"), ("leaf", "u1")])
    assert rec.itemsize == 41
    levels, first_id, prev_desc, prev_ids = [], 1, None, np.array([0])
    for lev in range(1, L + 1):
        n = len(prev_ids) * k
        a = np.zeros(n, rec)
        a["parent"] = np.repeat(prev_ids, k)
        if prev_desc is None:
            a["desc"] = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        else:
            flip = np.packbits(rng.random((n, 256)) < 0.25 / lev, axis=1)
            a["desc"] = np.repeat(prev_desc, k, axis=0) ^ flip
        if lev == L:
            a["leaf"] = 1
            w = rng.uniform(0.5, 8.0, n).astype(np.float32)
            w[rng.random(n) < stop_frac] = 0.0
            a["weight"] = w
        levels.append(a)
        prev_desc, prev_ids = a["desc"], np.arange(first_id, first_id + n)
        first_id += n
    body = np.concatenate(levels)
    return struct.pack("<IIiiii", len(body), 41, k, L, 0, 0) + body.tobytes()          # scoring L1_NORM (0), weighting TF_IDF (0)
