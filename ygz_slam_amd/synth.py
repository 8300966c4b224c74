"""Synthetic inputs for tests and bench.py (host-side numpy; not part of the device path).

The reference's own tests need a TUM RGB-D directory that is not in its repo
(test/test_feature_extraction.cpp:16-38), so the workloads of BASELINE.json are
rebuilt procedurally: a textured plane at Z = 2 m seen by a moving
pinhole camera with the TUM-fr2 intrinsics of config/default.yaml:32-35, and local-BA
windows extending the fixture of test/test_local_ba.cpp:9-37.
"""
import numpy as np

FX, FY, CX, CY = (float(np.float32(v)) for v in (520.9, 521.0, 325.1, 249.7))   # Camera.h:107 stores float
PLANE_Z = 2.0


def make_texture(seed=1, w=640, h=480, margin=160, n_squares=None):
    """World texture = what the identity camera would see, extended by `margin` px."""
    rng = np.random.default_rng(seed)
    W, H = w + 2 * margin, h + 2 * margin
    x = np.arange(W, dtype=np.float64)[None, :] - margin
    y = np.arange(H, dtype=np.float64)[:, None] - margin
    tex = 128.0 + 30.0 * np.sin(x / 23.0) * np.ones_like(y) + 30.0 * np.cos(y / 17.0) * np.ones_like(x)
    tex += 20.0 * np.sin((x + 2 * y) / 41.0)
    if n_squares is None:
        n_squares = int(730 * (W * H) / (640.0 * 480.0))
    xs = rng.integers(0, W - 7, n_squares)
    ys = rng.integers(0, H - 7, n_squares)
    sz = rng.integers(4, 8, n_squares)
    amp = rng.uniform(45, 95, n_squares) * rng.choice([-1.0, 1.0], n_squares)
    for i in range(n_squares):
        tex[ys[i]:ys[i] + sz[i], xs[i]:xs[i] + sz[i]] += amp[i]
    return np.clip(tex, 0, 255), margin


def quat_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def se3_exp(v):
    """[upsilon; omega] -> (qx,qy,qz,qw,tx,ty,tz); same maths as Sophus SE3::exp."""
    v = np.asarray(v, np.float64)
    ups, om = v[:3], v[3:]
    th = np.linalg.norm(om)
    Om = np.array([[0, -om[2], om[1]], [om[2], 0, -om[0]], [-om[1], om[0], 0]])
    if th < 1e-10:
        q = np.array([0.5 * om[0], 0.5 * om[1], 0.5 * om[2], 1.0])
        V = np.eye(3) + 0.5 * Om
    else:
        s = np.sin(th / 2) / th
        q = np.array([s * om[0], s * om[1], s * om[2], np.cos(th / 2)])
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * Om + (th - np.sin(th)) / th ** 3 * (Om @ Om)
    q = q / np.linalg.norm(q)
    return np.concatenate([q, V @ ups])


def render(tex, margin, T_cw, w=640, h=480, noise_sigma=0.0, seed=0):
    """Image of the plane Z=PLANE_Z (world) from camera pose T_cw (7-vector), bilinear."""
    R = quat_to_R(T_cw[:4])
    t = np.asarray(T_cw[4:], np.float64)
    # camera ray r = K^-1 [u v 1]; world point = R^T (s r - t) with Z = PLANE_Z
    u = np.arange(w, dtype=np.float64)[None, :]
    v = np.arange(h, dtype=np.float64)[:, None]
    rx = (u - CX) / FX * np.ones_like(v)
    ry = (v - CY) / FY * np.ones_like(u)
    rz = np.ones_like(rx)
    Rt = R.T
    dw = [Rt[i, 0] * rx + Rt[i, 1] * ry + Rt[i, 2] * rz for i in range(3)]
    ow = -Rt @ t
    s = (PLANE_Z - ow[2]) / dw[2]
    X = ow[0] + s * dw[0]
    Y = ow[1] + s * dw[1]
    tx = X / PLANE_Z * FX + CX + margin
    ty = Y / PLANE_Z * FY + CY + margin
    H, W = tex.shape
    tx = np.clip(tx, 0, W - 1.001)
    ty = np.clip(ty, 0, H - 1.001)
    x0 = np.floor(tx).astype(np.int64)
    y0 = np.floor(ty).astype(np.int64)
    fx_, fy_ = tx - x0, ty - y0
    img = (tex[y0, x0] * (1 - fx_) * (1 - fy_) + tex[y0, x0 + 1] * fx_ * (1 - fy_) +
           tex[y0 + 1, x0] * (1 - fx_) * fy_ + tex[y0 + 1, x0 + 1] * fx_ * fy_)
    if noise_sigma > 0:
        img = img + np.random.default_rng(seed).normal(0, noise_sigma, img.shape)
    depth = s * 1.0          # rz == 1 so the camera-frame depth of each pixel is s
    return np.clip(np.rint(img), 0, 255).astype(np.uint8), depth


def gray_to_bgr(gray, seed=0):
    """A BGR image whose OpenCV-3.1 gray conversion is close to `gray` (for InitFrame)."""
    rng = np.random.default_rng(seed)
    g = gray.astype(np.int16)
    d = rng.integers(-6, 7, gray.shape).astype(np.int16)
    b = np.clip(g + d, 0, 255)
    r = np.clip(g - d // 2, 0, 255)
    return np.stack([b, g, r], axis=-1).astype(np.uint8)


def trajectory(n, seed=11, step=0.004):
    """Smooth camera path T_cw[i] (n x 7): small translations/rotations around identity."""
    rng = np.random.default_rng(seed)
    ph = rng.uniform(0, 2 * np.pi, 6)
    out = np.empty((n, 7))
    for i in range(n):
        s = i * step
        v = np.array([0.08 * np.sin(2.1 * s + ph[0]) + 0.5 * s * 0, 0.05 * np.sin(1.7 * s + ph[1]),
                      0.06 * np.sin(1.3 * s + ph[2]), 0.02 * np.sin(1.9 * s + ph[3]),
                      0.02 * np.sin(2.3 * s + ph[4]), 0.03 * np.sin(1.1 * s + ph[5])])
        out[i] = se3_exp(v)
    return out


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


# ---- local BA windows ---------------------------------------------------------------
TEST_LOCAL_BA_POSES = [  # test/test_local_ba.cpp:9-18  (omega, t)
    ((0, 0, 0), (0, 0, 0)), ((0.1, 0, 0), (0, 0, 0)), ((0, 0.1, 0), (0, 0, 0)), ((0, 0, 0.1), (0, 0, 0)),
    ((0, 0, 0), (0.1, 0, 0)), ((0, 0, 0), (0, 0.1, 0)), ((0, 0, 0), (0, 0, 0.1)), ((0, 0, 0), (0.1, 0.1, 0.1)),
]
TEST_LOCAL_BA_POINTS = [(x, y, z) for z in (2, 3, 4, 5) for (x, y) in ((0, 0), (0, 1), (1, 0), (1, 1))]  # :20-37


def project(T_cw7, pts):
    R = quat_to_R(T_cw7[:4])
    pc = pts @ R.T + T_cw7[4:]
    return np.stack([FX * pc[:, 0] / pc[:, 2] + CX, FY * pc[:, 1] / pc[:, 2] + CY], axis=1), pc[:, 2]


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


def se3_log_t(om, t):
    """upsilon of log(T) given omega and translation t (V^-1 t), as Sophus SE3::log."""
    th = np.linalg.norm(om)
    Om = np.array([[0, -om[2], om[1]], [om[2], 0, -om[0]], [-om[1], om[0], 0]])
    if th < 1e-10:
        Vi = np.eye(3) - 0.5 * Om + (1. / 12.) * (Om @ Om)
    else:
        Vi = np.eye(3) - 0.5 * Om + (1 - th / (2 * np.tan(th / 2))) / (th * th) * (Om @ Om)
    return Vi @ t


def ba_window(K=10, P=2000, seed=7, w=640, h=480, sigma_obs=1.0, sigma_pose=0.1, sigma_pt=0.1, sort_by_point=True):
    """BASELINE config 4: K keyframes on a 0.1-spaced pose lattice, P points in
    [-2,2]x[-1.5,1.5]x[2,6] m, every point observed by every keyframe whose image contains it."""
    rng = np.random.default_rng(seed)
    lattice = []
    for k in range(K):
        om = 0.1 * np.array([(k % 3) - 1, ((k // 3) % 3) - 1, 0]) * 0.5
        t = 0.1 * np.array([(k % 2), ((k // 2) % 2), ((k // 4) % 2)]) + 0.02 * k
        lattice.append(np.concatenate([om, se3_log_t(om, t)]))
    true_poses = np.array(lattice)
    true_pts = np.stack([rng.uniform(-2, 2, P), rng.uniform(-1.5, 1.5, P), rng.uniform(2, 6, P)], axis=1)
    ep, el, obs = [], [], []
    for k in range(K):
        T = se3_exp(np.concatenate([true_poses[k, 3:], true_poses[k, :3]]))
        uv, z = project(T, true_pts)
        ok = (z > 0.1) & (uv[:, 0] >= 0) & (uv[:, 0] < w) & (uv[:, 1] >= 0) & (uv[:, 1] < h)
        idx = np.nonzero(ok)[0]
        ep.append(np.full(len(idx), k, np.int32))
        el.append(idx.astype(np.int32))
        obs.append(uv[idx])
    ep, el, obs = np.concatenate(ep), np.concatenate(el), np.concatenate(obs)
    if sort_by_point:
        order = np.lexsort((ep, el))
        ep, el, obs = ep[order], el[order], obs[order]
    obs = obs + rng.normal(0, sigma_obs, obs.shape)
    est_poses = true_poses.copy()
    est_poses[1:] += rng.normal(0, sigma_pose, est_poses[1:].shape) * 0.3
    est_pts = true_pts + rng.normal(0, sigma_pt, true_pts.shape)
    fixed = np.zeros(K, np.uint8)
    fixed[0] = 1
    return dict(poses=est_poses, fixed=fixed, points=est_pts, edge_pose=ep, edge_point=el, obs=obs,
                true_poses=true_poses, true_points=true_pts)


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


class Sequence:
    """Procedural RGB-D sequence for the offline run (BASELINE configs[4]): frame(i) -> BGR uint8 [h, w, 3],
    depth(i) -> float64 [h, w].  Every frame depends on (seed, i) only, so any rank renders exactly the frames it owns."""

    def __init__(self, n, w=1280, h=720, seed=11, step=0.02, noise_sigma=1.0):
        self.n, self.w, self.h, self.seed, self.noise = n, w, h, seed, noise_sigma
        self.tex, self.margin = make_texture(seed, w, h, margin=max(160, w // 4))
        self.poses = trajectory(n, seed + 10, step)
        self.poses[0] = [0, 0, 0, 1, 0, 0, 0]
        self._last = (None, None, None)

    def _render(self, i):
        if self._last[0] != i:
            im, d = render(self.tex, self.margin, self.poses[i], self.w, self.h, self.noise, self.seed * 100003 + i)
            self._last = (i, gray_to_bgr(im, i), d)
        return self._last

    def frame(self, i):
        return self._render(i)[1]

    def depth(self, i):
        return self._render(i)[2]
