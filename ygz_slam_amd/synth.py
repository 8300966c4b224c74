"""Synthetic inputs for bench.py and the tests (host-side numpy; not part of the device path; the fixtures that only tests use
live in tests/fixtures.py).

The reference's own tests need a TUM RGB-D directory that is not in its repo
(test/test_feature_extraction.cpp:16-38), so the workloads of BASELINE.json are
rebuilt procedurally: a textured plane at Z = 2 m seen by a moving
pinhole camera with the TUM-fr2 intrinsics of config/default.yaml:32-35, and local-BA
windows on a pose lattice like the one of test/test_local_ba.cpp:9-18.
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


# ---- local BA windows ---------------------------------------------------------------
def project(T_cw7, pts):
    R = quat_to_R(T_cw7[:4])
    pc = pts @ R.T + T_cw7[4:]
    return np.stack([FX * pc[:, 0] / pc[:, 2] + CX, FY * pc[:, 1] / pc[:, 2] + CY], axis=1), pc[:, 2]


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
