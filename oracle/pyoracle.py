"""ctypes binding of the CPU ORACLE (oracle/libygz_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under ygz_slam_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MAX_LEVELS = 8


def build(force=False, variant=""):
    """Compile the oracle with gcc (seconds).  variant '' or 'o3'."""
    name = "libygz_oracle.so" if not variant else "libygz_oracle_%s.so" % variant
    path = os.path.join(_HERE, name)
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    stale = (not os.path.exists(path)) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, name], stdout=subprocess.DEVNULL)
    return path


class Pyramid(C.Structure):
    _fields_ = [("levels", C.c_int), ("w", C.c_int * MAX_LEVELS), ("h", C.c_int * MAX_LEVELS),
                ("img", C.POINTER(C.c_uint8) * MAX_LEVELS)]


class DetectParams(C.Structure):
    _fields_ = [("image_width", C.c_int), ("image_height", C.c_int), ("cell_size", C.c_int),
                ("detection_threshold", C.c_double), ("pyramid_levels", C.c_int),
                ("nms_tie_suppress", C.c_int)]


class Keypoint(C.Structure):
    _fields_ = [("px", C.c_double), ("py", C.c_double), ("level", C.c_int), ("score", C.c_float),
                ("angle", C.c_float), ("desc", C.c_uint8 * 32)]


KP_DTYPE = np.dtype([("px", "<f8"), ("py", "<f8"), ("level", "<i4"), ("score", "<f4"),
                     ("angle", "<f4"), ("desc", "u1", (32,))], align=True)
assert KP_DTYPE.itemsize == C.sizeof(Keypoint)


class SE3(C.Structure):
    _fields_ = [("q", C.c_double * 4), ("t", C.c_double * 3)]

    @staticmethod
    def from_array(a):          # a: 7 doubles qx qy qz qw tx ty tz
        s = SE3()
        for i in range(4):
            s.q[i] = float(a[i])
        for i in range(3):
            s.t[i] = float(a[4 + i])
        return s

    def to_array(self):
        return np.array(list(self.q) + list(self.t), dtype=np.float64)


class Camera(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float)]


class KltParams(C.Structure):
    _fields_ = [("win", C.c_int), ("max_level", C.c_int), ("max_iter", C.c_int), ("eps", C.c_double),
                ("min_eig_threshold", C.c_double), ("use_initial_flow", C.c_int)]


class SparseAlignStats(C.Structure):
    _fields_ = [("n_iter_total", C.c_int), ("n_meas_last", C.c_int), ("chi2_last", C.c_double),
                ("iters_per_level", C.c_int * MAX_LEVELS)]


class BaProblem(C.Structure):
    _fields_ = [("n_poses", C.c_int), ("n_points", C.c_int), ("n_edges", C.c_int),
                ("poses", C.POINTER(C.c_double)), ("pose_fixed", C.POINTER(C.c_uint8)),
                ("points", C.POINTER(C.c_double)), ("edge_pose", C.POINTER(C.c_int32)),
                ("edge_point", C.POINTER(C.c_int32)), ("obs", C.POINTER(C.c_double)),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("huber_delta", C.c_double)]


class Vocab(C.Structure):
    _fields_ = [("k", C.c_int), ("L", C.c_int), ("scoring", C.c_int), ("weighting", C.c_int), ("n_nodes", C.c_int),
                ("n_words", C.c_int), ("parent", C.POINTER(C.c_int32)), ("desc", C.POINTER(C.c_uint8)),
                ("weight", C.POINTER(C.c_double)), ("word_id", C.POINTER(C.c_int32)), ("child_off", C.POINTER(C.c_int32)),
                ("child", C.POINTER(C.c_int32))]


class CeresProblem(C.Structure):
    _fields_ = [("n_poses", C.c_int), ("n_points", C.c_int), ("n_edges", C.c_int),
                ("poses", C.POINTER(C.c_double)), ("pose_fixed", C.POINTER(C.c_uint8)),
                ("points", C.POINTER(C.c_double)), ("point_fixed", C.POINTER(C.c_uint8)),
                ("edge_pose", C.POINTER(C.c_int32)), ("edge_point", C.POINTER(C.c_int32)),
                ("obs_n", C.POINTER(C.c_double)), ("edge_huber", C.POINTER(C.c_double)),
                ("edge_enable", C.POINTER(C.c_uint8)), ("fail_behind_camera", C.c_int)]


class CeresOptions(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int), ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
                ("parameter_tolerance", C.c_double), ("initial_trust_region_radius", C.c_double),
                ("max_trust_region_radius", C.c_double), ("min_trust_region_radius", C.c_double),
                ("min_relative_decrease", C.c_double), ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
                ("jacobi_scaling", C.c_int), ("max_num_consecutive_invalid_steps", C.c_int), ("trust_region_strategy", C.c_int)]


class CeresSummary(C.Structure):
    _fields_ = [("iterations", C.c_int), ("successful_steps", C.c_int), ("unsuccessful_steps", C.c_int),
                ("termination", C.c_int), ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("final_radius", C.c_double)]


class LmStats(C.Structure):
    _fields_ = [("iterations", C.c_int), ("lm_trials", C.c_int), ("chi2_initial", C.c_double),
                ("chi2_final", C.c_double), ("lambda_final", C.c_double)]


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _u8(a):
    return _p(a, C.c_uint8)


def _f64(a):
    return _p(a, C.c_double)


class Oracle:
    """Thin numpy front-end; every method names the yo_* function it calls."""

    def __init__(self, variant=None):
        if variant is None:
            variant = os.environ.get("YGZ_ORACLE_VARIANT", "")      # e.g. "asan": the sanitizer run of the test suite
        self.lib = C.CDLL(build(variant=variant))
        L = self.lib
        L.yo_shi_tomasi.restype = C.c_float
        L.yo_fast_atan2.restype = C.c_float
        L.yo_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.yo_ic_angle.restype = C.c_float
        L.yo_ic_angle.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_double, C.c_double]
        L.yo_orb_descriptor.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_double, C.c_double,
                                        C.c_int, C.c_float, C.POINTER(C.c_uint8)]
        L.yo_sparse_align.restype = C.c_size_t
        L.yo_sparse_align_linearize.restype = C.c_double
        L.yo_ba_linearize.restype = C.c_double
        L.yo_ba_edge_error.argtypes = [C.POINTER(C.c_double)] * 3 + [C.c_double] * 4 + [C.POINTER(C.c_double)]
        L.yo_ba_edge_jacobians.argtypes = [C.POINTER(C.c_double)] * 2 + [C.c_double] * 2 + [C.POINTER(C.c_double)] * 2
        L.yo_find_direct_projection.argtypes = [
            C.POINTER(Camera), C.POINTER(Pyramid), C.POINTER(SE3), C.POINTER(Pyramid), C.POINTER(SE3),
            C.POINTER(C.c_double), C.c_double, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]

    # ---- images ----
    def bgr2gray(self, bgr):
        h, w, _ = bgr.shape
        bgr = np.ascontiguousarray(bgr, np.uint8)
        out = np.empty((h, w), np.uint8)
        self.lib.yo_bgr2gray(_u8(bgr), w, h, w * 3, _u8(out))
        return out

    def pyr_down(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        out = np.empty(((h + 1) // 2, (w + 1) // 2), np.uint8)
        self.lib.yo_pyr_down(_u8(img), w, h, _u8(out))
        return out

    def pyramid(self, gray, levels=3):
        lv = [np.ascontiguousarray(gray, np.uint8)]
        for _ in range(1, levels):
            lv.append(self.pyr_down(lv[-1]))
        return lv

    @staticmethod
    def _pyr_struct(levels):
        p = Pyramid()
        p.levels = len(levels)
        for i, im in enumerate(levels):
            assert im.flags["C_CONTIGUOUS"] and im.dtype == np.uint8
            p.w[i] = im.shape[1]
            p.h[i] = im.shape[0]
            p.img[i] = _u8(im)
        return p

    # ---- FAST ----
    def fast_detect(self, img, thr):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        xy = np.empty((w * h, 2), np.int16)
        n = self.lib.yo_fast10_detect(_u8(img), w, h, w, int(thr), _p(xy, C.c_int16), w * h)
        return xy[:n].copy()

    def fast_score(self, img, xy, thr):
        img = np.ascontiguousarray(img, np.uint8)
        xy = np.ascontiguousarray(xy, np.int16)
        sc = np.empty(len(xy), np.int32)
        self.lib.yo_fast10_score(_u8(img), img.shape[1], _p(xy, C.c_int16), len(xy), int(thr), _p(sc, C.c_int))
        return sc

    def fast_score_closed_form(self, img, x, y):
        img = np.ascontiguousarray(img, np.uint8)
        p = C.cast(C.addressof(_u8(img).contents) + y * img.shape[1] + x, C.POINTER(C.c_uint8))
        return self.lib.yo_fast10_score_closed_form(p, img.shape[1])

    def fast_nonmax(self, xy, scores, tie_suppress=0):
        xy = np.ascontiguousarray(xy, np.int16)
        scores = np.ascontiguousarray(scores, np.int32)
        nm = np.empty(max(len(xy), 1), np.int32)
        m = self.lib.yo_fast_nonmax_3x3(_p(xy, C.c_int16), _p(scores, C.c_int), len(xy), tie_suppress, _p(nm, C.c_int))
        return nm[:m].copy()

    def level_corners(self, img, thr, tie_suppress=0):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        xy = np.empty((w * h, 2), np.int16)
        sc = np.empty(w * h, np.int32)
        m = self.lib.yo_detect_level_corners(_u8(img), w, h, int(thr), tie_suppress, _p(xy, C.c_int16), _p(sc, C.c_int), w * h)
        return xy[:m].copy(), sc[:m].copy()

    # ---- extractor ----
    def default_params(self, w=640, h=480, levels=3):
        p = DetectParams()
        self.lib.yo_detect_params_default(C.byref(p))
        p.image_width, p.image_height, p.pyramid_levels = w, h, levels
        return p

    def shi_tomasi(self, img, u, v):
        img = np.ascontiguousarray(img, np.uint8)
        return float(self.lib.yo_shi_tomasi(_u8(img), img.shape[1], img.shape[0], img.shape[1], int(u), int(v)))

    def fast_atan2(self, y, x):
        return float(self.lib.yo_fast_atan2(float(y), float(x)))

    def ic_angle(self, img, ptx, pty):
        img = np.ascontiguousarray(img, np.uint8)
        return float(self.lib.yo_ic_angle(_u8(img), img.shape[1], img.shape[0], float(ptx), float(pty)))

    def orb_descriptor(self, img, px, py, level, angle):
        img = np.ascontiguousarray(img, np.uint8)
        d = np.empty(32, np.uint8)
        self.lib.yo_orb_descriptor(_u8(img), img.shape[1], img.shape[0], float(px), float(py), int(level), float(angle), _u8(d))
        return d

    def detect(self, levels, params=None, occupied=None):
        """levels: list of uint8 level images.  Returns KP_DTYPE structured array."""
        prm = params or self.default_params(levels[0].shape[1], levels[0].shape[0], len(levels))
        pyr = self._pyr_struct(levels)
        rows = -(-prm.image_height // prm.cell_size)
        cols = -(-prm.image_width // prm.cell_size)
        out = np.zeros(rows * cols, KP_DTYPE)
        occ = None
        if occupied is not None:
            occupied = np.ascontiguousarray(occupied, np.uint8)
            occ = _u8(occupied)
        n = self.lib.yo_detect(C.byref(pyr), C.byref(prm), occ, out.ctypes.data_as(C.POINTER(Keypoint)))
        return out[:n].copy()

    def describe(self, levels, kps):
        pyr = self._pyr_struct(levels)
        kps = np.ascontiguousarray(kps, KP_DTYPE).copy()
        self.lib.yo_describe(C.byref(pyr), kps.ctypes.data_as(C.POINTER(Keypoint)), len(kps))
        return kps

    # ---- hamming ----
    def descriptor_distance(self, a, b):
        a = np.ascontiguousarray(a, np.uint8)
        b = np.ascontiguousarray(b, np.uint8)
        return int(self.lib.yo_descriptor_distance(_u8(a), _u8(b)))

    def hamming_nn(self, q, t):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        idx = np.empty(len(q), np.int32)
        d = np.empty(len(q), np.int32)
        d2 = np.empty(len(q), np.int32)
        self.lib.yo_hamming_nn(_u8(q), len(q), _u8(t), len(t), _p(idx, C.c_int32), _p(d, C.c_int32), _p(d2, C.c_int32))
        return idx, d, d2

    def bf_match(self, q, t, cross_check=1):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        idx = np.empty(len(q), np.int32)
        d = np.empty(len(q), np.int32)
        n = self.lib.yo_bf_match(_u8(q), len(q), _u8(t), len(t), int(cross_check), _p(idx, C.c_int32), _p(d, C.c_int32))
        return idx, d, n

    def good_match_filter(self, idx, dist):
        idx = np.ascontiguousarray(idx, np.int32)
        dist = np.ascontiguousarray(dist, np.int32)
        keep = np.empty(len(idx), np.uint8)
        n = self.lib.yo_good_match_filter(_p(idx, C.c_int32), _p(dist, C.c_int32), len(idx), _u8(keep))
        return keep.astype(bool), n

    def check_frame_descriptors(self, desc1, desc2, idx1, idx2, init_low=30, init_high=80, ratio=3.0):
        d1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32); d2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
        i1 = np.ascontiguousarray(idx1, np.int32); i2 = np.ascontiguousarray(idx2, np.int32)
        n = len(i1)
        dist = np.zeros(max(n, 1), np.int32); keep = np.zeros(max(n, 1), np.uint8); best = C.c_int(0)
        cnt = self.lib.yo_check_frame_descriptors(_u8(d1), _u8(d2), _p(i1, C.c_int32), _p(i2, C.c_int32), n, int(init_low), int(init_high),
                                                  C.c_float(ratio), _p(dist, C.c_int32), _u8(keep), C.byref(best))
        return dist[:n].copy(), keep[:n].astype(bool), cnt, best.value

    # ---- SE3 ----
    def se3_exp(self, v):
        v = np.ascontiguousarray(v, np.float64)
        T = SE3()
        self.lib.yo_se3_exp(_f64(v), C.byref(T))
        return T.to_array()

    def se3_log(self, T7):
        T = SE3.from_array(T7)
        out = np.empty(6)
        self.lib.yo_se3_log(C.byref(T), _f64(out))
        return out

    def se3_mul(self, A7, B7):
        A, B, Cc = SE3.from_array(A7), SE3.from_array(B7), SE3()
        self.lib.yo_se3_mul(C.byref(A), C.byref(B), C.byref(Cc))
        return Cc.to_array()

    def se3_inv(self, A7):
        A, B = SE3.from_array(A7), SE3()
        self.lib.yo_se3_inv(C.byref(A), C.byref(B))
        return B.to_array()

    def se3_act(self, T7, p):
        T = SE3.from_array(T7)
        p = np.ascontiguousarray(p, np.float64)
        out = np.empty(3)
        self.lib.yo_se3_act(C.byref(T), _f64(p), _f64(out))
        return out

    def camera(self):
        c = Camera()
        self.lib.yo_camera_default(C.byref(c))
        return c

    # ---- alignment ----
    def align2d(self, cur, pwb, patch, u, v, n_iter=10):
        cur = np.ascontiguousarray(cur, np.uint8)
        pwb = np.ascontiguousarray(pwb, np.uint8)
        patch = np.ascontiguousarray(patch, np.uint8)
        uu, vv = C.c_double(u), C.c_double(v)
        chi2, it = C.c_float(0), C.c_int(0)
        ok = self.lib.yo_align2d(_u8(cur), cur.shape[1], cur.shape[0], cur.shape[1], _u8(pwb), _u8(patch), n_iter,
                                 C.byref(uu), C.byref(vv), C.byref(chi2), C.byref(it))
        return bool(ok), uu.value, vv.value, chi2.value, it.value

    def find_direct_projection(self, ref_levels, T_ref, cur_levels, T_cur, px_ref, depth, level_ref, px_cur, cam=None):
        cam = cam or self.camera()
        pr, pc = self._pyr_struct(ref_levels), self._pyr_struct(cur_levels)
        Tr, Tc = SE3.from_array(T_ref), SE3.from_array(T_cur)
        pxr = np.ascontiguousarray(px_ref, np.float64)
        pxc = np.ascontiguousarray(px_cur, np.float64).copy()
        sl = C.c_int(0)
        ok = self.lib.yo_find_direct_projection(C.byref(cam), C.byref(pr), C.byref(Tr), C.byref(pc), C.byref(Tc),
                                                _f64(pxr), float(depth), int(level_ref), _f64(pxc), C.byref(sl))
        return bool(ok), pxc, sl.value

    def find_direct_projection_n(self, ref_levels, T_ref, cur_levels, T_cur, px_ref, depth, level_ref, px_cur, cam=None):
        cam = cam or self.camera()
        pr, pc = self._pyr_struct(ref_levels), self._pyr_struct(cur_levels)
        Tr, Tc = SE3.from_array(T_ref), SE3.from_array(T_cur)
        pxr = np.ascontiguousarray(px_ref, np.float64).reshape(-1, 2)
        dep = np.ascontiguousarray(depth, np.float64); lvl = np.ascontiguousarray(level_ref, np.int32)
        pxc = np.ascontiguousarray(px_cur, np.float64).reshape(-1, 2).copy()
        n = len(dep)
        sl = np.zeros(n, np.int32); ok = np.zeros(n, np.uint8)
        self.lib.yo_find_direct_projection_n(C.byref(cam), C.byref(pr), C.byref(Tr), C.byref(pc), C.byref(Tc), n, _f64(pxr), _f64(dep),
                                             _p(lvl, C.c_int32), _f64(pxc), _p(sl, C.c_int32), _u8(ok))
        return ok.astype(bool), pxc, sl

    def track_candidates(self, T_ref, T_cur, px_ref, depth, w, h, cam=None):
        """yo_track_candidates: map points of the reference features + LocalMapping::FindCandidates with the current pose"""
        cam = cam or self.camera()
        Tr, Tc = SE3.from_array(T_ref), SE3.from_array(T_cur)
        pxr = np.ascontiguousarray(px_ref, np.float64).reshape(-1, 2)
        dep = np.ascontiguousarray(depth, np.float64)
        n = len(dep)
        pw = np.zeros((max(n, 1), 3)); pred = np.zeros((max(n, 1), 2)); cand = np.zeros(max(n, 1), np.uint8)
        self.lib.yo_track_candidates(C.byref(cam), C.byref(Tr), C.byref(Tc), _f64(pxr), _f64(dep), n, int(w), int(h), _f64(pw), _f64(pred),
                                     _u8(cand))
        return pw[:n], pred[:n], cand[:n].astype(bool)

    def depth_filter_update(self, cur_levels, T_cur, ref_levels_list, T_refs, seeds, batch_counter, max_n_kfs=5, conv_thresh=100.0, cam=None):
        """yo_depth_filter_update: legacy DepthFilter::UpdateSeeds for the seeds (dict as in _lib.HipContext.depth_filter_update)"""
        cam = cam or self.camera()
        R = len(ref_levels_list)
        pyrs = (Pyramid * max(R, 1))(*[self._pyr_struct(l) for l in ref_levels_list])
        Ts = (SE3 * max(R, 1))(*[SE3.from_array(t) for t in T_refs])
        pc, Tc = self._pyr_struct(cur_levels), SE3.from_array(T_cur)
        kp = np.ascontiguousarray(seeds["kp"], np.float32).reshape(-1, 2); n = len(kp)
        oc = np.ascontiguousarray(seeds["octave"], np.int32); sr = np.ascontiguousarray(seeds["ref"], np.int32)
        fid = np.ascontiguousarray(seeds["frame_id"], np.uint64)
        f = {k: np.ascontiguousarray(seeds[k], np.float32).copy() for k in ("a", "b", "mu", "z_range", "sigma2")}
        st = np.zeros(max(n, 1), np.int32); z = np.zeros(max(n, 1)); mp = np.zeros((max(n, 1), 2)); pw = np.zeros((max(n, 1), 3))
        fp = lambda k: _p(f[k], C.c_float)
        nu = self.lib.yo_depth_filter_update(C.byref(cam), pyrs, Ts, _p(sr, C.c_int32), _p(fid, C.c_uint64), int(batch_counter), int(max_n_kfs),
                                             C.c_double(conv_thresh), C.byref(pc), C.byref(Tc), n, _p(kp, C.c_float), _p(oc, C.c_int32),
                                             fp("a"), fp("b"), fp("mu"), fp("z_range"), fp("sigma2"), _p(st, C.c_int32), _f64(z), _f64(mp), _f64(pw))
        return dict(a=f["a"], b=f["b"], mu=f["mu"], sigma2=f["sigma2"], state=st[:n], z=z[:n], matched_px=mp[:n], pos_world=pw[:n], updated=int(nu))

    def create_map_points(self, lv1, T1, lv2, T2, px1, level1, px2, cam=None):
        """yo_create_map_points: the triangulation loop of LocalMapping::CreateNewMapPoints"""
        cam = cam or self.camera()
        p1s, p2s = self._pyr_struct(lv1), self._pyr_struct(lv2)
        Ta, Tb = SE3.from_array(T1), SE3.from_array(T2)
        p1 = np.ascontiguousarray(px1, np.float64).reshape(-1, 2); l1 = np.ascontiguousarray(level1, np.int32)
        p2 = np.ascontiguousarray(px2, np.float64).reshape(-1, 2).copy()
        n = len(l1)
        code = np.zeros(max(n, 1), np.int32); d1 = np.zeros(max(n, 1)); d2 = np.zeros(max(n, 1)); pw = np.zeros((max(n, 1), 3))
        sl = np.zeros(max(n, 1), np.int32)
        created = self.lib.yo_create_map_points(C.byref(cam), C.byref(p1s), C.byref(Ta), C.byref(p2s), C.byref(Tb), n, _f64(p1), _p(l1, C.c_int32),
                                                _f64(p2), _p(code, C.c_int32), _f64(d1), _f64(d2), _f64(pw), _p(sl, C.c_int32))
        return dict(px2=p2, code=code[:n], depth1=d1[:n], depth2=d2[:n], pos_world=pw[:n], search_level=sl[:n], created=int(created))

    def find_direct_projection_mp(self, ref_levels, T_ref, cur_levels, T_cur, pos_world, px_ref, level_ref, px_cur, cam=None):
        cam = cam or self.camera()
        pr, pc = self._pyr_struct(ref_levels), self._pyr_struct(cur_levels)
        Tr, Tc = SE3.from_array(T_ref), SE3.from_array(T_cur)
        pw = np.ascontiguousarray(pos_world, np.float64)
        pxr = np.ascontiguousarray(px_ref, np.float64)
        pxc = np.ascontiguousarray(px_cur, np.float64).copy()
        sl = C.c_int(0)
        ok = self.lib.yo_find_direct_projection_mp(C.byref(cam), C.byref(pr), C.byref(Tr), C.byref(pc), C.byref(Tc), _f64(pw),
                                                   _f64(pxr), int(level_ref), _f64(pxc), C.byref(sl))
        return bool(ok), pxc, sl.value

    def track_local_map(self, kf_levels, kf_T, cur_levels, T_cur, pos_world, point_bad, cand_point, cand_kf, cand_px_ref,
                        cand_level, cam=None):
        """yo_track_local_map: LocalMapping::FindCandidates + ProjectMapPoints (LocalMapping.cpp:47-120)"""
        cam = cam or self.camera()
        K = len(kf_levels)
        pyrs = (Pyramid * max(K, 1))(*[self._pyr_struct(l) for l in kf_levels])
        Ts = (SE3 * max(K, 1))(*[SE3.from_array(t) for t in kf_T])
        pc, Tc = self._pyr_struct(cur_levels), SE3.from_array(T_cur)
        pw = np.ascontiguousarray(pos_world, np.float64).reshape(-1, 3)
        P = pw.shape[0]
        bad = np.ascontiguousarray(point_bad if point_bad is not None else np.zeros(P), np.uint8)
        cp = np.ascontiguousarray(cand_point, np.int32); ck = np.ascontiguousarray(cand_kf, np.int32)
        cx = np.ascontiguousarray(cand_px_ref, np.float64).reshape(-1, 2); cl = np.ascontiguousarray(cand_level, np.int32)
        Cn = cp.shape[0]
        in_view = np.zeros(P, np.uint8); px_proj = np.zeros((P, 2)); match = np.zeros(P, np.int32)
        px_match = np.zeros((P, 2)); lvl = np.zeros(P, np.int32)
        n = self.lib.yo_track_local_map(C.byref(cam), pyrs, Ts, K, C.byref(pc), C.byref(Tc), _f64(pw), _u8(bad), P,
                                        _p(cp, C.c_int32), _p(ck, C.c_int32), _f64(cx), _p(cl, C.c_int32), Cn,
                                        _u8(in_view), _f64(px_proj), _p(match, C.c_int32), _f64(px_match), _p(lvl, C.c_int32))
        return int(n), in_view, px_proj, match, px_match, lvl

    def sparse_align(self, ref_levels, T_ref, cur_levels, T_cur, px, depth, has_mp, max_level=2, min_level=0,
                     n_iter=30, cam=None, method="gn"):
        """SparseImgAlign::run; method "gn": NLLSSolver::optimizeGaussNewton (the live path), "lm": optimizeLevenbergMarquardt"""
        cam = cam or self.camera()
        pr, pc = self._pyr_struct(ref_levels), self._pyr_struct(cur_levels)
        Tr, Tc = SE3.from_array(T_ref), SE3.from_array(T_cur)
        px = np.ascontiguousarray(px, np.float64)
        depth = np.ascontiguousarray(depth, np.float64)
        has_mp = np.ascontiguousarray(has_mp, np.uint8)
        st = SparseAlignStats()
        fn = self.lib.yo_sparse_align_lm if method == "lm" else self.lib.yo_sparse_align
        fn.restype = C.c_size_t
        n = fn(C.byref(cam), C.byref(pr), C.byref(Tr), C.byref(pc), C.byref(Tc), _f64(px),
               _f64(depth), _u8(has_mp), len(depth), max_level, min_level, n_iter, C.byref(st))
        return int(n), Tc.to_array(), st

    def sparse_align_linearize(self, ref_levels, cur_levels, T_cur_ref, px, depth, has_mp, level, visible=None, cam=None):
        cam = cam or self.camera()
        pr, pc = self._pyr_struct(ref_levels), self._pyr_struct(cur_levels)
        T = SE3.from_array(T_cur_ref)
        px = np.ascontiguousarray(px, np.float64)
        depth = np.ascontiguousarray(depth, np.float64)
        has_mp = np.ascontiguousarray(has_mp, np.uint8)
        vis = np.zeros(len(depth), np.uint8) if visible is None else np.ascontiguousarray(visible, np.uint8).copy()
        H, J = np.empty(36), np.empty(6)
        nm = C.c_int(0)
        chi2 = self.lib.yo_sparse_align_linearize(C.byref(cam), C.byref(pr), C.byref(pc), C.byref(T), _f64(px),
                                                  _f64(depth), _u8(has_mp), len(depth), level, _u8(vis), _f64(H),
                                                  _f64(J), C.byref(nm))
        return float(chi2), H.reshape(6, 6), J, nm.value, vis

    def ldlt6_solve(self, H, b):
        H = np.ascontiguousarray(H, np.float64)
        b = np.ascontiguousarray(b, np.float64)
        x = np.empty(6)
        ok = self.lib.yo_ldlt6_solve(_f64(H), _f64(b), _f64(x))
        return bool(ok), x

    # ---- KLT ----
    def klt_params(self):
        p = KltParams()
        self.lib.yo_klt_params_default(C.byref(p))
        return p

    def klt_track(self, prev, nxt, prev_pts, next_pts_init, params=None):
        prm = params or self.klt_params()
        prev = np.ascontiguousarray(prev, np.uint8)
        nxt = np.ascontiguousarray(nxt, np.uint8)
        pp = np.ascontiguousarray(prev_pts, np.float32)
        npts = np.ascontiguousarray(next_pts_init, np.float32).copy()
        st = np.empty(len(pp), np.uint8)
        err = np.empty(len(pp), np.float32)
        self.lib.yo_klt_track(_u8(prev), _u8(nxt), prev.shape[1], prev.shape[0], _p(pp, C.c_float), _p(npts, C.c_float),
                              len(pp), C.byref(prm), _u8(st), _p(err, C.c_float))
        return npts, st, err

    # ---- BA ----
    def ba_linearize(self, poses, pose_fixed, points, edge_pose, edge_point, obs, cam=None, huber_delta=5.991):
        cam = cam or self.camera()
        poses = np.ascontiguousarray(poses, np.float64)
        pose_fixed = np.ascontiguousarray(pose_fixed, np.uint8)
        points = np.ascontiguousarray(points, np.float64)
        edge_pose = np.ascontiguousarray(edge_pose, np.int32)
        edge_point = np.ascontiguousarray(edge_point, np.int32)
        obs = np.ascontiguousarray(obs, np.float64)
        pb = BaProblem(len(poses), len(points), len(edge_pose), _f64(poses), _u8(pose_fixed), _f64(points),
                       _p(edge_pose, C.c_int32), _p(edge_point, C.c_int32), _f64(obs),
                       float(cam.fx), float(cam.fy), float(cam.cx), float(cam.cy), float(huber_delta))
        K, P, E = len(poses), len(points), len(edge_pose)
        out = dict(Hpp=np.empty((K, 6, 6)), bp=np.empty((K, 6)), Hll=np.empty((P, 3, 3)), bl=np.empty((P, 3)),
                   Hpl=np.empty((E, 6, 3)), err=np.empty((E, 2)), chi2_edge=np.empty(E))
        out["chi2"] = float(self.lib.yo_ba_linearize(C.byref(pb), _f64(out["Hpp"]), _f64(out["bp"]), _f64(out["Hll"]),
                                                     _f64(out["bl"]), _f64(out["Hpl"]), _f64(out["err"]),
                                                     _f64(out["chi2_edge"])))
        return out

    def ba_edge(self, pose, pt, obs, cam=None):
        cam = cam or self.camera()
        pose = np.ascontiguousarray(pose, np.float64)
        pt = np.ascontiguousarray(pt, np.float64)
        obs = np.ascontiguousarray(obs, np.float64)
        err, Jp, Jx = np.empty(2), np.empty(6), np.empty(12)
        self.lib.yo_ba_edge_error(_f64(pose), _f64(pt), _f64(obs), float(cam.fx), float(cam.fy), float(cam.cx),
                                  float(cam.cy), _f64(err))
        self.lib.yo_ba_edge_jacobians(_f64(pose), _f64(pt), float(cam.fx), float(cam.fy), _f64(Jp), _f64(Jx))
        return err, Jp.reshape(2, 3), Jx.reshape(2, 6)

    def ba_edge_norm(self, pose_tw, pt, obs_n):
        pose_tw = np.ascontiguousarray(pose_tw, np.float64)
        pt = np.ascontiguousarray(pt, np.float64)
        obs_n = np.ascontiguousarray(obs_n, np.float64)
        err, Jp, Jx = np.empty(2), np.empty(6), np.empty(12)
        self.lib.yo_ba_edge_error_norm(_f64(pose_tw), _f64(pt), _f64(obs_n), _f64(err))
        self.lib.yo_ba_edge_jacobians_norm(_f64(pose_tw), _f64(pt), _f64(Jp), _f64(Jx))
        return err, Jp.reshape(2, 3), Jx.reshape(2, 6)

    def depth_from_triangulation(self, T_search_ref, f_ref, f_cur, determinant_th=1e-5):
        T = SE3.from_array(T_search_ref)
        fr = np.ascontiguousarray(f_ref, np.float64).reshape(-1, 3); fc = np.ascontiguousarray(f_cur, np.float64).reshape(-1, 3)
        d1, d2, ok = np.full(len(fr), np.nan), np.full(len(fr), np.nan), np.zeros(len(fr), np.uint8)
        a, b = C.c_double(), C.c_double()
        self.lib.yo_depth_from_triangulation.argtypes = [C.POINTER(SE3), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double,
                                                         C.POINTER(C.c_double), C.POINTER(C.c_double)]
        for i in range(len(fr)):
            if self.lib.yo_depth_from_triangulation(C.byref(T), _f64(fr[i]), _f64(fc[i]), float(determinant_th), C.byref(a), C.byref(b)):
                d1[i], d2[i], ok[i] = a.value, b.value, 1
        return d1, d2, ok

    # ---- BoW (oracle/bow.c) ----
    def vocab_parse(self, blob):
        """blob: bytes of a DBoW3 binary vocabulary (loadFromBinaryFile format).  Returns an opaque handle (keep it)."""
        v = Vocab()
        buf = np.frombuffer(blob, np.uint8).copy()
        if self.lib.yo_vocab_parse(_u8(buf), C.c_size_t(len(buf)), C.byref(v)) != 0:
            raise ValueError("not a DBoW3 binary vocabulary")
        return v

    def bow_transform(self, vocab, desc, levelsup=4):
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        n = len(desc)
        word, weight, node = np.empty(max(n, 1), np.int32), np.empty(max(n, 1)), np.empty(max(n, 1), np.int32)
        bw, bv = np.empty(max(n, 1), np.int32), np.empty(max(n, 1))
        m = self.lib.yo_bow_transform(C.byref(vocab), _u8(desc), n, levelsup, _p(word, C.c_int32), _f64(weight), _p(node, C.c_int32),
                                      _p(bw, C.c_int32), _f64(bv))
        return word[:n], weight[:n], node[:n], bw[:m].copy(), bv[:m].copy()

    def search_by_bow(self, desc1, node1, desc2, node2, th_low=65, knn_ratio=0.7):
        d1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32); d2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
        n1a = np.ascontiguousarray(node1, np.int32); n2a = np.ascontiguousarray(node2, np.int32)
        m = np.empty(max(len(d1), 1), np.int32)
        self.lib.yo_search_by_bow.argtypes = [C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_int32),
                                              C.c_int, C.c_int, C.c_float, C.POINTER(C.c_int32)]
        cnt = self.lib.yo_search_by_bow(_u8(d1), _p(n1a, C.c_int32), len(d1), _u8(d2), _p(n2a, C.c_int32), len(d2), int(th_low),
                                        float(knn_ratio), _p(m, C.c_int32))
        return m[:len(d1)], cnt

    def bow_orientation(self, angle1, angle2, match12):
        """the checkOrientation part of SearchByBoW: (count after the histogram test, hist [30], the three maxima)"""
        a1 = np.ascontiguousarray(angle1, np.float64); a2 = np.ascontiguousarray(angle2, np.float64); m = np.ascontiguousarray(match12, np.int32)
        hist = np.zeros(30, np.int32); ind = np.zeros(3, np.int32)
        self.lib.yo_bow_orientation.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int32),
                                                C.POINTER(C.c_int32)]
        cnt = self.lib.yo_bow_orientation(_f64(a1), _f64(a2), _p(m, C.c_int32), len(m), _p(hist, C.c_int32), _p(ind, C.c_int32))
        return cnt, hist, ind

    def search_for_triangulation(self, desc1, node1, px1, desc2, node2, px2, E12, th_low=65, epipolar_dsqr=1e-4, cam=None):
        cam = cam or self.camera()
        d1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32); d2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
        n1a = np.ascontiguousarray(node1, np.int32); n2a = np.ascontiguousarray(node2, np.int32)
        p1 = np.ascontiguousarray(px1, np.float64); p2 = np.ascontiguousarray(px2, np.float64)
        E = np.ascontiguousarray(E12, np.float64).reshape(9)
        m = np.empty(max(len(d1), 1), np.int32)
        self.lib.yo_search_for_triangulation.argtypes = [C.POINTER(Camera), C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_double),
                                                         C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_int,
                                                         C.POINTER(C.c_double), C.c_int, C.c_double, C.POINTER(C.c_int32)]
        cnt = self.lib.yo_search_for_triangulation(C.byref(cam), _u8(d1), _p(n1a, C.c_int32), _f64(p1), len(d1), _u8(d2), _p(n2a, C.c_int32),
                                                   _f64(p2), len(d2), _f64(E), int(th_low), float(epipolar_dsqr), _p(m, C.c_int32))
        return m[:len(d1)], cnt

    # ---- ceres-side rows (oracle/ceres_ba.c) ----
    def ceres_edge(self, pose_taa, pt, obs_n):
        pose = np.ascontiguousarray(pose_taa, np.float64)
        pt = np.ascontiguousarray(pt, np.float64)
        obs_n = np.ascontiguousarray(obs_n, np.float64)
        r, Jx, Jp, z = np.empty(2), np.empty(12), np.empty(6), C.c_double()
        self.lib.yo_ceres_edge(_f64(pose), _f64(pt), _f64(obs_n), _f64(r), _f64(Jx), _f64(Jp), C.byref(z))
        return r, Jp.reshape(2, 3), Jx.reshape(2, 6), z.value

    def ceres_rotate_point(self, aa, p):
        aa = np.ascontiguousarray(aa, np.float64)
        p = np.ascontiguousarray(p, np.float64)
        out = np.empty(3)
        self.lib.yo_ceres_rotate_point(_f64(aa), _f64(p), _f64(out))
        return out

    def _ceres_problem(self, poses, pose_fixed, points, point_fixed, edge_pose, edge_point, obs_n, edge_huber,
                       edge_enable, fail_behind):
        keep = dict(poses=np.ascontiguousarray(poses, np.float64).reshape(-1, 6).copy(),
                    points=np.ascontiguousarray(points, np.float64).reshape(-1, 3).copy(),
                    edge_pose=np.ascontiguousarray(edge_pose, np.int32), edge_point=np.ascontiguousarray(edge_point, np.int32),
                    obs_n=np.ascontiguousarray(obs_n, np.float64))
        for k, v in (("pose_fixed", pose_fixed), ("point_fixed", point_fixed), ("edge_enable", edge_enable)):
            keep[k] = None if v is None else np.ascontiguousarray(v, np.uint8)
        keep["edge_huber"] = None if edge_huber is None else np.ascontiguousarray(edge_huber, np.float64)
        nul8, nulf = C.POINTER(C.c_uint8)(), C.POINTER(C.c_double)()
        pb = CeresProblem(len(keep["poses"]), len(keep["points"]), len(keep["edge_pose"]), _f64(keep["poses"]),
                          nul8 if keep["pose_fixed"] is None else _u8(keep["pose_fixed"]), _f64(keep["points"]),
                          nul8 if keep["point_fixed"] is None else _u8(keep["point_fixed"]),
                          _p(keep["edge_pose"], C.c_int32), _p(keep["edge_point"], C.c_int32), _f64(keep["obs_n"]),
                          nulf if keep["edge_huber"] is None else _f64(keep["edge_huber"]),
                          nul8 if keep["edge_enable"] is None else _u8(keep["edge_enable"]), int(bool(fail_behind)))
        return pb, keep

    def ceres_linearize(self, poses, pose_fixed, points, edge_pose, edge_point, obs_n, point_fixed=None, edge_huber=None,
                        edge_enable=None, fail_behind=False):
        pb, keep = self._ceres_problem(poses, pose_fixed, points, point_fixed, edge_pose, edge_point, obs_n, edge_huber,
                                       edge_enable, fail_behind)
        K, P, E = pb.n_poses, pb.n_points, pb.n_edges
        out = dict(Hpp=np.empty((K, 6, 6)), bp=np.empty((K, 6)), Hll=np.empty((P, 3, 3)), bl=np.empty((P, 3)),
                   Hpl=np.empty((E, 6, 3)), Jx=np.empty((E, 2, 6)), Jp=np.empty((E, 2, 3)), res=np.empty((E, 2)))
        cost = C.c_double()
        out["rc"] = self.lib.yo_ceres_linearize(C.byref(pb), _f64(keep["poses"]), _f64(keep["points"]), C.byref(cost),
                                                _f64(out["Hpp"]), _f64(out["bp"]), _f64(out["Hll"]), _f64(out["bl"]),
                                                _f64(out["Hpl"]), _f64(out["Jx"]), _f64(out["Jp"]), _f64(out["res"]))
        out["cost"] = cost.value
        return out

    def ceres_options(self, **kw):
        o = CeresOptions()
        self.lib.yo_ceres_default_options(C.byref(o))
        for k, v in kw.items():
            setattr(o, k, v)
        return o

    def ceres_solve(self, poses, pose_fixed, points, edge_pose, edge_point, obs_n, point_fixed=None, edge_huber=None,
                    edge_enable=None, fail_behind=False, options=None):
        """ceres::Solve restatement; returns (poses, points, summary dict)."""
        pb, keep = self._ceres_problem(poses, pose_fixed, points, point_fixed, edge_pose, edge_point, obs_n, edge_huber,
                                       edge_enable, fail_behind)
        opt = options or self.ceres_options()
        sm = CeresSummary()
        rc = self.lib.yo_ceres_solve(C.byref(pb), C.byref(opt), C.byref(sm))
        summary = {k: getattr(sm, k) for k, _ in CeresSummary._fields_}
        summary["rc"] = rc
        return keep["poses"], keep["points"], summary

    def g2o_lm(self, poses, pose_fixed, points, edge_pose, edge_point, obs, cam=None, huber_delta=5.991, max_iterations=20):
        """OptimizationAlgorithmLevenberg restatement around yo_ba_linearize; returns (poses, points, stats dict)."""
        cam = cam or self.camera()
        poses = np.ascontiguousarray(poses, np.float64).copy()
        pose_fixed = np.ascontiguousarray(pose_fixed, np.uint8)
        points = np.ascontiguousarray(points, np.float64).copy()
        edge_pose = np.ascontiguousarray(edge_pose, np.int32)
        edge_point = np.ascontiguousarray(edge_point, np.int32)
        obs = np.ascontiguousarray(obs, np.float64)
        pb = BaProblem(len(poses), len(points), len(edge_pose), _f64(poses), _u8(pose_fixed), _f64(points),
                       _p(edge_pose, C.c_int32), _p(edge_point, C.c_int32), _f64(obs),
                       float(cam.fx), float(cam.fy), float(cam.cx), float(cam.cy), float(huber_delta))
        st = LmStats()
        self.lib.yo_g2o_lm(C.byref(pb), _f64(poses), _f64(points), int(max_iterations), C.byref(st))
        return poses, points, {k: getattr(st, k) for k, _ in LmStats._fields_}

    def optimize_current_pose_only(self, pose_taa, px, pw, cam=None):
        """ba::OptimizeCurrentPoseOnly; returns (pose [t;aa], bad[n], depth[n], inliers, rounds)."""
        cam = cam or self.camera()
        pose = np.ascontiguousarray(pose_taa, np.float64).copy()
        px = np.ascontiguousarray(px, np.float64)
        pw = np.ascontiguousarray(pw, np.float64)
        n = len(px)
        bad, depth, rounds = np.zeros(max(n, 1), np.uint8), np.full(max(n, 1), np.nan), C.c_int()
        inl = self.lib.yo_optimize_current_pose_only(C.byref(cam), _f64(pose), n, _f64(px), _f64(pw), _u8(bad), _f64(depth),
                                                     C.byref(rounds))
        return pose, bad[:n], depth[:n], int(inl), rounds.value

    def two_view_ba_ceres(self, T_ref, T_cur, px_ref, px_cur, inlier, pts_ref, cam=None):
        """ba::TwoViewBACeres; returns (T_cur, inlier, pts_ref, summary dict)"""
        cam = cam or self.camera()
        Tr, Tc = SE3.from_array(T_ref), SE3.from_array(T_cur)
        pr = np.ascontiguousarray(px_ref, np.float64).reshape(-1, 2); pc = np.ascontiguousarray(px_cur, np.float64).reshape(-1, 2)
        inl = np.ascontiguousarray(inlier, np.uint8).copy(); pts = np.ascontiguousarray(pts_ref, np.float64).reshape(-1, 3).copy()
        sm = CeresSummary()
        self.lib.yo_two_view_ba_ceres(C.byref(cam), C.byref(Tr), C.byref(Tc), len(pr), _f64(pr), _f64(pc), _u8(inl), _f64(pts), C.byref(sm))
        return Tc.to_array(), inl.astype(bool), pts, {k: getattr(sm, k) for k, _ in CeresSummary._fields_}

    def _kf_obs(self, kf_T, obs_off, obs_kf, obs_px):
        Ts = (SE3 * max(len(kf_T), 1))(*[SE3.from_array(t) for t in kf_T])
        return (Ts, np.ascontiguousarray(obs_off, np.int32), np.ascontiguousarray(obs_kf, np.int32),
                np.ascontiguousarray(obs_px, np.float64).reshape(-1, 2))

    def optimize_current(self, T_cur, px, feat_point, points, kf_T, obs_off, obs_kf, obs_px, cam=None):
        """ba::OptimizeCurrent; returns (T_cur, points, bad, depth, inliers)"""
        cam = cam or self.camera()
        Tc = SE3.from_array(T_cur)
        px = np.ascontiguousarray(px, np.float64).reshape(-1, 2); fp = np.ascontiguousarray(feat_point, np.int32)
        pts = np.ascontiguousarray(points, np.float64).reshape(-1, 3).copy()
        Ts, oo, ok, op = self._kf_obs(kf_T, obs_off, obs_kf, obs_px)
        n = len(px)
        bad = np.zeros(max(n, 1), np.uint8); depth = np.full(max(n, 1), np.nan); sm = CeresSummary()
        inl = self.lib.yo_optimize_current(C.byref(cam), C.byref(Tc), n, _f64(px), _p(fp, C.c_int32), len(pts), _f64(pts), len(kf_T), Ts,
                                           _p(oo, C.c_int32), _p(ok, C.c_int32), _f64(op), _u8(bad), _f64(depth), C.byref(sm))
        return Tc.to_array(), pts, bad[:n].astype(bool), depth[:n], int(inl)

    def optimize_current_point_only(self, T_cur, px, feat_point, feat_bad, points, kf_T, obs_off, obs_kf, obs_px, cam=None):
        """ba::OptimizeCurrentPointOnly; returns the points"""
        cam = cam or self.camera()
        Tc = SE3.from_array(T_cur)
        px = np.ascontiguousarray(px, np.float64).reshape(-1, 2); fp = np.ascontiguousarray(feat_point, np.int32)
        fb = np.ascontiguousarray(feat_bad, np.uint8)
        pts = np.ascontiguousarray(points, np.float64).reshape(-1, 3).copy()
        Ts, oo, ok, op = self._kf_obs(kf_T, obs_off, obs_kf, obs_px)
        sm = CeresSummary()
        self.lib.yo_optimize_current_point_only(C.byref(cam), C.byref(Tc), len(px), _f64(px), _p(fp, C.c_int32), _u8(fb), len(pts), _f64(pts),
                                                len(kf_T), Ts, _p(oo, C.c_int32), _p(ok, C.c_int32), _f64(op), C.byref(sm))
        return pts

    def ba_pose_oplus(self, pose, upd):
        pose = np.ascontiguousarray(pose, np.float64).copy()
        upd = np.ascontiguousarray(upd, np.float64)
        self.lib.yo_ba_pose_oplus(_f64(pose), _f64(upd))
        return pose
