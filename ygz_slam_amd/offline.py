"""Python binding of the batched offline run (BASELINE.json configs[4]) -- the DRIVER is C++: ygz_slam_amd/host/ygz_offline.cpp in
libygz_host.so (include/ygz_offline.h), which owns the shard, the chunk plan, the lanes, window readiness, the resident-LM launches and the
two exchanges (RCCL directly; a host hook over gloo where two test ranks share one GPU).  This module is harness code for tests/ and
bench.py: it renders or hands over page-locked frame buffers, calls ygz_offline_run through ctypes and turns the result arrays into the
dictionaries the tests compare.  Rounds 2-4 drove the run from an interpreter loop here (864 lines, a dozen ABI calls per chunk through
ctypes); that loop is gone.

What runs per frame pair (cur = i, ref = i - 1), all pairs of a chunk per launch, mirrors VisualOdometry::AddFrame in state VO_GOOD
(src/Module/VisualOdometry.cpp:62-93):
    Frame::InitFrame + FeatureDetector::Detect                  (Frame.cpp:22-40, FeatureDetector.cpp:345-444)
    cv::BFMatcher(crossCheck) + the good-match filter           (test/test_orb_match.cpp:86-104)
    Tracker::TrackKLT from the reference keypoints              (Tracker.cpp:65-113)
    TrackRefFrame = Matcher::SparseImageAlignment               (VisualOdometry.cpp:281-302, Matcher.cpp:468-492)
    TrackLocalMap = FindCandidates + ProjectMapPoints (FindDirectProjection) + OptimizeCurrentPoseOnly
                                                                (LocalMapping.cpp:24-146, BA.cpp:188-264)
Every pair starts from T_ref = identity, so its result T_rel (pose of cur in the frame of ref) is a function of the two frames alone: a
shard needs no pose from its neighbour, and the global trajectory T[i] = T_rel[i] * T[i-1] is chained after the all-gather, identically
on every rank.  The reference gets Feature::_depth from map points made by its initialiser / triangulation (out of scope);
here the sequence supplies a depth image per frame (RGB-D style), sampled at the keypoints ON THE DEVICE.

BA round (LocalMapping::LocalBA -> ba::LocalBAG2O, LocalMapping.cpp:149-208, BA.cpp:386-543): keyframes are every `kf_stride`-th frame, a
window = `window_kfs` consecutive keyframes owned by the rank that owns its first keyframe (the anchor, held fixed like keyframe 0 at
BA.cpp:404); map points = the anchor's features with depth; observations (obs_mode "direct") = what LocalMapping::ProjectMapPoints leaves in
a keyframe (LocalMapping.cpp:47-120) or (obs_mode "match") the good cross-checked Hamming matches of the anchor's descriptors; after
optimize(20) the edges with chi2 > 5.991 are counted as BA.cpp:503-515 does.  Windows are built ON THE DEVICE from a store of keyframe rows
(csrc/window.hip) in the gauge of their anchor, as soon as the chunk holding their last keyframe has been enqueued.

This module never touches oracle/.
"""
import ctypes as C
import os as _os
import numpy as np

from . import dist as ydist

I7 = np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0])


# ---- SE3 on 7-vectors (qx,qy,qz,qw,tx,ty,tz); same formulas as thirdparty/Sophus/sophus/{so3,se3}.cpp ----------------
def _qmul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def _qrot(q, v):
    u = q[:3]
    t = 2.0 * np.cross(u, v)
    return v + q[3] * t + np.cross(u, t)


def se3_mul(A, B):
    q = _qmul(A[:4], B[:4])
    q = q / np.sqrt(np.dot(q, q))
    return np.concatenate([q, _qrot(A[:4], B[4:]) + A[4:]])


def se3_inv(A):
    qi = np.array([-A[0], -A[1], -A[2], A[3]])
    return np.concatenate([qi, -_qrot(qi, A[4:])])


def se3_act(A, p):
    p = np.asarray(p, np.float64)
    if p.ndim == 1:
        return _qrot(A[:4], p) + A[4:]
    u = A[:3]
    t = 2.0 * np.cross(u[None, :], p)
    return p + A[3] * t + np.cross(u[None, :], t) + A[4:]


def so3_log(q):
    n2 = float(np.dot(q[:3], q[:3]))
    n = np.sqrt(n2)
    w = q[3]
    if n < 1e-10:
        two_atan = 2.0 / w - 2.0 * n2 / (w * w * w)
    elif abs(w) < 1e-10:
        two_atan = (np.pi if w > 0 else -np.pi) / n
    else:
        two_atan = 2.0 * np.arctan(n / w) / n
    return two_atan * q[:3]


def se3_log_g2o(T):
    """[omega; upsilon] -- the estimate order of VertexSE3Sophus (G2oTypes.h:88, BA.cpp:407-409)"""
    om = so3_log(T[:4])
    th = np.sqrt(np.dot(om, om))
    Om = np.array([[0, -om[2], om[1]], [om[2], 0, -om[0]], [-om[1], om[0], 0]])
    if th < 1e-10:
        Vi = np.eye(3) - 0.5 * Om + (1.0 / 12.0) * (Om @ Om)
    else:
        Vi = np.eye(3) - 0.5 * Om + (1 - th / (2 * np.tan(th / 2))) / (th * th) * (Om @ Om)
    return np.concatenate([om, Vi @ T[4:]])


def se3_exp_g2o(v):
    """inverse of se3_log_g2o: [omega; upsilon] -> 7-vector"""
    om, ups = np.asarray(v[:3], np.float64), np.asarray(v[3:], np.float64)
    th = np.sqrt(np.dot(om, om))
    Om = np.array([[0, -om[2], om[1]], [om[2], 0, -om[0]], [-om[1], om[0], 0]])
    if th < 1e-10:
        q = np.array([0.5 * om[0], 0.5 * om[1], 0.5 * om[2], 1.0])
        V = np.eye(3) + 0.5 * Om
    else:
        s = np.sin(th / 2) / th
        q = np.array([s * om[0], s * om[1], s * om[2], np.cos(th / 2)])
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * Om + (th - np.sin(th)) / th ** 3 * (Om @ Om)
    q = q / np.sqrt(np.dot(q, q))
    return np.concatenate([q, V @ ups])


def chain(T_rel):
    """T[0] = identity (the first frame defines the world); T[i] = T_rel[i] * T[i-1]: host code of the library (ygz_hip_se3_chain; the
    same Sophus product as the device's window chains).  chain_py is the interpreter form (1 ms per thousand poses) kept as its check."""
    from . import _lib
    return _lib.se3_chain(T_rel)


def chain_py(T_rel):
    out = np.empty_like(T_rel)
    out[0] = I7
    bx, by, bz, bw, px, py, pz = (float(v) for v in I7)
    rows = T_rel.tolist()
    for i in range(1, len(rows)):
        ax, ay, az, aw, tx, ty, tz = rows[i]
        qx = aw * bx + ax * bw + ay * bz - az * by
        qy = aw * by - ax * bz + ay * bw + az * bx
        qz = aw * bz + ax * by - ay * bx + az * bw
        qw = aw * bw - ax * bx - ay * by - az * bz
        n = (qx * qx + qy * qy + qz * qz + qw * qw) ** 0.5
        c0, c1, c2 = 2.0 * (ay * pz - az * py), 2.0 * (az * px - ax * pz), 2.0 * (ax * py - ay * px)       # 2 u x p
        nx = px + aw * c0 + (ay * c2 - az * c1) + tx
        ny = py + aw * c1 + (az * c0 - ax * c2) + ty
        nz = pz + aw * c2 + (ax * c1 - ay * c0) + tz
        bx, by, bz, bw, px, py, pz = qx / n, qy / n, qz / n, qw / n, nx, ny, nz
        out[i] = (bx, by, bz, bw, px, py, pz)
    return out


# ---- windows ---------------------------------------------------------------------------------------------------------
def keyframes(n_total, kf_stride):
    return list(range(0, n_total, kf_stride))


def ba_windows(n_total, kf_stride, window_kfs):
    """non-overlapping windows of `window_kfs` consecutive keyframes (a trailing window needs >= 2 keyframes)"""
    kfs = keyframes(n_total, kf_stride)
    out = [kfs[a:a + window_kfs] for a in range(0, len(kfs), window_kfs)]
    return [w for w in out if len(w) >= 2]


def frame_owner(frame, n_total, world):
    for r in range(world):
        s, c, _ = ydist.shard_frames(n_total, r, world)
        if s <= frame < s + c:
            return r
    raise ValueError(frame)


def depth_image(d, div=1, dtype=np.float64, scale=1.0 / 5000.0):
    """depth map of the sequence (metres) -> the image the device samples: every div-th sample as float64 / float32 metres or, for
    uint16, round(depth / scale) (TUM RGB-D: scale = 1 / 5000)"""
    d = np.asarray(d)[::div, ::div]
    if np.dtype(dtype) == np.uint16:
        return np.clip(np.rint(d / scale), 0, 65535).astype(np.uint16)
    return np.ascontiguousarray(d, dtype)


def depth_at(dimg, px, w, h, scale=1.0 / 5000.0):
    """what ygz_hip_keypoint_depths_from_image reads for level-0 pixels px [n, 2] of a w x h frame from depth image dimg
    (host restatement of the look-up: the tests' reference and the oracle legs' input)"""
    dh, dw = dimg.shape
    ix = (px[:, 0].astype(np.int64) * dw) // w
    iy = (px[:, 1].astype(np.int64) * dh) // h
    v = dimg[iy, ix]
    d = v.astype(np.float64) * scale if dimg.dtype == np.uint16 else v.astype(np.float64)
    return np.where(d > 0, d, 0.0)


# ---- ctypes mirror of include/ygz_offline.h ------------------------------------------------------------------------------------------------
class OffParams(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("levels", C.c_int), ("n_frames", C.c_int), ("rank", C.c_int), ("world", C.c_int),
                ("device", C.c_int), ("chunk", C.c_int), ("kf_stride", C.c_int), ("window_kfs", C.c_int), ("max_points", C.c_int),
                ("ba_iterations", C.c_int), ("lanes", C.c_int), ("lm_group", C.c_int), ("obs_mode", C.c_int), ("ba_rounds", C.c_int),
                ("outlier_chi2", C.c_double), ("frame_channels", C.c_int), ("depth_w", C.c_int), ("depth_h", C.c_int), ("depth_kind", C.c_int),
                ("depth_scale", C.c_double), ("pipeline_ba", C.c_int), ("defer_gaps", C.c_int), ("ramp", C.c_int), ("kf_tail", C.c_int),
                ("stage_overlap", C.c_int), ("bg_team_budget", C.c_int), ("bg_team_spread", C.c_int)]


ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
SEND_RECV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int)
CHUNK_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int32))


class OffExchange(C.Structure):
    _fields_ = [("user", C.c_void_p), ("all_gather", ALL_GATHER_FN), ("send_recv", SEND_RECV_FN)]


class OffResults(C.Structure):
    _fields_ = [("n_frames", C.c_int), ("first_frame", C.c_int), ("n_own", C.c_int), ("T_rel", C.POINTER(C.c_double)), ("trajectory", C.POINTER(C.c_double)),
                ("summary", C.POINTER(C.c_double)), ("n_kp", C.POINTER(C.c_int32)), ("n_windows", C.c_int), ("state_doubles", C.c_int),
                ("window_state", C.POINTER(C.c_double)), ("window_owner", C.POINTER(C.c_int32)), ("window_kfs", C.POINTER(C.c_int32)),
                ("n_chunks", C.c_int), ("lm_launches", C.c_int), ("lm_retries", C.c_int), ("n_degenerate", C.c_int),
                ("ms_track", C.c_double), ("ms_gather", C.c_double), ("ms_ba_tail", C.c_double), ("ms_exchange", C.c_double), ("backend", C.c_int)]


OFFLINE_SYMBOLS = ["ygz_offline_default_params", "ygz_offline_shard", "ygz_offline_plan", "ygz_offline_plan_range", "ygz_offline_ragged_all_gather",
                   "ygz_offline_rccl_unique_id", "ygz_offline_create", "ygz_offline_destroy", "ygz_offline_last_error", "ygz_offline_run", "ygz_offline_track",
                   "ygz_offline_gather", "ygz_offline_ba_round", "ygz_offline_get_results", "ygz_offline_set_chunk_callback", "ygz_offline_contexts",
                   "ygz_offline_owned_windows", "ygz_offline_build_windows", "ygz_offline_retry_windows"]
HOST_LIB_PATH = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "libygz_host.so")
_host = None


def host_lib():
    """libygz_host.so (the C++ class surfaces + the offline driver); raises if it has not been built"""
    global _host
    if _host is None:
        from . import _lib
        _lib.load()                                            # libygz_hip.so first (and torch's HIP runtime before it, see _lib.load)
        if not _os.path.exists(HOST_LIB_PATH):
            raise ImportError("libygz_host.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _host = C.CDLL(HOST_LIB_PATH)
        _host.ygz_offline_last_error.restype = C.c_char_p
        _host.ygz_offline_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _host.ygz_offline_track.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        for f in ("ygz_offline_gather", "ygz_offline_ba_round"):
            getattr(_host, f).argtypes = [C.c_void_p]
        _host.ygz_offline_destroy.argtypes = [C.c_void_p]
        _host.ygz_offline_destroy.restype = None
        _host.ygz_offline_last_error.argtypes = [C.c_void_p]
        _host.ygz_offline_get_results.argtypes = [C.c_void_p, C.POINTER(OffResults)]
        _host.ygz_offline_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(OffParams), C.c_void_p, C.POINTER(OffExchange)]
        _host.ygz_offline_set_chunk_callback.argtypes = [C.c_void_p, CHUNK_FN, C.c_void_p]
        _host.ygz_offline_contexts.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]
        _host.ygz_offline_owned_windows.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int)]
        _host.ygz_offline_build_windows.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        _host.ygz_offline_retry_windows.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int]
        _host.ygz_offline_ragged_all_gather.argtypes = [C.POINTER(OffExchange), C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_size_t, C.c_void_p, C.c_void_p]
        _host.ygz_offline_rccl_unique_id.argtypes = [C.c_void_p]
    return _host


def gloo_exchange(pg=None):
    """the two exchange primitives of include/ygz_offline.h over torch.distributed on host memory (gloo): what the tests hand to the C++
    driver where RCCL cannot run (two ranks on one GPU, no GPU at all).  Returns (OffExchange, keep-alive tuple)."""
    import torch
    import torch.distributed as dist

    def _view(ptr, n):
        return torch.from_numpy(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n,)))

    def all_gather(_user, send, recv, nbytes):
        try:
            world = dist.get_world_size(pg)
            dist.all_gather(list(_view(recv, nbytes * world).view(world, nbytes).unbind(0)), _view(send, nbytes), group=pg)
            return 0
        except Exception as e:                                   # an exception must not cross the C boundary
            print("gloo_exchange.all_gather: %r" % (e,))
            return 1

    def send_recv(_user, buf, nbytes, src, dst):
        try:
            t = _view(buf, nbytes)
            if dist.get_rank(pg) == src:
                dist.send(t, dst, group=pg)
            else:
                dist.recv(t, src, group=pg)
            return 0
        except Exception as e:
            print("gloo_exchange.send_recv: %r" % (e,))
            return 1
    ag, sr = ALL_GATHER_FN(all_gather), SEND_RECV_FN(send_recv)
    return OffExchange(None, ag, sr), (ag, sr)


def ragged_all_gather(local, counts, pg=None):
    """ygz_offline_ragged_all_gather over gloo: rank r contributes counts[r] rows (`local`, a C-contiguous numpy array [counts[rank], ...]);
    returns the concatenation in rank order.  One collective of blocks padded to max(counts) rows -- the exchange the C++ driver performs
    for T_rel and for the window states."""
    import torch.distributed as dist
    hook, keep = gloo_exchange(pg)
    rank, world = dist.get_rank(pg), dist.get_world_size(pg)
    local = np.ascontiguousarray(local)
    row_bytes = int(np.prod(local.shape[1:])) * local.dtype.itemsize
    cnt = np.ascontiguousarray(counts, np.int32)
    assert len(cnt) == world and local.shape[0] == cnt[rank]
    full = np.zeros((int(cnt.sum()),) + tuple(local.shape[1:]), local.dtype)
    rc = host_lib().ygz_offline_ragged_all_gather(C.byref(hook), rank, world, cnt.ctypes.data_as(C.POINTER(C.c_int32)), row_bytes,
                                                 C.c_void_p(local.ctypes.data), C.c_void_p(full.ctypes.data))
    if rc != 0:
        raise RuntimeError("ygz_offline_ragged_all_gather failed: %d" % rc)
    del keep
    return full


_DEPTH_KIND = {np.dtype(np.float32): 0, np.dtype(np.uint16): 1, np.dtype(np.float64): 2}


class OfflineVO:
    """One rank of the offline run: a handle of the C++ driver (ygz_offline_create) plus the page-locked frame buffers it reads.
    frame_source(i) -> BGR uint8 [h, w, 3] (or gray [h, w] with gray=True); depth_source(i) -> depth map [h, w] (metres): rendered into
    page-locked memory before the run.  block_source(frames) -> (frames [n, h, w(, 3)] uint8, depth images [n, dh, dw]) replaces them when the
    caller already holds the sequence in page-locked memory in the form the ABI uploads: it is asked ONCE for all frames this rank needs and
    must return views of contiguous page-locked arrays.  keep=True: after every chunk everything a parity test wants to look at is read from
    the chunk's lane through the C ABI (ygz_offline_set_chunk_callback)."""

    def __init__(self, width, height, n_total, rank=0, world=1, device=0, chunk=128, levels=3, kf_stride=8, window_kfs=8,
                 max_points=2000, ba_iterations=20, overlap=False, process_group=None, exchange_on_device=True, keep=False, lanes=3,
                 depth_div=1, depth_dtype=np.float64, depth_scale=1.0 / 5000.0, pipeline_ba=True, lm_group=None,
                 obs_mode="direct", ba_rounds=1, outlier_chi2=5.991, gray=False, defer_gaps=None, bg_team_budget=0, rccl_single=False, bg_team_spread=True):
        from . import _lib
        self.lib, self.h_lib = _lib, host_lib()
        assert obs_mode in ("direct", "match") and ba_rounds in (1, 2)
        self.w, self.h, self.levels = width, height, levels
        self.n_total, self.rank, self.world, self.device = n_total, rank, world, device
        self.chunk, self.kf_stride, self.window_kfs, self.max_points = chunk, kf_stride, window_kfs, max_points
        self.pg, self.keep, self.gray, self.obs_mode = process_group, keep, gray, obs_mode
        self.depth_div, self.depth_dtype, self.depth_scale = depth_div, np.dtype(depth_dtype), depth_scale
        self.start, self.count, self.halo = ydist.shard_frames(n_total, rank, world)
        self.wins = ba_windows(n_total, kf_stride, window_kfs)
        self.owner = [frame_owner(w[0], n_total, world) for w in self.wins]
        self.mine = [i for i, o in enumerate(self.owner) if o == rank]
        self.dh, self.dw = -(-height // depth_div), -(-width // depth_div)
        p = OffParams()
        self.h_lib.ygz_offline_default_params(C.byref(p))
        p.width, p.height, p.levels, p.n_frames, p.rank, p.world, p.device = width, height, levels, n_total, rank, world, device
        p.chunk, p.kf_stride, p.window_kfs, p.max_points, p.ba_iterations = chunk, kf_stride, window_kfs, max_points, ba_iterations
        p.lanes, p.lm_group, p.obs_mode, p.ba_rounds, p.outlier_chi2 = lanes, int(lm_group or 0), 1 if obs_mode == "direct" else 0, ba_rounds, outlier_chi2
        p.frame_channels, p.depth_w, p.depth_h, p.depth_kind, p.depth_scale = (1 if gray else 3), self.dw, self.dh, _DEPTH_KIND[self.depth_dtype], depth_scale
        p.pipeline_ba, p.defer_gaps, p.stage_overlap, p.bg_team_budget = int(pipeline_ba), (-1 if defer_gaps is None else int(defer_gaps)), int(overlap), int(bg_team_budget)
        p.bg_team_spread = int(bool(bg_team_spread))
        self.params = p
        self._keepalive = []
        hook_p, rccl_id = None, None
        if world > 1:
            import torch.distributed as dist
            if exchange_on_device:
                # RCCL directly from the C++ driver; torch.distributed only carries the 128-byte ncclUniqueId from rank 0 to the others (bootstrap)
                ident = [None]
                if rank == 0:
                    buf = C.create_string_buffer(128)
                    rc = self.h_lib.ygz_offline_rccl_unique_id(buf)
                    if rc != 0:
                        raise RuntimeError("ygz_offline_rccl_unique_id failed: %d (librccl.so.1 not loadable?)" % rc)
                    ident = [bytes(buf.raw)]
                dist.broadcast_object_list(ident, src=0, group=process_group)
                rccl_id = C.create_string_buffer(ident[0], 128)
            else:
                hook, ka = gloo_exchange(process_group)
                self._keepalive += [hook, ka]
                hook_p = C.byref(hook)
        elif rccl_single:                                         # a communicator of one rank: the RCCL path of the driver on a single GPU (tests)
            rccl_id = C.create_string_buffer(128)
            rc = self.h_lib.ygz_offline_rccl_unique_id(rccl_id)
            if rc != 0:
                raise RuntimeError("ygz_offline_rccl_unique_id failed: %d (librccl.so.1 not loadable?)" % rc)
        self._h = C.c_void_p()
        rc = self.h_lib.ygz_offline_create(C.byref(self._h), C.byref(p), rccl_id, hook_p)
        if rc != 0:
            self._h = C.c_void_p()
            raise _lib.YgzHipError(rc, "ygz_offline_create")
        ba, nl = C.c_void_p(), C.c_int(0)
        lanes_p = (C.c_void_p * 16)()
        self._chk(self.h_lib.ygz_offline_contexts(self._h, C.byref(ba), lanes_p, 16, C.byref(nl)), "contexts")
        mk = lambda ptr: _lib.HipContext.from_handle(ptr, width, height, levels)
        self.ba = mk(ba.value)
        self.lanes = [mk(lanes_p[i]) for i in range(nl.value)]
        self.ctx = self.lanes[0]
        self.S = 6 * window_kfs + 3 * max_points + 12
        self.lm_group = lm_group
        self.timing, self.degenerate_windows, self.lm_retries = {}, [], 0
        self._pin = None
        self._rec_extra = {}
        if keep:
            self._cb = CHUNK_FN(self._on_chunk)
            self._chk(self.h_lib.ygz_offline_set_chunk_callback(self._h, self._cb, None), "set_chunk_callback")

    def _chk(self, rc, what):
        if rc != 0:
            msg = self.h_lib.ygz_offline_last_error(self._h) if self._h else b""
            raise RuntimeError("ygz_offline %s failed: %d (%s)" % (what, rc, (msg or b"").decode()))

    def close(self):
        if self._h:
            self.h_lib.ygz_offline_destroy(self._h)
            self._h = C.c_void_p()
        if self._pin is not None:
            for a in self._pin:
                a.free()
            self._pin = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ depth images
    def depth_image(self, d):
        return depth_image(d, self.depth_div, self.depth_dtype, self.depth_scale)

    def depth_at(self, dimg, px):
        return depth_at(dimg, px, self.w, self.h, self.depth_scale)

    # ------------------------------------------------------------------ frame buffers
    def _buffers(self, frame_source, depth_source, block_source):
        """(frames, depth images, sequence index of row 0): page-locked arrays covering [start - halo, start + count)"""
        need = list(range(self.start - self.halo, self.start + self.count))
        if block_source is not None:
            img, dimg = block_source(need)
            assert img.flags["C_CONTIGUOUS"] and dimg.flags["C_CONTIGUOUS"] and img.dtype == np.uint8 and dimg.dtype == self.depth_dtype
            assert img.shape == ((len(need), self.h, self.w) if self.gray else (len(need), self.h, self.w, 3)) and dimg.shape == (len(need), self.dh, self.dw)
            return img, dimg, need[0]
        if self._pin is None:
            self._pin = [self.lib.PinnedArray((len(need), self.h, self.w) if self.gray else (len(need), self.h, self.w, 3), np.uint8),
                         self.lib.PinnedArray((len(need), self.dh, self.dw), self.depth_dtype)]
        for k, f in enumerate(need):
            self._pin[0].array[k] = frame_source(f)
            self._pin[1].array[k] = self.depth_image(depth_source(f))
        return self._pin[0].array, self._pin[1].array, need[0]

    # ------------------------------------------------------------------ keep=True: everything a parity test wants to look at, per chunk
    def _on_chunk(self, _user, lane_ptr, _chunk, n_frames, frames_p, n_pairs, pairs_p):
        try:
            c = self.lib.HipContext.from_handle(lane_ptr, self.w, self.h, self.levels)
            frames = [frames_p[i] for i in range(n_frames)]
            proper = {pairs_p[2 * p] for p in range(n_pairs)} | {0}    # a frame of the chunk proper is the `cur` of a pair (or frame 0); the others are halo frames
            for k, f in enumerate(frames):
                kp = c.get_keypoints(k)
                kp["depth"], has_mp = c.get_keypoint_depths(k)
                assert np.array_equal(kp["depth"], self.depth_at(self._dimg[f - self._base], kp["px"])) and np.array_equal(has_mp, kp["depth"] > 0)   # the device's look-up
                if f in proper:
                    self._rec_extra.setdefault(f, {})["kp"] = kp
            for p in range(n_pairs):
                cur = pairs_p[2 * p]
                n_meas, T_sa, iters = c.track_get_pose(p)
                po = c.track_get_pose_only(p)
                good, n_good, min_dis = c.get_good_matches(p)
                idx, dist_ = c.get_matches(p)
                pts, st, err = c.track_get_klt(p)
                ok, pxd, lvl = c.track_get_direct(p)
                self._rec_extra.setdefault(cur, {}).update(
                    sa_iters=iters, m_idx=idx, m_dist=dist_, m_good=good, klt_pts=pts, klt_status=st, klt_err=err, fdp_ok=ok, fdp_px=pxd,
                    fdp_level=lvl, po_bad=po["bad"], po_pose=po["pose"],
                    _chk=(T_sa, po["T"], n_good, int(st.astype(bool).sum()), int(ok.sum()), int((idx >= 0).sum())))
        except BaseException as e:                                # (an exception must not cross the C boundary: re-raised after the run)
            self._cb_error = e

    # ------------------------------------------------------------------ results
    def _results(self):
        r = OffResults()
        self._chk(self.h_lib.ygz_offline_get_results(self._h, C.byref(r)), "get_results")
        n, nw, S = r.n_frames, r.n_windows, r.state_doubles
        arr = lambda p, shape, dt: np.ctypeslib.as_array(p, shape=shape).astype(dt, copy=True) if int(np.prod(shape)) else np.zeros(shape, dt)
        return dict(r=r, T_rel=arr(r.T_rel, (n, 7), np.float64), trajectory=arr(r.trajectory, (n, 7), np.float64), summary=arr(r.summary, (n, 32), np.float64),
                    n_kp=arr(r.n_kp, (n,), np.int32), state=arr(r.window_state, (nw, S), np.float64), owner=arr(r.window_owner, (nw,), np.int32),
                    kfs=arr(r.window_kfs, (nw, self.window_kfs), np.int32))

    def _records(self, R):
        rec = {}
        for f in range(self.start, self.start + self.count):
            r = dict(n_kp=int(R["n_kp"][f]))
            if f > 0:
                S = R["summary"][f]
                r.update(T_sa=S[0:7].copy(), sa_n_meas=int(S[7]), T_rel=S[24:31].copy(), po_inliers=int(S[14]), po_rounds=int(S[15]), n_match=int(S[16]),
                         n_good=int(S[17]), min_dis=float(S[18]), n_klt=int(S[19]), n_fdp=int(S[20]))
            ex = self._rec_extra.get(f)
            if ex:
                ex = dict(ex)
                chk = ex.pop("_chk", None)
                if chk is not None:                               # the 32-field summary equals the per-stage getters
                    assert np.array_equal(chk[0], r["T_sa"]) and np.array_equal(chk[1], r["T_rel"]) and chk[2] == r["n_good"]
                    assert chk[3] == r["n_klt"] and chk[4] == r["n_fdp"] and chk[5] == r["n_match"]
                r.update(ex)
            rec[f] = r
        return rec

    def _begin(self, frame_source, depth_source, block_source):
        img, dimg, base = self._buffers(frame_source, depth_source, block_source)
        self._img, self._dimg, self._base = img, dimg, base
        self._rec_extra, self._cb_error = {}, None
        return C.c_void_p(img.ctypes.data), C.c_void_p(dimg.ctypes.data), base

    def _after(self):
        if getattr(self, "_cb_error", None) is not None:
            raise self._cb_error

    # ------------------------------------------------------------------ phases
    def track_shard(self, frame_source, depth_source, block_source=None):
        """phase 1 only (ygz_offline_track): the hot path over this shard + the windows it completes; returns the per-frame records"""
        fp, dp, base = self._begin(frame_source, depth_source, block_source)
        self._chk(self.h_lib.ygz_offline_track(self._h, fp, dp, base), "track")
        self._after()
        return self._records(self._results())

    def _ba_launch(self, wis, optimize=True):
        """build (+ optimise) the given owned windows now (ygz_offline_build_windows)"""
        if wis:
            assert list(wis) == list(range(wis[0], wis[-1] + 1))
            self._chk(self.h_lib.ygz_offline_build_windows(self._h, self.mine.index(wis[0]), len(wis), int(optimize)), "build_windows")

    def retry_windows(self, slots):
        """rebuild the owned windows in BA slots `slots` and solve each with ONE workgroup (the retry of a timed-out LM team)"""
        a = np.ascontiguousarray(slots, np.int32)
        self._chk(self.h_lib.ygz_offline_retry_windows(self._h, a.ctypes.data_as(C.POINTER(C.c_int32)), len(a)), "retry_windows")
        self.lm_retries = self._results()["r"].lm_retries

    def run(self, frame_source, depth_source, block_source=None):
        """the whole run in ONE call of the C++ driver"""
        fp, dp, base = self._begin(frame_source, depth_source, block_source)
        self._chk(self.h_lib.ygz_offline_run(self._h, fp, dp, base), "run")
        self._after()
        R = self._results()
        r = R["r"]
        self.timing = {"track_shard": r.ms_track, "gather": r.ms_gather, "ba_round": r.ms_ba_tail + r.ms_exchange, "ba_tail_after_tracking": r.ms_ba_tail,
                       "ba_exchange_download": r.ms_exchange}
        self.lm_retries, self.backend = r.lm_retries, {0: "single rank", 1: "rccl", 2: "host hook (gloo)"}[r.backend]
        self.lm_launches, self.n_chunks = r.lm_launches, r.n_chunks
        K, P = self.window_kfs, self.max_points
        windows, dims, self.degenerate_windows = [], {}, []
        for wi, w in enumerate(self.wins):
            assert list(R["kfs"][wi][:len(w)]) == w and int(R["owner"][wi]) == self.owner[wi]        # the driver's windows are this module's
            st = R["state"][wi]
            tail = st[6 * K + 3 * P:]
            dims[wi] = (int(tail[0]), int(tail[1]), int(tail[2]))
            degenerate = bool(dims[wi][1] == 0 or dims[wi][2] == 0)
            if degenerate:
                self.degenerate_windows.append(wi)
            windows.append(dict(kfs=w, owner=self.owner[wi], poses=st[:len(w) * 6].reshape(len(w), 6).copy(), state=st.copy(),
                                stats=np.array([tail[5], tail[6], tail[3], tail[2]]),
                                inliers=dict(edges=int(tail[8]), outliers=int(tail[9]), chi2=float(tail[10]), chi2_inliers=float(tail[11])),
                                lm=dict(iterations=int(tail[3]), trials=int(tail[4]), degenerate=degenerate)))
        assert r.n_degenerate == len(self.degenerate_windows)
        res = dict(records=self._records(R), T_rel=R["T_rel"], trajectory=R["trajectory"], windows=windows, built=dims)
        res["keyframe_pose"] = _LazyKeyframePoses(windows, R["trajectory"])
        return res


class _LazyKeyframePoses(dict):
    """refined world poses of the keyframes: the windows live in the gauge of their anchor, world pose = T(anchor -> kf) * T(world ->
    anchor).  Composed on first use (128 small numpy products are 3 ms -- as much as the exchange of the whole BA round)"""

    def __init__(self, windows, traj):
        super().__init__()
        self._src = (windows, traj)

    def _fill(self):
        if self._src is not None:
            windows, traj = self._src
            self._src = None
            for w in windows:
                for k, f in enumerate(w["kfs"]):
                    dict.__setitem__(self, f, se3_mul(se3_exp_g2o(w["poses"][k]), traj[w["kfs"][0]]))

    def __getitem__(self, k):
        self._fill(); return dict.__getitem__(self, k)

    def __iter__(self):
        self._fill(); return dict.__iter__(self)

    def __len__(self):
        self._fill(); return dict.__len__(self)

    def items(self):
        self._fill(); return dict.items(self)

    def keys(self):
        self._fill(); return dict.keys(self)

    def values(self):
        self._fill(); return dict.values(self)

    def __reduce__(self):                                   # pickles as a plain dict
        self._fill(); return (dict, (dict(dict.items(self)),))


def window_pose_errors(windows, trajectory, gt):
    """For every non-anchor keyframe of every window: the error of its pose RELATIVE TO THE ANCHOR against the ground truth, before the BA round
    (the chained tracking result) and after it (the window's refined vertex).  gt / trajectory: [n_frames, 7] world poses.
    Returns dict(t_before, t_after [m], r_before, r_after [rad]) as arrays over those keyframes."""
    tb, ta, rb, ra = [], [], [], []

    def err(E, G):
        D = se3_mul(E, se3_inv(G))
        return float(np.linalg.norm(D[4:])), float(2.0 * np.arctan2(np.linalg.norm(D[:3]), abs(D[3])))
    for w in windows:
        a = w["kfs"][0]
        for k, f in enumerate(w["kfs"]):
            if k == 0:
                continue
            G = se3_mul(gt[f], se3_inv(gt[a]))
            e0 = err(se3_mul(trajectory[f], se3_inv(trajectory[a])), G)
            e1 = err(se3_exp_g2o(w["poses"][k]), G)
            tb.append(e0[0]); rb.append(e0[1]); ta.append(e1[0]); ra.append(e1[1])
    return dict(t_before=np.array(tb), t_after=np.array(ta), r_before=np.array(rb), r_after=np.array(ra))
