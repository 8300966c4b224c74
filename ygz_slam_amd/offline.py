"""Batched offline visual odometry (BASELINE.json configs[4]): a sequence of N frames sharded over `world` GPUs in
contiguous chunks with a one-frame halo, the per-frame hot path run for every consecutive pair, a local-BA round per
window of keyframes, and the two exchanges SURVEY 8e names: the refined BA-window states (map points + keyframe poses) from
their owners to every rank, and the all-gather of per-shard trajectories.

What runs per frame pair (cur = i, ref = i - 1), all pairs of a chunk per launch, mirrors VisualOdometry::AddFrame in state
VO_GOOD (src/Module/VisualOdometry.cpp:62-93):
    Frame::InitFrame + FeatureDetector::Detect                  (Frame.cpp:22-40, FeatureDetector.cpp:345-444)
    cv::BFMatcher(crossCheck) + the good-match filter           (test/test_orb_match.cpp:86-104)
    Tracker::TrackKLT from the reference keypoints              (Tracker.cpp:65-113)
    TrackRefFrame = Matcher::SparseImageAlignment               (VisualOdometry.cpp:281-302, Matcher.cpp:468-492)
    TrackLocalMap = FindCandidates + ProjectMapPoints (FindDirectProjection) + OptimizeCurrentPoseOnly
                                                                (LocalMapping.cpp:24-146, BA.cpp:188-264)
Every pair starts from T_ref = identity, so its result T_rel (pose of cur in the frame of ref) is a function of the two
frames alone: a shard needs no pose from its neighbour, and the global trajectory T[i] = T_rel[i] * T[i-1] is chained
after the all-gather, identically on every rank.  The reference gets Feature::_depth from map points made by its
initialiser / triangulation (out of scope); here the sequence supplies a depth image per frame (RGB-D style),
sampled at the keypoints ON THE DEVICE (ygz_hip_keypoint_depths_from_image).

BA round (LocalMapping::LocalBA -> ba::LocalBAG2O, LocalMapping.cpp:149-208, BA.cpp:386-543): keyframes are every
`kf_stride`-th frame, a window = `window_kfs` consecutive keyframes owned by the rank that owns its first keyframe (the
anchor, held fixed like keyframe 0 at BA.cpp:404); map points = the anchor's features with depth; observations (obs_mode "direct",
the default since round 4) = what LocalMapping::ProjectMapPoints leaves in a keyframe (LocalMapping.cpp:47-120): the map point projected
with the tracked pose, kept if in view (FindCandidates), refined to sub-pixel by Matcher::FindDirectProjection from the anchor's image --
the keyframe rows carry the pyramids for that -- or (obs_mode "match", rounds 2-3) the good cross-checked Hamming matches of the anchor's
descriptors in the other keyframes; after optimize(20) the edges with chi2 > 5.991 are counted as BA.cpp:503-515 does.  The windows are built ON
THE DEVICE from a store of keyframe rows (csrc/window.hip), in the gauge of their anchor (anchor pose = identity, the other
vertices chained from the frames' relative poses), so a window depends on its own frames only: it is built and optimised as soon
as the chunk holding its last keyframe has been enqueued -- beside the uploads and kernels of the following chunks -- and the
sharded run reproduces the unsharded one bit for bit.

The run is driven from ONE host thread: every call on the tracking path is asynchronous (page-locked buffers, staged tables), two
or three contexts ("lanes") take the chunks in turn so that the H2D copy of one chunk runs under the kernels of another (three: the
copy of the next chunk is already queued when a copy ends, PCIe never waits for the host), one more context owns the keyframe
store and the BA windows; the host blocks only when it re-uses a lane and reads that lane's 32-double-per-pair
summary.  The ORDER of the chunks is free (chunk_plan): a chunk is a tuple of frame ranges, each with its halo frame, and on long
shards the keyframe-free frames behind the last windows are processed at the very end, beside the last resident-LM launch.

This module is host logic over the C ABI (ygz_slam_amd._lib); it never touches oracle/.
"""
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


def exchange_rows(buf, owner, world, pg=None, via_host=None):
    """The map exchange: row i of `buf` (a torch tensor, in HBM on the GPU box) is owned by rank owner[i] -- owners hold contiguous
    row ranges, windows being ordered by anchor frame --; afterwards every rank holds every owner's rows.  ONE collective
    (all-gather of fixed-shape blocks; RCCL over xGMI with backend nccl, gloo on host copies in the CPU tests)."""
    if world == 1:
        return
    import torch.distributed as dist
    rank = dist.get_rank(pg)
    rows = [[i for i, o in enumerate(owner) if o == r] for r in range(world)]
    for r in rows:
        assert r == list(range(r[0], r[-1] + 1)) if r else True
    mine = rows[rank]
    local = buf[mine[0]:mine[-1] + 1] if mine else buf[:0]
    if via_host is None:
        via_host = buf.device.type == "cpu" or dist.get_backend(pg) == "gloo"
    g = ydist.all_gather_rows(local, [len(r) for r in rows], pg, via_host)
    for r in range(world):
        if r != rank and rows[r]:
            buf[rows[r][0]:rows[r][-1] + 1] = g[r, :len(rows[r])]


KF_SMALL = int(_os.environ.get("YGZ_OFF_KF_SMALL", "0"))     # frames of the chunk that ends with a shard's last keyframe (0: not cut; 8: -0.5 ms at 1024 frames, +1 ms at 128 / 512)


def chunk_schedule(first, last, chunk, ramp=True, kf_stride=0, kf_small=KF_SMALL, ramp_from=4):
    """[first, last) cut into chunks of `chunk` frames, with a ramp at both ends (chunk / 4, chunk / 2, chunk ... chunk, chunk / 2, chunk / 4)
    when there is room: nothing overlaps the upload of the first chunk or the kernels of the last one, so those two are kept short.
    kf_stride > 0: the frames behind the shard's last keyframe (they complete no BA window) form a chunk of their own at the very end, so
    that every window is complete one chunk earlier and the last resident-LM launch runs beside that chunk instead of after it"""
    n = last - first
    sizes = []
    if ramp and chunk >= 64 and n >= ramp_from * chunk:
        head = [chunk // 4, chunk // 2]
        tail = [chunk // 2, chunk // 4]
        body = n - sum(head) - sum(tail)
        sizes = head + [chunk] * (body // chunk) + ([body % chunk] if body % chunk else []) + tail
    else:
        sizes = [chunk] * (n // chunk) + ([n % chunk] if n % chunk else [])
    out, c0 = [], first
    for s_ in sizes:
        out.append((c0, c0 + s_)); c0 += s_
    assert c0 == last or n <= 0
    if kf_stride > 0 and out:
        a, b = out[-1]
        k_last = ((b - 1) // kf_stride) * kf_stride                # the last keyframe of the shard
        if a <= k_last and k_last + 1 < b and k_last + 1 > a:
            out[-1:] = [(a, k_last + 1), (k_last + 1, b)]
        # the chunk that ENDS with the last keyframe decides when the last resident-LM launch starts (upload of the keyframe -> the chunk's
        # kernels -> window build -> LM, nothing hides it): kept to kf_small frames -- a chunk of 8 frames passes through the kernels in
        # 1.6 ms, one of 26 in 2.7 (profiles/r04_offline_timeline_summary.md)
        for k, (a, b) in enumerate(out):
            if b == k_last + 1 and kf_small > 0 and b - a > kf_small + kf_small // 2:
                out[k:k + 1] = [(a, b - kf_small), (b - kf_small, b)]
                break
    return out


DEFER_GROUP = int(_os.environ.get("YGZ_OFF_DEFER_GROUP", "45"))      # frames per deferred chunk (about)
DEFER_LAST_MAIN = int(_os.environ.get("YGZ_OFF_LAST_MAIN", "0"))     # > 0: the last chunk of the main pass cut to this many frames (16: + 2 ms at 1024 frames -- one more chunk for the host to hand to a busy lane)


def chunk_plan(first, last, chunk, ramp, kf_stride, windows, defer, group=DEFER_GROUP, last_main=DEFER_LAST_MAIN):
    """The chunks of the shard [first, last) IN PROCESSING ORDER; a chunk is a tuple of frame ranges ((a, b), ...) -- normally one.  Every
    frame pair is solved from the identity, so the order is free; what it decides is when a BA window is complete (all frames from its anchor
    to its last keyframe tracked) and therefore where its resident-LM launch -- a latency chain of ~5 ms that uses a fraction of the GPU --
    falls.  The frames BEHIND a window's last keyframe (kf_stride - 1 of them, up to the next anchor) complete nothing: for the last `defer`
    windows that end inside the shard they are taken out of the main pass and processed at the very end, about `group` frames per chunk (a
    chunk of several short ranges: one small chunk per gap costs a pass of latency-bound kernels each), so that the last LM launch runs
    beside their uploads and kernels instead of after everything else; the last chunk of the main pass -- the LM waits for its kernels --
    can be cut to last_main frames (measured slower, off).  Cost: two more halo frames per deferred gap (the range after a gap and the gap itself each upload their
    predecessor once more)."""
    plain = [((a, b),) for a, b in chunk_schedule(first, last, chunk, ramp, kf_stride)]
    if defer <= 0:
        return plain
    inside = [w for w in windows if w[0] >= first and w[-1] < last]
    anchors = sorted(w[0] for w in windows)
    gaps = []
    for w in inside[-defer:]:
        nxt = [a for a in anchors if a > w[-1]]
        g0, g1 = w[-1] + 1, min(last, nxt[0] if nxt else last)
        if g1 > g0 and g0 > first:
            gaps.append((g0, g1))
    if not gaps:
        return plain
    main, a = [], first
    for g0, g1 in gaps:
        if g0 > a:
            main.append((a, g0))
        a = g1
    if a < last:
        main.append((a, last))
    # the main pass: the schedule of a shard of n_main frames (ramp at both ends), its intervals mapped back onto the ranges that are left
    n_main = sum(b - a for a, b in main)
    virt = chunk_schedule(0, n_main, chunk, ramp, 0, ramp_from=3)
    if last_main > 0 and virt and virt[-1][1] - virt[-1][0] > last_main + last_main // 2:
        v0, v1 = virt[-1]
        virt[-1:] = [(v0, v1 - last_main), (v1 - last_main, v1)]
    out = []
    for v0, v1 in virt:
        rs, pos = [], 0
        for a, b in main:
            lo, hi = max(v0, pos), min(v1, pos + (b - a))
            if hi > lo:
                rs.append((a + lo - pos, a + hi - pos))
            pos += b - a
        out.append(tuple(rs))
    # the deferred gaps: whole gaps (a split gap would need one more halo frame), in n_groups chunks of about `group` frames each
    tot = sum(b - a for a, b in gaps)
    n_groups = max(1, int(round(tot / float(max(1, group)))))
    per = -(-len(gaps) // n_groups)
    for k in range(0, len(gaps), per):
        ch = []
        for g0, g1 in gaps[k:k + per]:
            while g1 - g0 > chunk:                                 # (a gap longer than a chunk)
                out.append(((g0, g0 + chunk),)); g0 += chunk
            ch.append((g0, g1))
        while sum(b - a for a, b in ch) > chunk:                   # (never more than `chunk` frames per chunk)
            out.append((ch.pop(0),))
        out.append(tuple(ch))
    return out


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


class _Traced:
    """debug aid (YGZ_OFFLINE_TRACE=1): times every ABI call of a context on the host -- a call that blocks shows up here"""

    def __init__(self, obj, log, tag):
        object.__setattr__(self, "_o", obj); object.__setattr__(self, "_log", log); object.__setattr__(self, "_tag", tag)

    def __getattr__(self, name):
        a = getattr(self._o, name)
        if not callable(a):
            return a
        import time

        def call(*args, **kw):
            args = tuple(x._o if isinstance(x, _Traced) else x for x in args)
            t0 = time.perf_counter()
            r = a(*args, **kw)
            self._log.append((self._tag + "." + name, t0, time.perf_counter()))
            return r
        return call


class OfflineVO:
    """One rank of the offline run.  frame_source(i) -> BGR uint8 [h, w, 3] (or gray [h, w]); depth_source(i) -> depth map [h, w]
    (metres).  block_source(frames) -> (frames [n, h, w, 3] or [n, h, w] uint8, depth images [n, dh, dw]) replaces the per-frame
    sources when the caller holds the sequence in page-locked memory in the form the ABI uploads (then every copy is asynchronous).
    depth_div / depth_dtype: the depth image handed to the device is depth_source(i)[::depth_div, ::depth_div] as float64 / float32
    metres or, for uint16, round(depth / depth_scale) (TUM RGB-D: depth_scale = 1 / 5000)."""

    def __init__(self, width, height, n_total, rank=0, world=1, device=0, chunk=128, levels=3, kf_stride=8, window_kfs=8,
                 max_points=2000, ba_iterations=20, overlap=False, process_group=None, exchange_on_device=True, keep=False, lanes=3,
                 depth_div=1, depth_dtype=np.float64, depth_scale=1.0 / 5000.0, pipeline_ba=True, lm_group=None, upload_ahead=0,
                 obs_mode="direct", ba_rounds=1, outlier_chi2=5.991):
        from . import _lib
        self.lib = _lib
        self.w, self.h, self.levels = width, height, levels
        self.n_total, self.rank, self.world = n_total, rank, world
        self.chunk, self.kf_stride, self.window_kfs = chunk, kf_stride, window_kfs
        self.max_points, self.ba_iterations = max_points, ba_iterations
        self.overlap, self.pg, self.exchange_on_device, self.keep = overlap, process_group, exchange_on_device, keep
        self.depth_div, self.depth_dtype, self.depth_scale = depth_div, np.dtype(depth_dtype), depth_scale
        self.pipeline_ba, self.lm_group = pipeline_ba, lm_group
        # observations of a window's map points in its keyframes: "direct" = what LocalMapping::ProjectMapPoints leaves in a keyframe
        # (projection with the tracked pose + FindDirectProjection from the anchor's image, LocalMapping.cpp:47-120) -- the keyframe rows
        # then carry the images; "match" = good cross-checked Hamming matches of the anchor's descriptors (rounds 2-3).  After the LM the
        # edges above outlier_chi2 are counted (BA.cpp:503-515); ba_rounds = 2 switches them off and optimises once more (what the next
        # LocalBAG2O of the reference sees after Feature::_bad was set).
        assert obs_mode in ("direct", "match") and ba_rounds in (1, 2)
        self.obs_mode, self.ba_rounds, self.outlier_chi2 = obs_mode, ba_rounds, outlier_chi2
        import os as _os
        self.fifo_uploads = _os.environ.get("YGZ_OFF_FIFO", "1") != "0"
        self.ramp = _os.environ.get("YGZ_OFF_RAMP", "1") != "0"
        self.kf_tail = _os.environ.get("YGZ_OFF_KF_TAIL", "1") != "0"
        # the keyframe-free frames behind the last keyframe of the last `defer_gaps` windows of the shard are processed at the very end
        # (chunk_plan), so that the last LM launch has company: on by default for long shards (>= 768 frames: 60.6 -> 59.2 ms at 1024 frames
        # with 13 gaps; 16 gaps 59.6), off for short ones (512 frames: + 2 ms; 128: + 0.7 ms -- two more halo frames per gap and the ramp of
        # the main pass cost more than the LM tail they hide).  One chunk PER gap (first form of the experiment) lost 5 ms: latency-bound
        # small chunks
        self.start, self.count, self.halo = ydist.shard_frames(n_total, rank, world)
        self.defer_gaps = int(_os.environ.get("YGZ_OFF_DEFER", "-1"))
        if self.defer_gaps < 0:
            self.defer_gaps = 13 if self.count >= 768 else 0
        self.device = device
        # the chunks of this shard in processing order (chunk_plan); a chunk's frames and the halo frame in front of each of its ranges take
        # consecutive slots of a lane
        self.wins = ba_windows(n_total, kf_stride, window_kfs)
        self.chunks = chunk_plan(self.start, self.start + self.count, chunk, self.ramp, kf_stride if self.kf_tail else 0, self.wins,
                                 self.defer_gaps if pipeline_ba else 0)
        n_slots = max([sum(b - a + (1 if a > 0 else 0) for a, b in ch) for ch in self.chunks] + [2])
        n_lanes = max(1, min(lanes, len(self.chunks)))                            # never more lanes than chunks
        # `depth` chunks compute at a time; with upload_ahead > 0 there are more contexts than that, and the kernels of chunk i wait for
        # chunk i - depth while its upload (first in its stream) does not: the link runs ahead of the kernels by upload_ahead chunks
        self.depth = n_lanes
        if n_lanes == lanes:
            n_lanes += max(0, int(_os.environ.get("YGZ_OFF_AHEAD", upload_ahead)))
        self.lanes = []
        for _ in range(n_lanes):
            c = _lib.HipContext(width=width, height=height, levels=levels, max_frames=max(n_slots, 2), device=device)
            c.set_overlap(overlap)
            self.lanes.append(c)
        self.ctx = self.lanes[0]
        # windows, owners; the third context holds the keyframe store and the BA windows of this rank
        self.owner = [frame_owner(w[0], n_total, world) for w in self.wins]
        self.mine = [i for i, o in enumerate(self.owner) if o == rank]
        last = self.start + self.count
        self.local = [i for i in self.mine if self.wins[i][-1] < last]            # every keyframe tracked by this rank
        self.any_cross = any(frame_owner(w[-1], n_total, world) != o for w, o in zip(self.wins, self.owner))
        self.n_kf = len(keyframes(n_total, kf_stride))
        K1 = max(1, window_kfs - 1)
        self.build_group = max(1, min(len(self.mine), 16))                       # windows per build call (matcher rows: group x (K - 1) pairs)
        if self.lm_group is None:
            # windows per resident-LM launch.  A launch takes 8-9 ms whether it holds one window or eight (latency-bound) and the launches
            # queue on one stream, so they only hide behind the tracking of the following chunks if there are few of them: half of this
            # rank's windows per launch (at most 8) on a long shard, all of them in one launch on a short one
            self.lm_group = min(8, len(self.mine) // 2) if self.count > 256 else min(8, len(self.mine))
        self.lm_group = max(1, self.lm_group)
        self.bg_team_budget = int(_os.environ.get("YGZ_OFF_BG_BUDGET", "0"))
        self.lm_sched = [max(1, int(x)) for x in _os.environ.get("YGZ_OFF_LM_SCHED", "").split(",") if x.strip()]
        self.ba = _lib.HipContext(width=width, height=height, levels=levels, max_frames=max(8, self.build_group * K1), device=device)
        self.rows_t = None
        if world > 1 and self.any_cross:                                          # rows of other ranks arrive by a collective: torch owns the memory
            import torch
            rb = self.ba.kf_row_bytes(self.obs_mode == "direct")
            self.rows_t = torch.zeros((self.n_kf + self.build_group) * rb, dtype=torch.uint8, device=torch.device("cuda", device))
            self.ba.kf_store_create(self.n_kf, n_total, self.build_group, self.rows_t.data_ptr(), self.rows_t.numel(), with_images=self.obs_mode == "direct")
        else:
            self.ba.kf_store_create(self.n_kf, n_total, self.build_group, with_images=self.obs_mode == "direct")
        if self.mine:
            self.ba.ba_reserve_windows(0, len(self.mine), window_kfs, max_points)
        self.trace = None
        import os
        if os.environ.get("YGZ_OFFLINE_TRACE") == "1":
            self.trace = []
            self.lanes = [_Traced(c, self.trace, "lane%d" % i) for i, c in enumerate(self.lanes)]
            self.ctx = self.lanes[0]
            self.ba = _Traced(self.ba, self.trace, "ba")
        self.S = 6 * window_kfs + 3 * max_points + 12                             # one window state row: poses | points | K P E its trials chi2_0 chi2 lambda | edges tested, outliers, chi2, chi2 of inliers
        # page-locked result rows (per-pair summary, keypoint counts), one set per chunk
        self._pin = [dict(sum=_lib.PinnedArray((max(2, sum(b - a + 1 for a, b in ch)), _lib.SUMMARY_FIELDS), np.float64),
                          cnt=_lib.PinnedArray((max(2, sum(b - a + 1 for a, b in ch)),), np.int32)) for ch in self.chunks]
        self.run_ahead = int(os.environ.get("YGZ_OFF_RUN_AHEAD", "0"))          # chunks enqueued beyond one per lane before the oldest is collected (experiment; 99 = all)
        self.timing = {}

    def close(self):
        for c in self.lanes:
            c.close()
        self.ba.close()
        for p in self._pin:
            p["sum"].free(); p["cnt"].free()

    # ------------------------------------------------------------------ depth images
    def depth_image(self, d):
        return depth_image(d, self.depth_div, self.depth_dtype, self.depth_scale)

    def depth_at(self, dimg, px):
        return depth_at(dimg, px, self.w, self.h, self.depth_scale)

    # ------------------------------------------------------------------ phase 1: the hot path over this shard (+ the BA windows it completes)
    def track_shard(self, frame_source, depth_source, block_source=None):
        """The hot path over the frames this rank owns, chunk by chunk on alternating lanes; returns per-frame records.  Per chunk:
        one upload of the frames and depth images, the batched kernels, the keyframes' rows and the pairs' relative poses into the
        store (device to device), one download of the per-pair summary -- no host round trip in between.  Windows whose keyframes
        are all in are built and optimised on the third context while the next chunks run (pipeline_ba)."""
        rec = {}
        first, last = self.start, self.start + self.count
        chunks = self.chunks
        tracked = np.zeros(self.n_total, bool)
        pending = []
        self._ba_done, self._ba_built = set(), []
        self._lm_launches = 0
        self._last_upload = None
        for ci, ranges in enumerate(chunks):
            # the host hands a lane its next chunk when it has read the results of the lane's previous one: `lanes` chunks are in flight.
            # (YGZ_OFF_RUN_AHEAD=99 enqueues every chunk at once -- every chunk has its own page-locked result rows, a lane's stream orders
            # the upload of its next chunk behind the kernels of its previous one.  Measured SLOWER, 63.6 against 59.8 ms per 1024 frames and
            # 46.7 against 42.1 with gray frames: with everything queued the resident-LM teams and the tracking kernels of three lanes compete
            # for the CUs -- the LM launch beside the tracking takes 8.6 instead of 5.6 ms and uploads wait behind kernels.)
            li = ci % len(self.lanes)
            while len(pending) >= len(self.lanes) + self.run_ahead:
                self._collect(*pending.pop(0), rec)
            info = self._enqueue(li, ranges, frame_source, depth_source, block_source, ci)
            if self.keep:                                      # parity runs read everything back before the lane moves on
                self._collect(li, info, rec)
            else:
                pending.append((li, info))
            if self.pipeline_ba:
                # windows whose keyframes are all in: built at once (a matcher launch and two small kernels); the resident LM is a latency-
                # bound kernel that takes as long for two windows as for eight, so it is launched per lm_group windows (and for the rest
                # after the last chunk): its launches then fit beside the tracking of the following chunks instead of queueing up
                for c0, c1 in ranges:
                    tracked[c0:c1] = True
                new = [i for i in self.local if i not in self._ba_done and i not in self._ba_built and tracked[self.wins[i][0]:self.wins[i][-1] + 1].all()]
                self._ba_launch(new, optimize=False)
                self._ba_built += new
                all_built = len(self._ba_done) + len(self._ba_built) == len(self.local)     # nothing more will come: the last launch need not wait for the last chunk
                if self._ba_built and (len(self._ba_built) >= self._lm_next() or ci == len(chunks) - 1 or all_built):
                    # beside the tracking of the next chunks a launch keeps to a few CUs (its members each own one, and wait for it while a
                    # tracking workgroup drains); the last launch, which nothing runs beside, takes the default half of the device
                    self.ba.ba_set_team_budget(self.bg_team_budget if ci < len(chunks) - 1 and not all_built else 0)
                    self._ba_optimize(self._ba_built)
                    self._ba_built = []
                    self._lm_launches += 1
        for li, info in pending:
            self._collect(li, info, rec)
        return rec

    def _lm_next(self):
        """windows the next resident-LM launch waits for: lm_group, or the k-th entry of the schedule (YGZ_OFF_LM_SCHED, experiment)"""
        if self.lm_sched:
            return self.lm_sched[min(self._lm_launches, len(self.lm_sched) - 1)]
        return self.lm_group

    def _enqueue(self, li, ranges, frame_source, depth_source, block_source, ci=0):
        """one chunk = the frame ranges `ranges`, each with a one-frame halo (its predecessor) in front; the ranges take consecutive slots"""
        c = self.lanes[li]
        asyn = block_source is not None
        frames, spans = [], []                                    # spans: (first slot, frames of the range incl. halo)
        for c0, c1 in ranges:
            fr = list(range(c0 - 1, c1)) if c0 > 0 else list(range(c0, c1))
            spans.append((len(frames), fr)); frames += fr
        assert len(set(frames)) == len(frames)                    # (ranges of a chunk are not adjacent: chunk_plan merges those)
        slot_of = {f: k for k, f in enumerate(frames)}
        n = len(frames)
        if self._last_upload is not None and self.fifo_uploads:
            c.wait_mark(self._last_upload)                     # uploads cross PCIe one after the other, each at the full rate
        keep_dimg = []
        for s0, fr in spans:
            if asyn:
                img, dimg = block_source(fr)
            else:
                img = np.ascontiguousarray(np.stack([frame_source(f) for f in fr]))
                dimg = np.ascontiguousarray(np.stack([self.depth_image(depth_source(f)) for f in fr]))
            if img.ndim == 3:                                  # [n, h, w]: the caller converted to gray (a third of the PCIe bytes)
                c.upload_gray_batch(s0, img, wait=not asyn)
            else:
                c.upload_bgr_batch(s0, img, wait=not asyn)
            c.upload_depth_batch(s0, dimg, self.depth_scale, wait=not asyn)
            from_bgr = img.ndim != 3
            if self.keep:
                keep_dimg.append(dimg)
        c.mark(); self._last_upload = c
        if len(self.lanes) > self.depth and ci >= self.depth:
            c.stream_wait(self.lanes[(ci - self.depth) % len(self.lanes)])      # at most `depth` chunks' kernels share the GPU
        c.build_pyramid(0, n, from_bgr=from_bgr)
        c.detect(0, n)
        c.keypoint_depths_from_image(0, n)                     # Feature::_depth / _mappoint of the fresh keypoints
        pairs = [(f, f - 1) for c0, c1 in ranges for f in range(c0, c1) if f - 1 in slot_of]
        if pairs:
            q = [slot_of[a] for a, _ in pairs]
            t = [slot_of[b] for _, b in pairs]
            ident = np.tile(I7, (len(pairs), 1))
            c.match_slots(q, t, 1)
            c.match_postfilter()
            c.track_begin(q, t, ident, ident, predict=False)
            c.track_sparse_align()
            c.track_klt()
            c.track_adopt_pose()
            c.track_direct()
            c.track_pose_only()
            c.track_get_summary(out=self._pin[ci]["sum"].array, wait=False)
            p0 = 0
            while p0 < len(pairs):                             # runs of consecutive frames (one per range)
                p1 = p0 + 1
                while p1 < len(pairs) and pairs[p1][0] == pairs[p1 - 1][0] + 1:
                    p1 += 1
                self.ba.kf_store_put_trel(c, p0, p1 - p0, pairs[p0][0])
                p0 = p1
        c.get_keypoint_counts(0, n, out=self._pin[ci]["cnt"].array, wait=False)
        kf = [f for c0, c1 in ranges for f in range(c0, c1) if f % self.kf_stride == 0]
        if kf:
            self.ba.kf_store_put(c, [slot_of[f] for f in kf], [f // self.kf_stride for f in kf])
        return dict(ranges=ranges, frames=frames, slot_of=slot_of, pairs=pairs, pin=self._pin[ci], dimg=np.concatenate(keep_dimg) if self.keep else None)

    def _collect(self, li, info, rec):
        """wait for the lane, then read its chunk's results out of the page-locked buffers"""
        c = self.lanes[li]
        c.synchronize()
        S = info["pin"]["sum"].array[:len(info["pairs"])].copy()
        cnt = info["pin"]["cnt"].array[:len(info["frames"])].copy()
        slot_of = info["slot_of"]
        for f in (f for c0, c1 in info["ranges"] for f in range(c0, c1)):
            k = slot_of[f]
            r = dict(n_kp=int(cnt[k]))
            if self.keep:
                kp = c.get_keypoints(k)
                kp["depth"], has_mp = c.get_keypoint_depths(k)
                assert np.array_equal(kp["depth"], self.depth_at(info["dimg"][k], kp["px"])) and np.array_equal(has_mp, kp["depth"] > 0)   # the device's look-up
                r["kp"] = kp
            rec[f] = r
        for p, (cur, ref) in enumerate(info["pairs"]):
            r = rec[cur]
            r.update(T_sa=S[p, 0:7].copy(), sa_n_meas=int(S[p, 7]), T_rel=S[p, 24:31].copy(), po_inliers=int(S[p, 14]),
                     po_rounds=int(S[p, 15]), n_match=int(S[p, 16]), n_good=int(S[p, 17]), min_dis=float(S[p, 18]),
                     n_klt=int(S[p, 19]), n_fdp=int(S[p, 20]))
            if self.keep:                      # everything a parity test wants to look at
                n_meas, T_sa, iters = c.track_get_pose(p)
                po = c.track_get_pose_only(p)
                good, n_good, min_dis = c.get_good_matches(p)
                idx, dist_ = c.get_matches(p)
                pts, st, err = c.track_get_klt(p)
                ok, pxd, lvl = c.track_get_direct(p)
                assert np.array_equal(T_sa, r["T_sa"]) and np.array_equal(po["T"], r["T_rel"]) and n_good == r["n_good"]
                assert int(st.astype(bool).sum()) == r["n_klt"] and int(ok.sum()) == r["n_fdp"] and int((idx >= 0).sum()) == r["n_match"]
                r.update(sa_iters=iters, m_idx=idx, m_dist=dist_, m_good=good, klt_pts=pts, klt_status=st, klt_err=err,
                         fdp_ok=ok, fdp_px=pxd, fdp_level=lvl, po_bad=po["bad"], po_pose=po["pose"])

    # ------------------------------------------------------------------ phase 2: trajectory all-gather
    def gather(self, rec):
        """all-gather of the per-shard relative poses -> the chained global trajectory, identical on every rank"""
        local = np.stack([rec[f].get("T_rel", I7) for f in range(self.start, self.start + self.count)]) if self.count else np.zeros((0, 7))
        if self.world > 1:
            dev = self._torch_device() if self.exchange_on_device else None
            T_rel = ydist.gather_trajectories(local, self.n_total, self.rank, self.world, device=dev)
        else:
            T_rel = local
        T_rel[0] = I7
        return T_rel, chain(T_rel)

    def gather_keyframes(self):
        """keyframe rows (pixels, levels, descriptors, depths: fixed-size rows of the device store) of every keyframe on every
        rank: ONE all-gather on the store's memory -- needed only when a window straddles a shard boundary"""
        if self.world == 1 or not self.any_cross:
            return
        import torch
        rb = self.ba.kf_row_bytes(self.obs_mode == "direct")
        rows = self.rows_t[:self.n_kf * rb].view(self.n_kf, rb)
        own = [[k for k, f in enumerate(keyframes(self.n_total, self.kf_stride)) if frame_owner(f, self.n_total, self.world) == r]
               for r in range(self.world)]
        mine = own[self.rank]
        local = rows[mine[0]:mine[-1] + 1] if mine else rows[:0]
        torch.cuda.synchronize(self._torch_device())
        g = ydist.all_gather_rows(local, [len(o) for o in own], self.pg, via_host=not self.exchange_on_device)
        for r in range(self.world):
            if r != self.rank and own[r]:
                rows[own[r][0]:own[r][-1] + 1] = g[r, :len(own[r])]
        torch.cuda.synchronize(self._torch_device())           # torch's stream wrote the rows, the ABI context has its own
        self.ba.kf_store_refresh()

    def _torch_device(self):
        import torch
        return torch.device("cuda", self.device)

    # ------------------------------------------------------------------ phase 3: BA round
    def _ba_optimize(self, wis):
        """the resident LM on windows that are built (consecutive owned windows), asynchronous"""
        if wis:
            assert wis == list(range(wis[0], wis[-1] + 1))
            self._lm(self.mine.index(wis[0]), len(wis))
            self._ba_done.update(wis)

    def _lm(self, slot0, n):
        """optimize(20) + the inlier test of BA.cpp:503-515 (+ a second optimisation without the outliers when ba_rounds == 2), asynchronous"""
        self.ba.ba_optimize_resident(slot0, n, self.ba_iterations, want_stats=False)
        self.ba.ba_mark_outliers(slot0, n, self.outlier_chi2, disable=self.ba_rounds == 2)
        if self.ba_rounds == 2:
            self.ba.ba_optimize_resident(slot0, n, self.ba_iterations, want_stats=False)
            self.ba.ba_mark_outliers(slot0, n, self.outlier_chi2, disable=False)

    def _ba_launch(self, wis, optimize=True):
        """build (+ optimise) the given (owned) windows on the BA context, behind everything the lanes have enqueued so far"""
        if not wis:
            return
        for c in self.lanes:
            self.ba.stream_wait(c)
        K = self.window_kfs
        for g0 in range(0, len(wis), self.build_group):
            grp = wis[g0:g0 + self.build_group]
            assert grp == list(range(grp[0], grp[-1] + 1))
            kfi = np.zeros((len(grp), K), np.int32); kff = np.zeros((len(grp), K), np.int32)
            for a, wi in enumerate(grp):
                w = self.wins[wi]
                kff[a, :len(w)] = w
                kfi[a, :len(w)] = [f // self.kf_stride for f in w]
            slot0 = self.mine.index(grp[0])
            self.ba.ba_build_windows(slot0, kfi, kff, [len(self.wins[wi]) for wi in grp], obs_mode=1 if self.obs_mode == "direct" else 0)
            if optimize:
                self._lm(slot0, len(grp))
        if optimize:
            self._ba_done.update(wis)

    def ba_round(self, T_rel):
        """the windows this rank owns that are not optimised yet (those straddling a shard boundary; all of them without
        pipeline_ba), then the exchange: every owner's refined window states to every rank in one all-gather"""
        import time
        t0 = time.perf_counter()
        rest = [i for i in self.mine if i not in self._ba_done]
        if rest:
            if self.world > 1:                                 # relative poses of the frames other ranks tracked
                self.ba.kf_store_set_trel(0, T_rel)
            self.ba.ba_set_team_budget(0)
            self._ba_launch(rest)
        self.ba.synchronize()
        self._retry_timed_out()
        t1 = time.perf_counter()
        n_w, S = len(self.wins), self.S
        if self.world == 1:
            host = self.ba.ba_pack_states(0, n_w, S) if n_w else np.zeros((0, S))
        else:
            import torch
            dev = self._torch_device()
            state = torch.zeros((n_w, S), dtype=torch.float64, device=dev)
            # the zero fill runs on torch's current stream, k_ba_pack on the BA context's own (non-blocking) stream: without this wait the
            # fill may land after the packed rows and zero them (gather_keyframes guards its buffer the same way)
            torch.cuda.synchronize(dev)
            if self.mine:
                self.ba.ba_pack_states(0, len(self.mine), S, dst_ptr=state[self.mine[0]].data_ptr(), wait=True)
            exchange_rows(state, self.owner, self.world, self.pg, via_host=not self.exchange_on_device)
            host = state.cpu().numpy()
        t2 = time.perf_counter()
        self.ba_timing = {"tail_after_tracking": (t1 - t0) * 1e3, "exchange_download": (t2 - t1) * 1e3}
        K, P = self.window_kfs, self.max_points
        out, dims = [], {}
        for wi, w in enumerate(self.wins):
            tail = host[wi, 6 * K + 3 * P:]
            if tail[3] < 0:
                raise RuntimeError("BA window %d: the resident LM did not finish (team barrier time-out)" % wi)
            dims[wi] = (int(tail[0]), int(tail[1]), int(tail[2]))
            if dims[wi][1] == 0 or dims[wi][2] == 0:            # no map point of the anchor was observed in another keyframe: the window was not optimised
                self.degenerate_windows = getattr(self, "degenerate_windows", []) + [wi]
            out.append(dict(kfs=w, owner=self.owner[wi], poses=host[wi, :len(w) * 6].reshape(len(w), 6).copy(), state=host[wi].copy(),
                            stats=np.array([tail[5], tail[6], tail[3], tail[2]]),
                            inliers=dict(edges=int(tail[8]), outliers=int(tail[9]), chi2=float(tail[10]), chi2_inliers=float(tail[11])),
                            lm=dict(iterations=int(tail[3]), trials=int(tail[4]), degenerate=bool(dims[wi][1] == 0 or dims[wi][2] == 0))))
        return out, dims

    def _retry_timed_out(self):
        """A resident-LM team whose members were not co-resident within the spin bound of a team barrier (possible beside the tracking
        kernels of several lanes) leaves without a result.  Its window is rebuilt (the loop updates the points in place) and solved once more
        by ONE workgroup, which needs no co-residency; a second failure raises."""
        if not self.mine:
            return
        bad = [k for k, it in enumerate(self.ba.ba_lm_iterations(0, len(self.mine))) if it < 0 and self.mine[k] in self._ba_done]
        if not bad:
            return
        self.lm_retries = getattr(self, "lm_retries", 0) + len(bad)
        self.ba.ba_set_team_budget(1)                          # G = 1: one workgroup per window
        done = set(self._ba_done)
        for k in bad:
            self._ba_launch([self.mine[k]])
        self._ba_done = done
        self.ba.ba_set_team_budget(0)
        self.ba.synchronize()
        still = [self.mine[k] for k in bad if self.ba.ba_lm_iterations(k, 1)[0] < 0]
        if still:
            raise RuntimeError("BA windows %s: the resident LM did not finish even with one workgroup per window" % still)

    # ------------------------------------------------------------------ whole run
    def run(self, frame_source, depth_source, block_source=None):
        import time
        t0 = time.perf_counter()
        rec = self.track_shard(frame_source, depth_source, block_source)
        t1 = time.perf_counter()
        T_rel, traj = self.gather(rec)
        self.gather_keyframes()
        t2 = time.perf_counter()
        windows, dims = self.ba_round(T_rel)
        t3 = time.perf_counter()
        self.timing = {"track_shard": (t1 - t0) * 1e3, "gather": (t2 - t1) * 1e3, "ba_round": (t3 - t2) * 1e3}
        self.timing.update({"ba_" + k: v for k, v in getattr(self, "ba_timing", {}).items()})
        res = dict(records=rec, T_rel=T_rel, trajectory=traj, windows=windows, built=dims)
        res["keyframe_pose"] = _LazyKeyframePoses(windows, traj)
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


def build_window_host(kf_tab, kfs, T_rel, fx, fy, cx, cy, max_points, match_sets=None, direct=None, width=0, height=0):
    """Host restatement of what ygz_hip_ba_build_windows assembles for one window (the tests compare the device-built graph with it):
    kf_tab[f] = dict(px, level, desc, depth) of keyframe f, match_sets(descs, pair_q, pair_t) = HipContext.match_sets.  Returns the
    graph of ba::LocalBAG2O in the anchor's gauge: poses (g2o order), points, edges sorted by (point, keyframe).
    With direct = fn(ref_frame, cur_frame, T_cur, px_ref, depth_ref, level_ref, px_cur) -> (ok, px) (a per-pair FindDirectProjection, e.g.
    HipContext.find_direct_projection on a context that holds the keyframes' pyramids) the observations are those of obs_mode 1: the map
    point projected with the chained pose, FindCandidates' test (z >= 0, InFrame(px, 20) of a width x height frame), FindDirectProjection."""
    A = kf_tab[kfs[0]]
    sel = np.nonzero(A["depth"] > 0)[0][:max_points]
    z = A["depth"][sel]
    pc = np.stack([(A["px"][sel, 0] - cx) * z / fx, (A["px"][sel, 1] - cy) * z / fy, z], axis=1)     # Pixel2Camera (Camera.h:56-62)
    ep, el, obs = [np.zeros(len(sel), np.int32)], [np.arange(len(sel), dtype=np.int32)], [A["px"][sel]]
    others = [(j, f) for j, f in enumerate(kfs[1:], start=1)]
    if others and len(sel) and direct is not None:
        from . import _lib
        Tc = _lib.se3_chain(T_rel[kfs[0]:kfs[-1] + 1])            # T(anchor) = identity, the same Sophus products as the device's chain
        for j, f in others:
            T = Tc[f - kfs[0]]
            q = se3_act(T, pc)
            pred = np.stack([fx * q[:, 0] / q[:, 2] + cx, fy * q[:, 1] / q[:, 2] + cy], axis=1)            # Camera2Pixel (Camera.h:46-51)
            vis = ~(q[:, 2] < 0) & (pred[:, 0] >= 20) & (pred[:, 0] < width - 20) & (pred[:, 1] >= 20) & (pred[:, 1] < height - 20)
            g = np.nonzero(vis)[0]
            if len(g):
                ok, pxo = direct(kfs[0], f, T, A["px"][sel][g], z[g], A["level"][sel][g], pred[g])
                g = g[ok]; pxo = pxo[ok]
            else:
                pxo = np.zeros((0, 2))
            ep.append(np.full(len(g), j, np.int32)); el.append(g.astype(np.int32)); obs.append(pxo)
    elif others and len(sel):
        res = match_sets([A["desc"][sel]] + [kf_tab[f]["desc"] for _, f in others], [0] * len(others), list(range(1, len(others) + 1)))
        for (j, f), r in zip(others, res):
            g = np.nonzero(r["good"])[0]
            ep.append(np.full(len(g), j, np.int32)); el.append(g.astype(np.int32)); obs.append(kf_tab[f]["px"][r["idx"][g]])
    ep, el, obs = np.concatenate(ep), np.concatenate(el), np.concatenate(obs)
    n_obs = np.bincount(el, minlength=len(sel))
    keep_pt = n_obs >= 2                                   # a point seen only by the fixed anchor constrains nothing
    remap = -np.ones(len(sel), np.int64); remap[keep_pt] = np.arange(int(keep_pt.sum()))
    ke = keep_pt[el]
    ep, el, obs = ep[ke], remap[el[ke]].astype(np.int32), obs[ke]
    order = np.lexsort((ep, el))
    T = I7.copy()
    poses = [se3_log_g2o(T)]
    for f in range(kfs[0] + 1, kfs[-1] + 1):
        T = se3_mul(T_rel[f], T)
        if f in kfs:
            poses.append(se3_log_g2o(T))
    fixed = np.zeros(len(kfs), np.uint8); fixed[0] = 1
    return dict(kfs=list(kfs), poses=np.stack(poses), fixed=fixed, points=pc[keep_pt], edge_pose=ep[order], edge_point=el[order], obs=obs[order],
                anchor_feature=sel[keep_pt])
