"""Batched offline visual odometry (BASELINE.json configs[4]): a sequence of N frames sharded over `world` GPUs in
contiguous chunks with a one-frame halo, the per-frame hot path run for every consecutive pair, a local-BA round per
window of keyframes, and the two exchanges SURVEY 8e names: the BA-window state (map points + keyframe poses) broadcast
from its owner straight into HBM, and the all-gather of per-shard trajectories.

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
initialiser / triangulation (out of scope); here the sequence supplies a depth map per frame.

BA round (LocalMapping::LocalBA -> ba::LocalBAG2O, LocalMapping.cpp:149-208, BA.cpp:386-543): keyframes are every
`kf_stride`-th frame, a window = `window_kfs` consecutive keyframes owned by the rank that owns its first keyframe (the
anchor, held fixed like keyframe 0 at BA.cpp:404); map points = the anchor's features with depth, observations = the
good cross-checked Hamming matches of the anchor's descriptors in the other keyframes of the window.

This module is host logic over the C ABI (ygz_slam_amd._lib); it never touches oracle/.
"""
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
    """T[0] = identity (the first frame defines the world); T[i] = T_rel[i] * T[i-1] -- se3_mul's formulas on Python floats (a
    thousand 7-vector products through numpy temporaries cost 25 ms, this loop 3)"""
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


def exchange_rows(buf, owner, world, pg=None):
    """The map exchange: row i of `buf` (a torch tensor, in HBM on the GPU box) is owned by rank owner[i]; every owner broadcasts
    its rows -- they are contiguous, windows being ordered by anchor frame -- so that all ranks end with the same replica.
    One collective per owner and round (RCCL over xGMI with backend nccl; gloo in the CPU tests)."""
    if world == 1:
        return
    import torch.distributed as dist
    for r in range(world):
        rows = [i for i, o in enumerate(owner) if o == r]
        if rows:
            assert rows == list(range(rows[0], rows[-1] + 1))
            dist.broadcast(buf[rows[0]:rows[-1] + 1], src=r, group=pg)


class OfflineVO:
    """One rank of the offline run.  frame_source(i) -> BGR uint8 [h, w, 3]; depth_source(i) -> float [h, w]."""

    def __init__(self, width, height, n_total, rank=0, world=1, device=0, chunk=128, levels=3, kf_stride=8, window_kfs=8,
                 max_points=2000, ba_iterations=20, overlap=True, process_group=None, exchange_on_device=True, keep=False, lanes=2):
        from . import _lib
        self.lib = _lib
        self.w, self.h, self.levels = width, height, levels
        self.n_total, self.rank, self.world = n_total, rank, world
        self.chunk, self.kf_stride, self.window_kfs = chunk, kf_stride, window_kfs
        self.max_points, self.ba_iterations = max_points, ba_iterations
        self.overlap, self.pg, self.exchange_on_device, self.keep = overlap, process_group, exchange_on_device, keep
        self.start, self.count, self.halo = ydist.shard_frames(n_total, rank, world)
        self.device = device
        n_slots = min(self.count, chunk) + 1
        self.ctx = _lib.HipContext(width=width, height=height, levels=levels, max_frames=max(n_slots, 2, window_kfs), device=device)
        self.ctx.set_overlap(overlap)
        # a second context (own streams, own slots) takes every other chunk from its own host thread: the H2D copy of one chunk and
        # the host-side depth look-up of its keypoints run under the kernels of the other (the C calls release the GIL)
        self.lanes = [self.ctx]
        if lanes > 1 and self.count > chunk:
            c2 = _lib.HipContext(width=width, height=height, levels=levels, max_frames=max(n_slots, 2, window_kfs), device=device)
            c2.set_overlap(overlap)
            self.lanes.append(c2)
        self.timing = {}

    def close(self):
        for c in self.lanes:
            c.close()

    # ------------------------------------------------------------------ phase 1: the hot path over this shard
    def track_shard(self, frame_source, depth_source, block_source=None):
        """The hot path over the frames this rank owns, chunk by chunk; returns per-frame records (+ the keyframe tables of the
        owned keyframes).  block_source(frames) -> (bgr [n, h, w, 3] uint8 C-contiguous, depth maps [n, h, w]) replaces the
        per-frame sources when the caller holds the sequence in (page-locked) memory.  Per chunk: one upload, the batched
        kernels, one download of the keypoint pixels (the depth look-up is the stand-in for the map, see the module text), one
        upload of the depths, and one download of the per-pair summary."""
        rec = {}                      # frame -> dict (the lanes write disjoint keys)
        first, last = self.start, self.start + self.count
        chunks = [(c0, min(c0 + self.chunk, last)) for c0 in range(first, last, self.chunk)]
        if len(self.lanes) == 1 or len(chunks) < 2:
            for c0, c1 in chunks:
                self._track_chunk(self.ctx, c0, c1, rec, frame_source, depth_source, block_source)
            return rec
        import threading
        errors = []

        def lane(k):
            try:
                for c0, c1 in chunks[k::len(self.lanes)]:
                    self._track_chunk(self.lanes[k], c0, c1, rec, frame_source, depth_source, block_source)
            except BaseException as e:                       # surfaces in the caller's thread
                errors.append(e)
        th = [threading.Thread(target=lane, args=(k,)) for k in range(len(self.lanes))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errors:
            raise errors[0]
        return rec

    def _track_chunk(self, c, c0, c1, rec, frame_source, depth_source, block_source):
        cells = c.cells
        frames = list(range(c0 - 1, c1)) if c0 > 0 else list(range(c0, c1))      # one-frame halo: the predecessor of the chunk
        slot_of = {f: k for k, f in enumerate(frames)}
        n = len(frames)
        if block_source is not None:
            bgr, dmaps = block_source(frames)
        else:
            bgr = np.ascontiguousarray(np.stack([frame_source(f) for f in frames]))
            dmaps = [depth_source(f) for f in frames]
        if bgr.ndim == 3:                                      # [n, h, w]: the caller converted to gray (a third of the PCIe bytes)
            c.upload_gray_batch(0, bgr)
            c.build_pyramid(0, n, from_bgr=False)
        else:
            c.upload_bgr_batch(0, bgr)
            c.build_pyramid(0, n, from_bgr=True)
        c.detect(0, n)
        px, cnt = c.get_keypoint_pixels_batch(0, n)
        depth = np.zeros((n, cells), np.float64)
        for k in range(n):
            m = int(cnt[k])
            if m:
                depth[k, :m] = dmaps[k][px[k, :m, 1].astype(np.int64), px[k, :m, 0].astype(np.int64)]
        c.set_keypoint_depths_batch(0, depth, (depth > 0).astype(np.uint8))
        pairs = [(f, f - 1) for f in frames if f - 1 in slot_of and f >= c0]
        S = None
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
            S = c.track_get_summary().copy()
        for f in range(c0, c1):
            k = slot_of[f]
            r = dict(n_kp=int(cnt[k]))
            if self.keep or f % self.kf_stride == 0:
                kp = c.get_keypoints(k)
                kp["depth"] = depth[k, :int(cnt[k])].copy()
                if self.keep:
                    r["kp"] = kp
                if f % self.kf_stride == 0:
                    r["kf"] = {key: kp[key] for key in ("px", "level", "desc", "depth")}
            rec[f] = r
        for p, (cur, ref) in enumerate(pairs):
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

    def gather_keyframes(self, rec):
        """keyframe tables (pixels, levels, descriptors, depths) of every keyframe on every rank"""
        mine = {f: rec[f]["kf"] for f in rec if "kf" in rec[f]}
        if self.world == 1:
            return mine
        import torch.distributed as dist
        allk = [None] * self.world
        dist.all_gather_object(allk, mine, group=self.pg)
        out = {}
        for d in allk:
            out.update(d)
        return out

    def _torch_device(self):
        import torch
        return torch.device("cuda", self.device)

    # ------------------------------------------------------------------ phase 3: BA round
    def build_window(self, kfs, kf_tab, traj, c=None):
        """graph of ba::LocalBAG2O for one window: poses (g2o order), points, edges; anchor = kfs[0] held fixed"""
        c = c or self.ctx
        A = kf_tab[kfs[0]]
        sel = np.nonzero(A["depth"] > 0)[0][:self.max_points]
        T_a = traj[kfs[0]]
        fx, fy, cx, cy = (float(c.params.fx), float(c.params.fy), float(c.params.cx), float(c.params.cy))
        z = A["depth"][sel]
        pc = np.stack([(A["px"][sel, 0] - cx) * z / fx, (A["px"][sel, 1] - cy) * z / fy, z], axis=1)     # Pixel2Camera (Camera.h:56-62)
        pw = se3_act(se3_inv(T_a), pc) if len(sel) else np.zeros((0, 3))
        ep, el, obs = [np.zeros(len(sel), np.int32)], [np.arange(len(sel), dtype=np.int32)], [A["px"][sel]]
        others = [(j, f) for j, f in enumerate(kfs[1:], start=1) if len(sel) and len(kf_tab[f]["level"])]
        if others:                                             # the anchor's descriptors against every other keyframe of the window: one call
            res = c.match_sets([A["desc"][sel]] + [kf_tab[f]["desc"] for _, f in others], [0] * len(others), list(range(1, len(others) + 1)))
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
        poses = np.stack([se3_log_g2o(traj[f]) for f in kfs])
        fixed = np.zeros(len(kfs), np.uint8); fixed[0] = 1
        return dict(kfs=list(kfs), poses=poses, fixed=fixed, points=pw[keep_pt], edge_pose=ep[order], edge_point=el[order], obs=obs[order],
                    anchor_feature=sel[keep_pt])

    def ba_round(self, kf_tab, traj):
        """every rank builds and optimises the windows it owns; window states travel through a device buffer that RCCL fills
        (owner -> everybody), and ygz_hip_ba_set_state_device installs them into the resident windows"""
        import time
        import torch
        c = self.ctx
        tb = time.perf_counter()
        wins = ba_windows(self.n_total, self.kf_stride, self.window_kfs)
        owner = [frame_owner(w[0], self.n_total, self.world) for w in wins]
        mine = [i for i, o in enumerate(owner) if o == self.rank]
        K, P = self.window_kfs, self.max_points
        S = K * 6 + P * 3                                       # one window state: poses | points
        dev = self._torch_device()
        state = torch.zeros((len(wins), S), dtype=torch.float64, device=dev)
        built = {}
        if len(self.lanes) > 1 and len(mine) > 1:                # the windows' matcher calls on both contexts, from two host threads
            import threading

            def lane(k):
                for wi in mine[k::len(self.lanes)]:
                    built[wi] = self.build_window(wins[wi], kf_tab, traj, self.lanes[k])
            th = [threading.Thread(target=lane, args=(k,)) for k in range(len(self.lanes))]
            for t in th:
                t.start()
            for t in th:
                t.join()
            assert len(built) == len(mine)
        t_match = time.perf_counter()
        for li, wi in enumerate(mine):
            b = built[wi] if wi in built else built.setdefault(wi, self.build_window(wins[wi], kf_tab, traj))
            c.ba_upload(li, b["poses"], b["fixed"], b["points"], b["edge_pose"], b["edge_point"], b["obs"])
            row = np.zeros(S)
            row[:b["poses"].size] = b["poses"].ravel()
            row[K * 6:K * 6 + b["points"].size] = b["points"].ravel()
            state[wi] = torch.from_numpy(row).to(dev)
        t_built = time.perf_counter()
        self._exchange(state, owner)                            # the map replica now holds every window's initial state
        torch.cuda.synchronize(dev)                             # the exchange ran on torch's stream, the ABI context has its own
        for li, wi in enumerate(mine):
            base = state[wi].data_ptr()
            c.ba_set_state_device(li, base, base + 8 * K * 6)
        t_x = time.perf_counter()
        stats = c.ba_optimize_resident(0, len(mine), self.ba_iterations) if mine else []
        t_s = time.perf_counter()
        self.ba_timing = {"build_windows": (t_match - tb) * 1e3, "build_upload": (t_built - tb) * 1e3, "exchange_install": (t_x - t_built) * 1e3, "lm_resident": (t_s - t_x) * 1e3}
        for li, wi in enumerate(mine):
            b = built[wi]
            poses, points = c.ba_get_state(li, len(b["poses"]), len(b["points"]))
            row = np.zeros(S)
            row[:poses.size] = poses.ravel()
            row[K * 6:K * 6 + points.size] = points.ravel()
            state[wi] = torch.from_numpy(row).to(dev)
        self._exchange(state, owner)                            # ... and every window's refined state
        chi2 = torch.zeros((len(wins), 4), dtype=torch.float64, device=dev)
        for li, wi in enumerate(mine):
            s = stats[li]
            chi2[wi] = torch.tensor([s.chi2_initial, s.chi2_final, float(s.iterations), float(len(built[wi]["obs"]))],
                                    dtype=torch.float64, device=dev)
        self._exchange(chi2, owner)
        host = state.cpu().numpy()
        out = []
        for wi, w in enumerate(wins):
            out.append(dict(kfs=w, owner=owner[wi], poses=host[wi, :len(w) * 6].reshape(len(w), 6).copy(),
                            state=host[wi].copy(), stats=chi2[wi].cpu().numpy()))
        return out, built

    def _exchange(self, buf, owner):
        exchange_rows(buf, owner, self.world, self.pg)

    # ------------------------------------------------------------------ whole run
    def run(self, frame_source, depth_source, block_source=None):
        import time
        t0 = time.perf_counter()
        rec = self.track_shard(frame_source, depth_source, block_source)
        self.ctx.synchronize()
        t1 = time.perf_counter()
        T_rel, traj = self.gather(rec)
        kf_tab = self.gather_keyframes(rec)
        t2 = time.perf_counter()
        windows, built = self.ba_round(kf_tab, traj)
        t3 = time.perf_counter()
        self.timing = {"track_shard": (t1 - t0) * 1e3, "gather": (t2 - t1) * 1e3, "ba_round": (t3 - t2) * 1e3}
        self.timing.update({"ba_" + k: v for k, v in getattr(self, "ba_timing", {}).items()})
        kf_pose = {}
        for w in windows:
            for k, f in enumerate(w["kfs"]):
                kf_pose[f] = se3_exp_g2o(w["poses"][k])
        return dict(records=rec, T_rel=T_rel, trajectory=traj, windows=windows, keyframe_pose=kf_pose, built=built)
