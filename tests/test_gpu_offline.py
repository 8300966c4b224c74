"""GPU tests (pytest -m gpu) of the offline run of BASELINE configs[4] -- the C++ driver ygz_slam_amd/host/ygz_offline.cpp in
libygz_host.so, bound by ygz_slam_amd/offline.py -- and of the stages it adds to the C ABI: the M3 / M6 match filters, the TrackRefFrame ->
TrackLocalMap hand-over on the device, pose-only BA on resident tracks.  The sharded run (2 ranks, both on the one GPU of the test box, the
driver's exchange hook over gloo because RCCL refuses two ranks on one device) must reproduce the unsharded run exactly; the RCCL path of the
driver is run with a communicator of one rank."""
import os
import pickle
import socket
import numpy as np
import pytest
from conftest import make_ctx, ROOT
from ygz_slam_amd import synth, offline
from ygz_slam_amd import dist as ydist
import window_ref

pytestmark = pytest.mark.gpu
I7 = np.array([0, 0, 0, 1.0, 0, 0, 0])


def test_match_postfilter_and_check_frame_descriptors(hip_lib, oracle):
    """M3 (test_orb_match.cpp:97-104) and M6 (Matcher.cpp:45-84) through the ABI vs the oracle: resident pairs, host arrays,
    clamp at both ends, no match at all, a single pair"""
    seq = synth.Sequence(3, 640, 480, seed=5, step=0.3)
    ctx = make_ctx(hip_lib, max_frames=3)
    for s in range(3):
        ctx.upload_bgr(s, seq.frame(s))
    ctx.build_pyramid(0, 3, from_bgr=True); ctx.detect(0, 3)
    kps = [ctx.get_keypoints(s) for s in range(3)]
    ctx.match_slots([1, 2, 0], [0, 1, 2], 1)
    ctx.match_postfilter()
    for p, (q, t) in enumerate([(1, 0), (2, 1), (0, 2)]):
        idx, dist = ctx.get_matches(p)
        good, ng, md = ctx.get_good_matches(p)
        okeep, on = oracle.good_match_filter(idx, dist)
        assert ng == on and np.array_equal(good, okeep) and 20 <= md <= 50
        g2, ng2, md2 = ctx.match_postfilter_host(idx, dist)
        assert ng2 == on and np.array_equal(g2, okeep) and md2 == md
    rng = np.random.default_rng(3)
    for lo_d, hi_d in ((0, 15), (60, 200), (25, 40)):                       # clamp to the floor / to the ceiling / inside
        n = 700
        idx = rng.integers(-1, 900, n).astype(np.int32); dist = rng.integers(lo_d, hi_d, n).astype(np.int32)
        dist[idx < 0] = 0x7FFFFFFF
        g, ng, md = ctx.match_postfilter_host(idx, dist)
        ok_, on = oracle.good_match_filter(idx, dist)
        assert ng == on and np.array_equal(g, ok_)
    g, ng, md = ctx.match_postfilter_host(np.full(5, -1, np.int32), np.full(5, 0x7FFFFFFF, np.int32))
    assert ng == 0 and not g.any() and md == 50.0
    g, ng, md = ctx.match_postfilter_host(np.zeros(0, np.int32), np.zeros(0, np.int32))
    assert ng == 0
    # M6 on the resident descriptors of two slots and on host rows
    n1, n2 = len(kps[0]["level"]), len(kps[1]["level"])
    for n in (1, 63, 500, 1400):
        i1 = rng.integers(0, n1, n).astype(np.int32); i2 = rng.integers(0, n2, n).astype(np.int32)
        if n >= 63:
            i2[:20] = ctx.get_matches(0)[0][:20].clip(0)                     # some true matches: small distances -> the floor clamp
            i1[:20] = np.arange(20)
        for lo, hi, ratio in ((30, 80, 3.0), (30, 100, 3.0), (150, 200, 0.9), (1, 2, 50.0)):
            d, keep, ng, best = ctx.check_frame_descriptors(0, 1, i1, i2, lo, hi, ratio)
            od, okeep, ong, obest = oracle.check_frame_descriptors(kps[0]["desc"], kps[1]["desc"], i1, i2, lo, hi, ratio)
            assert np.array_equal(d, od) and np.array_equal(keep, okeep) and ng == ong and best == obest
            d2, keep2, ng2, best2 = ctx.check_descriptor_pairs(kps[0]["desc"][i1], kps[1]["desc"][i2], lo, hi, ratio)
            assert np.array_equal(d2, od) and np.array_equal(keep2, okeep) and ng2 == ong and best2 == obest
    d, keep, ng, best = ctx.check_frame_descriptors(0, 1, np.zeros(0, np.int32), np.zeros(0, np.int32))
    assert ng == 0 and len(d) == 0
    ctx.close()


def test_match_sets_equals_the_per_pair_calls(hip_lib, oracle):
    """ygz_hip_match_sets (all pairs of a keyframe window in one call: sets of different sizes, an empty set, a set matched against
    itself) against the oracle's BFMatcher + good-match rule, and against the one-pair entry points"""
    rng = np.random.default_rng(3)
    base = rng.integers(0, 256, (700, 32), dtype=np.uint8)
    sets = [base[:500].copy()]
    for n in (640, 333, 0, 700):
        d = base[:n].copy()
        flip = rng.random(d.shape) < 0.02                        # related descriptors: matches with small distances exist
        d[flip] ^= rng.integers(1, 256, int(flip.sum()), dtype=np.uint8)
        sets.append(d[rng.permutation(n)] if n else d)
    pq, pt = [0, 0, 0, 0, 2, 1], [1, 2, 3, 4, 1, 1]
    ctx = make_ctx(hip_lib, max_frames=8)
    res = ctx.match_sets(sets, pq, pt)
    for p, r in enumerate(res):
        q, t = sets[pq[p]], sets[pt[p]]
        oi, od, _ = oracle.bf_match(q, t, 1)
        og, ong = oracle.good_match_filter(oi, od)
        assert np.array_equal(r["idx"], oi) and np.array_equal(r["dist"], od), p
        assert np.array_equal(r["good"], og.astype(bool)) and r["n_good"] == ong, p
        if len(t):
            gi, gd = ctx.hamming_match(q, t, cross_check=1)
            gg, gn, gm = ctx.match_postfilter_host(gi, gd)
            assert np.array_equal(gi, r["idx"]) and np.array_equal(gg, r["good"]) and gn == r["n_good"] and gm == r["min_dis"], p
    assert res[2]["n_good"] == 0 and np.all(res[2]["idx"] == -1)       # empty train set
    ctx.close()


def _oracle_pair(oracle, seq, ref, cur, T_sa=None, depth_fn=None):
    """the oracle's composition of one frame pair of the offline run (T_ref = identity); depth_fn(frame, px) = Feature::_depth of
    the keypoints (default: the sequence's depth map at the pixel, what a full-resolution float64 depth image gives)"""
    w, h = seq.w, seq.h
    lv_r = oracle.pyramid(oracle.bgr2gray(seq.frame(ref)), 3)
    lv_c = oracle.pyramid(oracle.bgr2gray(seq.frame(cur)), 3)
    prm = oracle.default_params(w, h, 3)
    kr, kc = oracle.detect(lv_r, prm), oracle.detect(lv_c, prm)
    px = np.stack([kr["px"], kr["py"]], axis=1).astype(np.float64)
    dep = seq.depth(ref)[px[:, 1].astype(np.int64), px[:, 0].astype(np.int64)].astype(np.float64) if depth_fn is None else depth_fn(ref, px)
    out = dict(kr=kr, kc=kc, px=px, depth=dep)
    out["m_idx"], out["m_dist"], _ = oracle.bf_match(kc["desc"], kr["desc"], 1)
    out["m_good"], out["n_good"] = oracle.good_match_filter(out["m_idx"], out["m_dist"])
    pts = px.astype(np.float32)
    out["klt"] = oracle.klt_track(lv_r[0], lv_c[0], pts, pts)
    n_meas, T, st = oracle.sparse_align(lv_r, I7, lv_c, I7, px, dep, (dep > 0).astype(np.uint8))
    out["sa"] = (n_meas, T, list(st.iters_per_level)[:3])
    T_use = T if T_sa is None else T_sa                  # downstream stages are checked on the GPU's own pose (they branch on floats)
    pw, pred, cand = oracle.track_candidates(I7, T_use, px, dep, w, h)
    ok = np.zeros(len(px), bool); pxo = pred.copy(); sl = np.zeros(len(px), np.int32)
    ci = np.nonzero(cand)[0]
    ok[ci], pxo[ci], sl[ci] = oracle.find_direct_projection_n(lv_r, I7, lv_c, T_use, px[ci], dep[ci], kr["level"][ci], pred[ci])
    out.update(pw=pw, pred=pred, cand=cand, fdp_ok=ok, fdp_px=pxo, fdp_level=sl)
    th = oracle.se3_log(T_use)                           # [upsilon; omega]
    entry = np.concatenate([T_use[4:], th[3:]])
    out["po"] = oracle.optimize_current_pose_only(entry, pxo[ok], pw[ok])
    return out


def _check_pair(rec, o):
    """record of the offline run (keep=True) vs the oracle composition"""
    assert rec["n_kp"] == len(o["kc"]["level"])
    assert np.array_equal(rec["m_idx"], o["m_idx"]) and np.array_equal(rec["m_dist"], o["m_dist"])
    assert np.array_equal(rec["m_good"], o["m_good"]) and rec["n_good"] == o["n_good"]
    oout, ost, oerr = o["klt"]
    assert np.array_equal(rec["klt_status"], ost)
    m = ost.astype(bool)
    assert np.all(np.abs(rec["klt_pts"][m] - oout[m]).max(1) <= 1e-5 * np.maximum(1.0, np.abs(oout[m]).max(1)))       # north_star: 1e-5 relative
    n_meas, T, iters = o["sa"]
    assert rec["sa_n_meas"] == n_meas and rec["sa_iters"] == iters
    assert np.allclose(rec["T_sa"], T, rtol=1e-9, atol=1e-11)
    assert np.array_equal(rec["fdp_ok"], o["fdp_ok"])
    c = o["cand"]
    assert np.array_equal(rec["fdp_px"][c], o["fdp_px"][c]) and np.array_equal(rec["fdp_level"][c], o["fdp_level"][c])
    assert np.array_equal(rec["fdp_px"][~c], o["pred"][~c])
    pose, bad, depth, inl, rounds = o["po"]
    assert rec["po_inliers"] == inl and rec["po_rounds"] == rounds
    full_bad = np.ones(len(c), bool); full_bad[np.nonzero(o["fdp_ok"])[0]] = bad.astype(bool)
    assert np.array_equal(rec["po_bad"], full_bad)
    assert np.allclose(rec["po_pose"], pose, rtol=1e-7, atol=1e-9)


def test_track_handover_and_pose_only_vga(hip_lib, oracle):
    """the device-side chain sparse alignment -> adopt pose -> FindDirectProjection -> OptimizeCurrentPoseOnly of 3 VGA pairs
    against the oracle's composition (bit-exact candidates / pixels / flags on the GPU's own alignment pose)"""
    seq = synth.Sequence(4, 640, 480, seed=7, step=0.25)
    vo = offline.OfflineVO(640, 480, 4, chunk=4, kf_stride=2, window_kfs=2, keep=True)
    rec = vo.track_shard(seq.frame, seq.depth)
    for cur in (1, 2, 3):
        o = _oracle_pair(oracle, seq, cur - 1, cur, T_sa=rec[cur]["T_sa"])
        _check_pair(rec[cur], o)
        assert rec[cur]["po_inliers"] > 100
        # the refined relative pose is close to the ground truth of the synthetic sequence
        gt = offline.se3_mul(seq.poses[cur], offline.se3_inv(seq.poses[cur - 1]))
        assert np.abs(rec[cur]["T_rel"] - gt).max() < 5e-3
    vo.close()


N_SEQ = 16
# (window_kfs, depth image): windows aligned with the two shards / a window that straddles the shard boundary (its keyframe rows and
# relative poses cross ranks) with the quarter-resolution uint16 depth image of the bench
VARIANTS = {"aligned": dict(window_kfs=4, depth_div=1, depth_dtype="float64"), "straddling": dict(window_kfs=3, depth_div=4, depth_dtype="uint16")}


def _run_offline(rank, world, port, outdir, chunk, variant="aligned", pipeline_ba=True):
    import sys
    sys.path.insert(0, ROOT)
    v = VARIANTS[variant]
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)         # RCCL refuses two ranks on one device; gloo carries the same calls
    seq = synth.Sequence(N_SEQ, 1280, 720, seed=11, step=0.05)
    vo = offline.OfflineVO(1280, 720, N_SEQ, rank=rank, world=world, device=0, chunk=chunk, kf_stride=2, window_kfs=v["window_kfs"],
                           max_points=2000, keep=True, exchange_on_device=False, depth_div=v["depth_div"], depth_dtype=np.dtype(v["depth_dtype"]),
                           pipeline_ba=pipeline_ba)
    res = vo.run(seq.frame, seq.depth)
    vo.close()
    with open(os.path.join(outdir, "r%d_of_%d.pkl" % (rank, world)), "wb") as f:
        pickle.dump(res, f)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _same(a, b, path=""):
    if isinstance(a, dict):
        assert set(a) == set(b), path
        for k in a:
            _same(a[k], b[k], path + "/" + str(k))
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _same(x, y, path + "/%d" % i)
    elif isinstance(a, np.ndarray):
        assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True), path
    else:
        assert a == b or (a != a and b != b), path


@pytest.mark.parametrize("variant", ["aligned", "straddling"])
def test_offline_sharded_equals_unsharded_720p(hip_lib, oracle, tmp_path, variant):
    """BASELINE configs[4] at test size: a 16-frame 1280x720 sequence (a) unsharded in one chunk, (b) unsharded in chunks of 5
    frames (halo between chunks, two lanes, windows built while later chunks run), (c) unsharded with the BA round after the tracking
    instead of pipelined, (d) as 2 shards in 2 processes (gloo, both on this box's GPU): keypoints, matches, tracks, poses, BA windows
    and the gathered trajectory are IDENTICAL; two pairs are checked against the oracle."""
    import torch.multiprocessing as mp
    out = str(tmp_path)
    v = VARIANTS[variant]
    _run_offline(0, 1, 0, out, N_SEQ, variant)
    full = pickle.load(open(os.path.join(out, "r0_of_1.pkl"), "rb"))
    _run_offline(0, 1, 0, out, 5, variant)
    chunked = pickle.load(open(os.path.join(out, "r0_of_1.pkl"), "rb"))
    _same(full, chunked)
    _run_offline(0, 1, 0, out, 7, variant, pipeline_ba=False)
    late = pickle.load(open(os.path.join(out, "r0_of_1.pkl"), "rb"))
    _same(full, late)
    mp.spawn(_run_offline, args=(2, _free_port(), out, N_SEQ, variant), nprocs=2, join=True)
    parts = [pickle.load(open(os.path.join(out, "r%d_of_2.pkl" % r), "rb")) for r in range(2)]
    # every rank ends with the same global trajectory, relative poses, windows and keyframe poses as the unsharded run
    owners = [[w.pop("owner") for w in r["windows"]] for r in [full] + parts]
    if variant == "aligned":
        assert owners == [[0, 0], [0, 1], [0, 1]]              # the second window belongs to the rank that owns frame 8
    else:
        assert owners == [[0, 0, 0], [0, 0, 1], [0, 0, 1]]     # window [6, 8, 10]: anchor on rank 0, the other keyframes on rank 1
    for part in parts:
        for k in ("T_rel", "trajectory", "windows", "keyframe_pose", "built"):
            _same(full[k], part[k], k)
    merged = {}
    for part in parts:
        merged.update(part["records"])
    assert sorted(merged) == list(range(N_SEQ))
    assert sorted(parts[0]["records"]) == list(range(8)) and sorted(parts[1]["records"]) == list(range(8, 16))
    _same(full["records"], merged, "records")
    # sanity of the run itself
    assert len(full["windows"]) == (2 if variant == "aligned" else 3)
    for wi, w in enumerate(full["windows"]):
        chi0, chi1, its, n_edges = w["stats"]
        K, P, E = full["built"][wi]
        assert K == len(w["kfs"]) and E == n_edges and n_edges > 1000 and P > 300 and chi1 < chi0 and its >= 1
    seq = synth.Sequence(N_SEQ, 1280, 720, seed=11, step=0.05)
    gt = np.stack([offline.se3_mul(seq.poses[i], offline.se3_inv(seq.poses[0])) for i in range(N_SEQ)])
    assert np.abs(full["trajectory"] - gt).max() < 2e-2
    # the refined keyframe poses stay in the neighbourhood of the ground truth (the windows live in their anchor's gauge: composed with the
    # trajectory).  Only that: the scene is a plane seen over short baselines and one pose is constant, so BA may trade rotation against
    # translation and rescale a window; the trajectory above is the tracking result
    for f, T in full["keyframe_pose"].items():
        assert np.abs(T[:4] - gt[f][:4]).max() < 0.15, f
    # oracle parity on a subset: one pair inside shard 0 and the pair that straddles the shard boundary (cur = 8, ref = 7: the halo)
    dt = np.dtype(v["depth_dtype"])
    dfn = lambda f, px: offline.depth_at(offline.depth_image(seq.depth(f), v["depth_div"], dt), px, 1280, 720)
    for cur in (3, 8):
        o = _oracle_pair(oracle, seq, cur - 1, cur, T_sa=full["records"][cur]["T_sa"], depth_fn=dfn)
        _check_pair(full["records"][cur], o)
        assert np.array_equal(full["records"][cur - 1]["kp"]["depth"], o["depth"])


@pytest.mark.parametrize("obs_mode", ["direct", "match"])
def test_device_built_window_equals_host_built(hip_lib, oracle, obs_mode):
    """ygz_hip_ba_build_windows against the host restatement tests/window_ref.py: build_window_host on the keyframes of a tracked sequence: the same
    map points (bit-equal), vertices (1e-13: the host chains the relative poses through numpy), and -- with the device's state installed
    in the host-built graph -- bit-identical linearisations (every edge in the same row, the same observation, the same pose).
    obs_mode "direct": the host takes its observations from per-pair ygz_hip_find_direct_projection calls on a context that holds the
    keyframes (FindCandidates' test and the prediction in numpy); "match": from ygz_hip_match_sets."""
    n = 10
    seq = synth.Sequence(n, 640, 480, seed=3, step=0.2)
    vo = offline.OfflineVO(640, 480, n, chunk=n, kf_stride=2, window_kfs=4, max_points=700, keep=True, pipeline_ba=False, obs_mode=obs_mode)
    rec = vo.track_shard(seq.frame, seq.depth)
    assert vo.wins == [[0, 2, 4, 6]] and vo.mine == [0]          # frames 8, 9: a lone keyframe makes no window
    vo._ba_launch(vo.mine, optimize=False)
    K, P, E, Kf = (int(x) for x in vo.ba.ba_get_stats(0, 1, want_stats=False)[1][0])
    kf_tab = {f: rec[f]["kp"] for f in vo.wins[0]}
    T_rel = np.stack([rec[f].get("T_rel", offline.I7) for f in range(n)])
    cam = vo.ba.params
    if obs_mode == "direct":
        kc = hip_lib.HipContext(width=640, height=480, levels=3, max_frames=4)
        slot = {f: k for k, f in enumerate(vo.wins[0])}
        for f, k in slot.items():
            kc.upload_bgr(k, seq.frame(f))
        kc.build_pyramid(0, 4, from_bgr=True)

        def direct(ref, cur, T, px_ref, depth, level, px_cur):
            ok, px, _ = kc.find_direct_projection(slot[ref], offline.I7, slot[cur], T, px_ref, depth, level, px_cur)
            return ok, px
        h = window_ref.build_window_host(kf_tab, vo.wins[0], T_rel, float(cam.fx), float(cam.fy), float(cam.cx), float(cam.cy), 700, direct=direct,
                                      width=640, height=480)
        kc.close()
    else:
        h = window_ref.build_window_host(kf_tab, vo.wins[0], T_rel, float(cam.fx), float(cam.fy), float(cam.cx), float(cam.cy), 700, vo.ba.match_sets)
    assert (K, P, E, Kf) == (4, len(h["points"]), len(h["obs"]), 3) and P > 200 and E > 2 * P
    poses, points = vo.ba.ba_get_state(0, 4, 700)
    assert np.array_equal(points[:P], h["points"])
    assert np.allclose(poses[:K], h["poses"], rtol=0, atol=1e-13) and np.all(poses[0] == 0)
    vo.ba.ba_upload(100, poses[:K], h["fixed"], points[:P], h["edge_pose"], h["edge_point"], h["obs"])
    vo.ba.ba_linearize_resident(0, 1); vo.ba.ba_linearize_resident(100, 1)
    a, b = vo.ba.ba_download(0, K, P, E), vo.ba.ba_download(100, K, P, E)
    for k in ("Hpp", "bp", "Hll", "bl", "Hpl", "err", "chi2_edge"):
        assert np.array_equal(a[k], b[k]), k
    assert a["chi2"] == b["chi2"] and a["chi2"] > 0
    if obs_mode == "direct":                                      # sub-pixel aligned observations: the initial graph is already consistent to a few pixels^2 per edge
        assert a["chi2"] / E < 25.0, a["chi2"] / E
    # ... and the resident LM on the two graphs walks the same trials
    sa = vo.ba.ba_optimize_resident(0, 1, 10)[0]
    sb = vo.ba.ba_optimize_resident(100, 1, 10)[0]
    assert (sa.iterations, sa.lm_trials, sa.chi2_initial, sa.chi2_final) == (sb.iterations, sb.lm_trials, sb.chi2_initial, sb.chi2_final)
    pa, qa = vo.ba.ba_get_state(0, 4, 700); pb, qb = vo.ba.ba_get_state(100, K, P)
    assert np.array_equal(pa[:K], pb) and np.array_equal(qa[:P], qb) and sa.chi2_final < sa.chi2_initial
    # the inlier test of BA.cpp:503-515 on both: the same counts; switched off, the outliers leave the next linearisation
    vo.ba.ba_mark_outliers(0, 1, 5.991, disable=True); vo.ba.ba_mark_outliers(100, 1, 5.991, disable=False)
    oa, ob = vo.ba.ba_get_outlier_stats(0, 1)[0], vo.ba.ba_get_outlier_stats(100, 1)[0]
    assert np.array_equal(oa, ob) and oa[0] == E and 0 <= oa[1] < E and oa[3] <= oa[2]
    vo.ba.ba_linearize_resident(0, 1)
    c = vo.ba.ba_download(0, K, P, E)
    assert int((c["chi2_edge"] > 0).sum()) == E - int(oa[1]) or oa[1] == 0
    vo.close()


def test_degenerate_window_is_reported_not_optimised(hip_lib):
    """a window whose anchor has no feature with depth (here: a sequence without depth) has no map point: the device builds P = E = 0, the
    resident LM leaves at once with ZERO iterations (not twenty failed pivots that look like a finished run) and the run says so"""
    n = 6
    seq = synth.Sequence(n, 640, 480, seed=5, step=0.2)
    vo = offline.OfflineVO(640, 480, n, chunk=n, kf_stride=2, window_kfs=3)
    res = vo.run(seq.frame, lambda f: np.zeros((480, 640)))
    assert len(res["windows"]) == 1 and res["built"][0][1:] == (0, 0)
    w = res["windows"][0]
    assert w["lm"] == dict(iterations=0, trials=0, degenerate=True) and vo.degenerate_windows == [0]
    assert np.all(w["poses"][0] == 0) and np.all(np.isfinite(w["poses"]))
    vo.close()


def test_timed_out_window_is_rebuilt_and_solved_by_one_workgroup(hip_lib):
    """the retry path of a resident-LM team that timed out at a barrier (ygz_offline_retry_windows, what the driver calls for every window
    whose statistics read "no result"): the window is rebuilt and solved by a single workgroup -- bit-identical to the team's result, because
    the points are reduced in fixed parts whatever the team size"""
    n = 10
    seq = synth.Sequence(n, 640, 480, seed=3, step=0.2)
    vo = offline.OfflineVO(640, 480, n, chunk=n, kf_stride=2, window_kfs=4, max_points=700)
    res = vo.run(seq.frame, seq.depth)
    before = vo.ba.ba_pack_states(0, 1, vo.S)
    assert before[0, -12 + 3] >= 1 and vo.lm_retries == 0
    assert np.array_equal(before[0], res["windows"][0]["state"])
    vo.retry_windows([0])
    after = vo.ba.ba_pack_states(0, 1, vo.S)
    assert vo.lm_retries == 1 and np.array_equal(before, after)
    vo.close()


def test_rccl_path_with_a_communicator_of_one_rank(hip_lib):
    """the driver's RCCL path (librccl.so.1 bound at run time, ncclCommInitRank, device exchange buffers, k_ba_pack straight into the send
    buffer, ncclAllGather on the BA context's stream) on the ONE GPU of the test box: a communicator of one rank must give the bits of the
    plain single-rank run"""
    n = 12
    seq = synth.Sequence(n, 640, 480, seed=3, step=0.2)
    out = []
    for single in (False, True):
        vo = offline.OfflineVO(640, 480, n, chunk=5, kf_stride=2, window_kfs=3, max_points=700, rccl_single=single)
        res = vo.run(seq.frame, seq.depth)
        out.append((res, vo.backend))
        vo.close()
    assert [b for _, b in out] == ["single rank", "rccl"]
    for k in ("T_rel", "trajectory", "windows", "built", "records"):
        _same(out[0][0][k], out[1][0][k], k)
    assert len(out[0][0]["windows"]) == 2 and out[0][0]["windows"][0]["lm"]["iterations"] >= 1


N_LONG = 128
N_FULL = 1024                                                   # BASELINE configs[4]: 1024 frames of 1280x720 sharded 8 ways


def _render_long(i, n=N_LONG):
    seq = synth.Sequence(n, 1280, 720, seed=11, step=0.02)
    return seq.frame(i), offline.depth_image(seq.depth(i), 4, np.uint16)


def _render_full(i):
    return _render_long(i, N_FULL)


def _run_long(rank, world, port, outdir, defer=None, n_frames=N_LONG, window_kfs=6):
    import sys
    sys.path.insert(0, ROOT)
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from ygz_slam_amd import _lib
    N_LONG = n_frames                                           # (shadows the module constant: configs[4]'s full length in the 8-rank test)
    s0, cnt, halo = ydist.shard_frames(N_LONG, rank, world)
    base = s0 - halo
    bgr = _lib.PinnedArray((cnt + halo, 720, 1280, 3), np.uint8); dimg = _lib.PinnedArray((cnt + halo, 180, 320), np.uint16)
    bgr.array[:] = np.load(os.path.join(outdir, "bgr.npy"), mmap_mode="r")[base:base + cnt + halo]
    dimg.array[:] = np.load(os.path.join(outdir, "depth.npy"), mmap_mode="r")[base:base + cnt + halo]
    vo = offline.OfflineVO(1280, 720, N_LONG, rank=rank, world=world, device=0, chunk=24, kf_stride=8, window_kfs=window_kfs, max_points=2000,
                           exchange_on_device=False, depth_div=4, depth_dtype=np.uint16, defer_gaps=defer)      # defer: the plan of the shard's chunks (chunk_plan)
    block = lambda frames: (bgr.array[frames[0] - base:frames[-1] + 1 - base], dimg.array[frames[0] - base:frames[-1] + 1 - base])   # page-locked: every copy is asynchronous
    res = vo.run(None, None, block)
    vo.close()
    with open(os.path.join(outdir, "long_r%d_of_%d%s.pkl" % (rank, world, "" if defer is None else "_defer%d" % defer)), "wb") as f:
        pickle.dump({k: res[k] for k in ("T_rel", "trajectory", "windows", "keyframe_pose", "built")}, f)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def test_offline_128_frames_720p_two_ranks(hip_lib, tmp_path):
    """configs[4] at 128 frames of 1280x720, the bench's settings (chunks on two lanes, quarter-resolution uint16 depth, asynchronous
    block source, windows pipelined behind the tracking): 2 shards on this box's GPU reproduce the unsharded run exactly -- including
    the window whose keyframes lie on both sides of the shard boundary."""
    import multiprocessing
    import torch.multiprocessing as mp
    out = str(tmp_path)
    with multiprocessing.get_context("spawn").Pool(min(16, os.cpu_count() or 1)) as pool:
        fr = pool.map(_render_long, range(N_LONG), chunksize=4)
    np.save(os.path.join(out, "bgr.npy"), np.stack([f[0] for f in fr])); np.save(os.path.join(out, "depth.npy"), np.stack([f[1] for f in fr]))
    del fr
    _run_long(0, 1, 0, out)
    full = pickle.load(open(os.path.join(out, "long_r0_of_1.pkl"), "rb"))
    # the order of the chunks is free (every pair is solved from the identity): the plain plan and the plan that processes the keyframe-free
    # frames behind the windows at the end, in chunks of several ranges, give the same bits
    for defer in (0, 2):
        _run_long(0, 1, 0, out, defer=defer)
        other = pickle.load(open(os.path.join(out, "long_r0_of_1_defer%d.pkl" % defer), "rb"))
        _same(full, other, "defer %d" % defer)
    mp.spawn(_run_long, args=(2, _free_port(), out), nprocs=2, join=True)
    owners = None
    for r in range(2):
        part = pickle.load(open(os.path.join(out, "long_r%d_of_2.pkl" % r), "rb"))
        owners = [w.pop("owner") for w in part["windows"]]
        for w in full["windows"]:
            w.pop("owner", None)
        _same(full, part, "rank %d" % r)
    assert owners == [0, 0, 1]                                    # window [48 .. 88] is anchored on rank 0 and ends on rank 1
    seq = synth.Sequence(N_LONG, 1280, 720, seed=11, step=0.02)
    gt = np.stack([offline.se3_mul(seq.poses[i], offline.se3_inv(seq.poses[0])) for i in range(N_LONG)])
    assert np.abs(full["trajectory"] - gt).max() < 2e-2
    for w in full["windows"]:
        assert w["stats"][1] < w["stats"][0] and w["stats"][3] > 5000
        # observations by direct projection (LocalMapping.cpp:82-120): after optimize(20) the inlier edges sit at sub-pixel reprojection error and
        # few edges fail the chi2 > 5.991 test of BA.cpp:503-515
        inl = w["inliers"]
        assert inl["edges"] == w["stats"][3] and inl["outliers"] < 0.2 * inl["edges"], inl
        assert inl["chi2_inliers"] / max(1, inl["edges"] - inl["outliers"]) < 2.0, inl
    # ... and the BA round moves the keyframes towards the ground truth (poses relative to each window's anchor)
    pe = offline.window_pose_errors(full["windows"], full["trajectory"], gt)
    assert pe["t_after"].mean() < pe["t_before"].mean() and pe["r_after"].mean() < pe["r_before"].mean(), {k: float(v.mean()) for k, v in pe.items()}


def test_offline_1024_frames_8_ranks_emulated(hip_lib, tmp_path):
    """BASELINE configs[4] at its stated shape -- 1024 frames of 1280x720, 8 shards of 128 frames -- with the 8 ranks EMULATED on this box's one
    GPU: eight processes, each driving its own context through the C++ driver, the exchange steps carried by gloo instead of RCCL (RCCL refuses
    two ranks on one device; the calls and their payloads are the same).  Every rank ends with the trajectory, relative poses, BA windows, window
    owners and keyframe poses of the unsharded 1024-frame run, bit for bit.  (What it cannot show is time: the scaling curve stays a prediction,
    DESIGN.md section 6.)"""
    import multiprocessing
    import torch.multiprocessing as mp
    out = str(tmp_path)
    with multiprocessing.get_context("spawn").Pool(min(32, os.cpu_count() or 1)) as pool:
        fr = pool.map(_render_full, range(N_FULL), chunksize=8)
    np.save(os.path.join(out, "bgr.npy"), np.stack([f[0] for f in fr])); np.save(os.path.join(out, "depth.npy"), np.stack([f[1] for f in fr]))
    del fr
    for window_kfs, n_windows, n_straddle_min in ((8, 16, 0), (6, 22, 4)):      # the bench's windows (64 frames: none crosses a shard boundary) / windows of 48 frames (5 do)
        _run_long(0, 1, 0, out, None, N_FULL, window_kfs)
        full = pickle.load(open(os.path.join(out, "long_r0_of_1.pkl"), "rb"))
        mp.spawn(_run_long, args=(8, _free_port(), out, None, N_FULL, window_kfs), nprocs=8, join=True)
        for w in full["windows"]:
            w.pop("owner", None)
        owners = None
        for r in range(8):
            part = pickle.load(open(os.path.join(out, "long_r%d_of_8.pkl" % r), "rb"))
            own_r = [w.pop("owner") for w in part["windows"]]
            assert owners is None or own_r == owners                 # every rank knows the same owner of every window
            owners = own_r
            _same(full, part, "window_kfs %d rank %d" % (window_kfs, r))
        # a window belongs to the rank that holds its anchor keyframe: all eight ranks own some, in shard order
        assert sorted(set(owners)) == list(range(8)) and owners == sorted(owners), owners
        for w, o in zip(full["windows"], owners):
            s0, cnt, _ = ydist.shard_frames(N_FULL, o, 8)
            assert s0 <= w["kfs"][0] < s0 + cnt, (w["kfs"], o)
        n_straddle = sum(1 for w in full["windows"] if (w["kfs"][0] // 128) != (w["kfs"][-1] // 128))
        assert len(full["windows"]) == n_windows and n_straddle >= n_straddle_min and (n_straddle_min > 0 or n_straddle == 0), (len(full["windows"]), n_straddle)
    assert len(full["trajectory"]) == N_FULL
    seq = synth.Sequence(N_FULL, 1280, 720, seed=11, step=0.02)
    gt = np.stack([offline.se3_mul(seq.poses[i], offline.se3_inv(seq.poses[0])) for i in range(N_FULL)])
    assert np.abs(full["trajectory"] - gt).max() < 5e-2


def test_create_map_points_triangulation_loop(hip_lib, oracle):
    """the triangulation loop of LocalMapping::CreateNewMapPoints (LocalMapping.cpp:416-495) for ~900 matched pairs of two keyframes:
    every exit of the loop is hit (parallel rays, rejected triangulations, failed direct projection, reprojection error, success);
    codes, refined pixels and search levels bit-exact, depths and map points to 1e-12"""
    seq = synth.Sequence(6, 640, 480, seed=9, step=0.6)
    ctx = make_ctx(hip_lib, max_frames=2)
    f1, f2 = 0, 5
    for s, f in enumerate((f1, f2)):
        ctx.upload_bgr(s, seq.frame(f))
    ctx.build_pyramid(0, 2, from_bgr=True); ctx.detect(0, 2)
    k1 = ctx.get_keypoints(0)
    T1, T2 = seq.poses[f1], seq.poses[f2]
    px1 = k1["px"]; lv1 = k1["level"]
    z = seq.depth(f1)[px1[:, 1].astype(int), px1[:, 0].astype(int)]
    cam = oracle.camera()
    fx, fy, cx, cy = float(cam.fx), float(cam.fy), float(cam.cx), float(cam.cy)
    pc1 = np.stack([(px1[:, 0] - cx) * z / fx, (px1[:, 1] - cy) * z / fy, z], 1)
    pc2 = offline.se3_act(offline.se3_mul(T2, offline.se3_inv(T1)), pc1)
    px2 = np.stack([fx * pc2[:, 0] / pc2[:, 2] + cx, fy * pc2[:, 1] / pc2[:, 2] + cy], 1)
    rng = np.random.default_rng(1)
    px2 += rng.normal(0, 0.7, px2.shape)                         # what a descriptor match gives: the right corner, roughly
    n = len(px2)
    px2[::17] += rng.uniform(15, 40, (len(px2[::17]), 2))       # wrong matches: direct projection fails or the reprojection test does
    px2[5::23] = px1[5::23] + rng.normal(0, 0.05, px1[5::23].shape)      # (almost) the same ray in both frames -> cos >= 0.9998 for most
    lv = [oracle.pyramid(oracle.bgr2gray(seq.frame(f)), 3) for f in (f1, f2)]
    o = oracle.create_map_points(lv[0], T1, lv[1], T2, px1, lv1, px2)
    g = ctx.create_map_points(0, T1, 1, T2, px1, lv1, px2)
    assert np.array_equal(g["code"], o["code"]) and g["created"] == o["created"]
    assert np.array_equal(g["px2"], o["px2"]) and np.array_equal(g["search_level"], o["search_level"])
    m = o["code"] == 0
    assert np.allclose(g["depth1"][m], o["depth1"][m], rtol=1e-12) and np.allclose(g["depth2"][m], o["depth2"][m], rtol=1e-12)
    assert np.allclose(g["pos_world"][m], o["pos_world"][m], rtol=1e-12, atol=1e-13)
    hist = np.bincount(o["code"], minlength=6)
    assert hist[0] > 0.5 * n and hist[1] > 0 and hist[3] > 0 and (hist[4] + hist[5] + hist[2]) > 0, hist
    # the new map points are where the scene is: depth of the triangulated point vs the rendered depth
    assert np.median(np.abs(g["depth1"][m] - z[m]) / z[m]) < 0.05
    e = ctx.create_map_points(0, T1, 1, T2, np.zeros((0, 2)), np.zeros(0, np.int32), np.zeros((0, 2)))
    assert e["created"] == 0 and len(e["code"]) == 0
    ctx.close()


def test_depth_filter_update_seeds(hip_lib, oracle):
    """the legacy SVO depth filter (DepthFilter::UpdateSeeds + FindEpipolarMatchDirect, src/optimizer.cpp:537-735, src/utils.cpp:330-661):
    seeds of one keyframe updated by four later frames, TWO chains side by side: the oracle iterates its own carried seed list and the GPU
    iterates ITS own (nothing is fed back from the oracle).  States, matched pixels and depths are equal (integer ZMSSD search and float
    Align2D chains are reproduced exactly); the Bayesian parameters mu, sigma2, a, b agree to 1e-6 relative after every frame: every
    operation of UpdateSeed is a correctly rounded IEEE float operation on both sides, including expf (the double exponential rounded
    once, oracle/mapping.c::yo_expf_cr) and sqrtf."""
    seq = synth.Sequence(6, 640, 480, seed=4, step=0.45)
    ctx = make_ctx(hip_lib, max_frames=6)
    for s in range(6):
        ctx.upload_bgr(s, seq.frame(s))
    ctx.build_pyramid(0, 6, from_bgr=True); ctx.detect(0, 1)
    k0 = ctx.get_keypoints(0)
    lv = [oracle.pyramid(oracle.bgr2gray(seq.frame(f)), 3) for f in range(6)]
    n = len(k0["level"])
    zt = seq.depth(0)[k0["px"][:, 1].astype(int), k0["px"][:, 0].astype(int)]
    rng = np.random.default_rng(2)
    depth_mean, depth_min = float(zt.mean()), float(zt.min()) * 0.6
    mu0 = (1.0 / (zt * rng.uniform(0.75, 1.3, n))).astype(np.float32)            # Seed::Seed uses 1 / depth_mean; spread it so that long and short
    z_range = np.full(n, 1.0 / depth_min, np.float32)                             # epipolar segments both occur
    seeds = dict(kp=k0["px"].astype(np.float32), octave=k0["level"], ref=np.zeros(n, np.int32), frame_id=np.zeros(n, np.uint64),
                 a=np.full(n, 10, np.float32), b=np.full(n, 10, np.float32), mu=mu0, z_range=z_range, sigma2=(z_range * z_range / 36).astype(np.float32))
    seeds["frame_id"][::97] = 9                                                   # seeds of a frame "from the future": int - unsigned wraps -> erased as too old
    og, gg = dict(seeds), dict(seeds)                                             # the oracle's and the GPU's own carried lists
    seen = set()
    total_conv = 0
    err0 = np.abs(1.0 / og["mu"] - zt)
    exact = []
    for f in (1, 2, 3, 5):
        o = oracle.depth_filter_update(lv[f], seq.poses[f], [lv[0]], [seq.poses[0]], og, batch_counter=1)
        g = ctx.depth_filter_update(f, seq.poses[f], [0], [seq.poses[0]], gg, batch_counter=1)
        assert np.array_equal(g["state"], o["state"]), np.nonzero(g["state"] != o["state"])
        assert g["updated"] == o["updated"]
        assert np.array_equal(g["matched_px"], o["matched_px"]) and np.allclose(g["z"], o["z"], rtol=1e-12)
        for k in ("mu", "a", "b", "sigma2"):
            assert np.allclose(g[k], o[k], rtol=1e-6, atol=1e-12), (k, f, float(np.nanmax(np.abs(g[k] - o[k]) / np.maximum(np.abs(o[k]), 1e-30))))
            exact.append(float(np.mean(g[k] == o[k])))
        m = o["state"] == 5
        assert np.allclose(g["pos_world"][m], o["pos_world"][m], rtol=1e-6)
        seen |= set(np.unique(o["state"]).tolist())
        total_conv += int(m.sum())
        keep = np.isin(o["state"], (0, 1, 2, 3))                                  # erased seeds leave the list
        for k in ("a", "b", "mu", "sigma2"):
            og[k] = o[k][keep]; gg[k] = g[k][keep]                                # each side keeps ITS values
        for k in ("kp", "octave", "ref", "frame_id", "z_range"):
            og[k] = og[k][keep]; gg[k] = gg[k][keep]
        zt, err0 = zt[keep], err0[keep]
    assert min(exact) > 0.99, exact                                               # in fact (almost) every parameter is bit-equal
    assert {0, 3, 4}.issubset(seen), seen
    # the filter does its job: after four frames the depth of the surviving seeds is closer to the rendered depth than the prior was
    assert np.median(np.abs(1.0 / gg["mu"] - zt)) < 0.5 * np.median(err0)
    e = ctx.depth_filter_update(1, seq.poses[1], [0], [seq.poses[0]], {k: v[:0] for k, v in og.items()}, batch_counter=1)
    assert e["updated"] == 0
    ctx.close()
