"""World-size-2 CPU tests (gloo) of the N>1 host path: frame/pair sharding, the BA-window broadcast, the
trajectory all-gather, and that a sharded run of the hot path (here computed by the oracle, there is no GPU in
the build container) gives exactly the unsharded results."""
import os
import socket
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from ygz_slam_amd import dist as ydist
from ygz_slam_amd import synth, offline
import plan_ref


def test_shards_partition_frames_and_pairs():
    for n in (1, 7, 8, 128, 1024):
        for world in (1, 2, 3, 8):
            seen, pairs = [], []
            for r in range(world):
                s, c, halo = ydist.shard_frames(n, r, world)
                seen += list(range(s, s + c))
                assert halo == (1 if s > 0 and c > 0 else 0)
                pairs += ydist.shard_pairs(n, r, world)
            assert seen == list(range(n))
            assert pairs == [(i, i - 1) for i in range(1, n)]
            sizes = [ydist.shard_frames(n, r, world)[1] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, n_frames, ret):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle.pyoracle import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = Oracle()
    # the BA window lives on rank 0; everybody else starts from garbage and must end with rank 0's state
    f = synth.ba_window(4, 50, seed=3)
    pts = f["points"] if rank == 0 else np.full_like(f["points"], -7.0)
    poses = f["poses"] if rank == 0 else np.zeros_like(f["poses"])
    pts, poses = ydist.broadcast_map(pts, poses, src=0)
    chi2 = o.ba_linearize(poses, f["fixed"], pts, f["edge_pose"], f["edge_point"], f["obs"])["chi2"]
    # sharded extraction + matching of a small sequence
    tex, m = synth.make_texture(4, 160, 120, margin=40)
    tr = synth.trajectory(n_frames, 5, 0.3)
    prm = o.default_params(160, 120, 3)
    start, count, halo = ydist.shard_frames(n_frames, rank, world)
    kps = {}
    for i in range(start - halo, start + count):
        img, _ = synth.render(tex, m, tr[i], 160, 120, 1.0, 40 + i)
        kps[i] = o.detect(o.pyramid(img, 3), prm)
    nmatch = [int((o.bf_match(kps[c]["desc"], kps[r]["desc"], 1)[0] >= 0).sum()) for c, r in ydist.shard_pairs(n_frames, rank, world)]
    traj = ydist.gather_trajectories(tr[start:start + count], n_frames, rank, world)
    all_n = [None] * world
    dist.all_gather_object(all_n, nmatch)
    if rank == 0:
        ret["chi2"] = chi2; ret["traj"] = traj; ret["nmatch"] = sum(all_n, [])
    ret["chi2_%d" % rank] = chi2
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo(oracle):
    n_frames = 5
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), n_frames, ret), nprocs=2, join=True)
    f = synth.ba_window(4, 50, seed=3)
    ref = oracle.ba_linearize(f["poses"], f["fixed"], f["points"], f["edge_pose"], f["edge_point"], f["obs"])["chi2"]
    assert ret["chi2_0"] == ref and ret["chi2_1"] == ref            # broadcast delivered rank 0's window bit-exactly
    tr = synth.trajectory(n_frames, 5, 0.3)
    assert np.array_equal(ret["traj"], tr)                           # ragged shards (3 + 2 frames) gathered in order
    # unsharded reference
    tex, m = synth.make_texture(4, 160, 120, margin=40)
    prm = oracle.default_params(160, 120, 3)
    ks = [oracle.detect(oracle.pyramid(synth.render(tex, m, tr[i], 160, 120, 1.0, 40 + i)[0], 3), prm) for i in range(n_frames)]
    exp = [int((oracle.bf_match(ks[i]["desc"], ks[i - 1]["desc"], 1)[0] >= 0).sum()) for i in range(1, n_frames)]
    assert list(ret["nmatch"]) == exp


# ---- host logic of the offline run (ygz_slam_amd/host/ygz_offline.cpp, bound by ygz_slam_amd/offline.py); its GPU half is tests/test_gpu_offline.py ------------------
def test_offline_windows_and_owners():
    assert offline.keyframes(17, 8) == [0, 8, 16]
    assert offline.ba_windows(1024, 8, 8) == [list(range(64 * w, 64 * w + 64, 8)) for w in range(16)]
    assert offline.ba_windows(16, 2, 4) == [[0, 2, 4, 6], [8, 10, 12, 14]]
    assert offline.ba_windows(18, 2, 4) == [[0, 2, 4, 6], [8, 10, 12, 14]]          # a trailing window needs two keyframes
    assert offline.ba_windows(20, 2, 4)[-1] == [16, 18]
    for n, world in ((1024, 8), (16, 2), (17, 3)):
        own = [offline.frame_owner(f, n, world) for f in range(n)]
        assert own == sorted(own) and set(own) == set(range(world))
        for r in range(world):
            s, c, _ = ydist.shard_frames(n, r, world)
            assert own[s:s + c] == [r] * c
    # 1024 frames / 8 ranks / stride 8 / 8 keyframes per window: every rank owns exactly two whole windows
    wins = offline.ba_windows(1024, 8, 8)
    assert [offline.frame_owner(w[0], 1024, 8) for w in wins] == [w // 2 for w in range(16)]


def test_offline_se3_helpers_match_the_oracle(oracle):
    rng = np.random.default_rng(2)
    for _ in range(50):
        a = synth.se3_exp(rng.normal(0, 0.4, 6)); b = synth.se3_exp(rng.normal(0, 0.4, 6))
        assert np.allclose(offline.se3_mul(a, b), oracle.se3_mul(a, b), rtol=0, atol=1e-15)
        assert np.allclose(offline.se3_inv(a), oracle.se3_inv(a), rtol=0, atol=1e-15)
        p = rng.normal(0, 2, 3)
        assert np.allclose(offline.se3_act(a, p), oracle.se3_act(a, p), rtol=0, atol=1e-14)
        assert np.allclose(offline.se3_act(a, np.stack([p, 2 * p])), np.stack([oracle.se3_act(a, p), oracle.se3_act(a, 2 * p)]), atol=1e-14)
        lg = oracle.se3_log(a)                                   # Sophus order [upsilon; omega]
        g = offline.se3_log_g2o(a)
        assert np.allclose(g, np.concatenate([lg[3:], lg[:3]]), rtol=0, atol=1e-13)
        assert np.allclose(offline.se3_exp_g2o(g), a, rtol=0, atol=1e-13)
    T_rel = np.stack([synth.se3_exp(rng.normal(0, 0.05, 6)) for _ in range(9)])
    tr = offline.chain(T_rel)
    ref = np.array([0, 0, 0, 1.0, 0, 0, 0])
    assert np.array_equal(tr[0], ref)
    for i in range(1, 9):
        ref = oracle.se3_mul(T_rel[i], ref)
        assert np.allclose(tr[i], ref, rtol=0, atol=1e-14)


def _exchange_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # the exchange of the C++ driver (ygz_offline_ragged_all_gather in libygz_host.so) over gloo: 5 BA windows with contiguous ownership
    # [0, 0, 1, 1, 1] -- rows of 7 doubles here, 48 KB state rows in a run -- ...
    counts = [2, 3]
    first = sum(counts[:rank])
    mine = np.stack([np.arange(7, dtype=np.float64) + 10 * (first + i) + 0.5 for i in range(counts[rank])])
    buf = offline.ragged_all_gather(mine, counts)
    # ... and the relative poses of ragged shards (5 frames: 3 + 2), followed by the chain, as ygz_offline_gather does
    n = 5
    rng = np.random.default_rng(9)
    T_rel_all = np.stack([synth.se3_exp(rng.normal(0, 0.05, 6)) for _ in range(n)])
    s, c, _ = ydist.shard_frames(n, rank, world)
    got = offline.ragged_all_gather(T_rel_all[s:s + c], [ydist.shard_frames(n, r, world)[1] for r in range(world)])
    # a rank that owns nothing contributes an empty block
    lone = offline.ragged_all_gather(np.arange(6, dtype=np.int32).reshape(2, 3) if rank == 1 else np.zeros((0, 3), np.int32), [0, 2])
    ret[rank] = (buf, got, lone)
    dist.barrier()
    dist.destroy_process_group()


def test_offline_exchange_world_size_2_gloo():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_exchange_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    exp = np.stack([np.arange(7) + 10 * i + 0.5 for i in range(5)])
    rng = np.random.default_rng(9)
    T_rel_all = np.stack([synth.se3_exp(rng.normal(0, 0.05, 6)) for _ in range(5)])
    for r in range(2):
        buf, got, lone = ret[r]
        assert np.array_equal(buf, exp)                          # every rank holds every owner's rows
        assert np.array_equal(got, T_rel_all)
        assert np.array_equal(offline.chain(got), offline.chain(T_rel_all))
        assert np.array_equal(lone, np.arange(6, dtype=np.int32).reshape(2, 3))


def test_offline_driver_partition_equals_the_python_statement():
    """the C++ driver's own partition (ygz_offline_shard) and plan (ygz_offline_plan for a parameter block) against dist.shard_frames and
    the plan built from this module's window list"""
    import ctypes as C
    lib = offline.host_lib()
    for n in (1, 7, 8, 128, 1024, 1000):
        for world in (1, 2, 3, 8):
            for r in range(world):
                a, b, c = C.c_int(), C.c_int(), C.c_int()
                assert lib.ygz_offline_shard(n, r, world, C.byref(a), C.byref(b), C.byref(c)) == 0
                assert (a.value, b.value, c.value) == ydist.shard_frames(n, r, world)
    for n, world, chunk, stride, wk, defer in ((1024, 1, 128, 8, 8, -1), (1024, 8, 32, 8, 8, -1), (128, 2, 24, 8, 6, 2), (16, 2, 5, 2, 3, 0), (900, 1, 128, 8, 8, -1)):
        wins = offline.ba_windows(n, stride, wk)
        for r in range(world):
            p = offline.OffParams()
            lib.ygz_offline_default_params(C.byref(p))
            p.n_frames, p.rank, p.world, p.chunk, p.kf_stride, p.window_kfs, p.defer_gaps = n, r, world, chunk, stride, wk, defer
            out = np.zeros((4096, 3), np.int32); k = C.c_int(0)
            assert lib.ygz_offline_plan(C.byref(p), out.ctypes.data_as(C.POINTER(C.c_int32)), 4096, C.byref(k)) == 0
            s, c, _ = ydist.shard_frames(n, r, world)
            inside = sum(1 for w in wins if w[0] >= s and w[-1] < s + c)
            d = (13 if c >= 768 else (inside if c <= 160 else 0)) if defer < 0 else defer      # the driver's rule (ygz_offline.cpp: make_plan)
            exp = plan_ref.cpp_chunk_plan(s, s + c, chunk, True, stride, wins, d)
            got = {}
            for ci, a, b in out[:k.value].tolist():
                got.setdefault(ci, []).append((a, b))
            assert [tuple(got[i]) for i in sorted(got)] == exp


def test_offline_plan_equals_the_python_restatement():
    """chunk_schedule / chunk_plan of ygz_offline.cpp against their round-4 interpreter form (tests/plan_ref.py) on random shards"""
    import plan_ref
    rng = np.random.default_rng(77)
    for _ in range(400):
        kf_stride = int(rng.choice([2, 4, 8, 8, 8, 16]))
        window_kfs = int(rng.integers(2, 9))
        n_total = int(rng.integers(kf_stride * window_kfs + 1, 1500))
        world = int(rng.choice([1, 1, 2, 3, 4, 8]))
        chunk = int(rng.choice([5, 16, 24, 32, 64, 128]))
        defer = int(rng.choice([0, 1, 2, 5, 13, 40]))
        ramp = bool(rng.integers(0, 2))
        kft = int(rng.choice([0, kf_stride]))
        wins = offline.ba_windows(n_total, kf_stride, window_kfs)
        first, count, _ = ydist.shard_frames(n_total, int(rng.integers(0, world)), world)
        assert plan_ref.cpp_chunk_plan(first, first + count, chunk, ramp, kft, wins, defer) == plan_ref.chunk_plan(first, first + count, chunk, ramp, kft, wins, defer)
        assert plan_ref.cpp_chunk_schedule(first, first + count, chunk, ramp, kft) == plan_ref.chunk_schedule(first, first + count, chunk, ramp, kft)


def test_offline_chunk_schedule_and_depth_images():
    """host logic of the offline driver (ygz_offline.cpp through ygz_offline_plan_range): the ramped chunk schedule covers the shard exactly, and the host restatement of the device's depth
    look-up (offline.depth_at) samples what the docstring says for every depth-image format"""
    for first, last, chunk in ((0, 1024, 128), (128, 256, 32), (0, 512, 128), (0, 16, 5), (8, 16, 16), (0, 600, 128), (3, 4, 128), (0, 1024, 64)):
        for ramp in (True, False):
            c = plan_ref.cpp_chunk_schedule(first, last, chunk, ramp)
            assert c[0][0] == first and c[-1][1] == last and all(a[1] == b[0] for a, b in zip(c, c[1:]))
            assert all(0 < b - a <= chunk for a, b in c)
            if ramp and chunk >= 64 and last - first >= 4 * chunk:
                assert c[0][1] - c[0][0] == chunk // 4 and c[-1][1] - c[-1][0] == chunk // 4     # short first upload, short last kernels
            # with a keyframe stride the frames behind the shard's last keyframe form the last chunk (no BA window waits for them)
            k = plan_ref.cpp_chunk_schedule(first, last, chunk, ramp, kf_stride=8)
            assert k[0][0] == first and k[-1][1] == last and all(a[1] == b[0] for a, b in zip(k, k[1:])) and all(0 < b - a <= chunk for a, b in k)
            k_last = ((last - 1) // 8) * 8
            if first <= k_last < last - 1 and c[-1][0] <= k_last:
                assert k[-1] == (k_last + 1, last) and k[:-2] == c[:-1] and k[-2] == (c[-1][0], k_last + 1)
            else:
                assert k == c
    # the plan with deferred gaps (a chunk = a tuple of frame ranges): every frame of the shard exactly once, the frames behind the last
    # keyframes of the last windows at the end -- grouped, about 45 frames per chunk --, every window complete (anchor .. last keyframe) before the
    # first deferred chunk is processed, never more than `chunk` frames per chunk
    for first, last, chunk, defer in ((0, 1024, 128, 12), (0, 1024, 128, 16), (128, 256, 32, 2), (0, 128, 32, 2), (0, 100, 32, 3), (64, 200, 32, 5), (0, 16, 5, 2),
                                      (0, 1024, 128, 13), (0, 512, 128, 13), (256, 512, 64, 13)):
        wins = offline.ba_windows(1024, 8, 8)
        plan = plan_ref.cpp_chunk_plan(first, last, chunk, True, 8, wins, defer)
        seen = np.zeros(1024, int)
        for ch in plan:
            assert len(ch) >= 1 and sum(b - a for a, b in ch) <= chunk
            assert all(first <= a < b <= last for a, b in ch) and all(x[1] < y[0] for x, y in zip(ch, ch[1:]))     # sorted, not adjacent
            for a, b in ch:
                seen[a:b] += 1
        assert np.all(seen[first:last] == 1) and seen.sum() == last - first
        inside = [w for w in wins if w[0] >= first and w[-1] < last]
        gaps = [(w[-1] + 1, min(last, w[0] + 64)) for w in inside[-defer:] if w[-1] + 1 < min(last, w[0] + 64) and w[-1] + 1 > first]
        gap_frames = sorted(f for a, b in gaps for f in range(a, b))
        n_tail = 0
        while gap_frames and sorted(f for ch in plan[len(plan) - n_tail:] for a, b in ch for f in range(a, b)) != gap_frames:
            n_tail += 1
            assert n_tail <= len(plan), (plan, gaps)
        if gap_frames:                                                              # grouped: whole gaps, about 45 frames per chunk, at the very end
            assert all(all(r in gaps for r in ch) for ch in plan[len(plan) - n_tail:]), (plan, gaps)
            assert n_tail <= max(1, int(round(len(gap_frames) / 45.0))) + (1 if chunk < 45 else 0), (plan, gaps)
        done = np.zeros(1024, bool)
        for ch in plan[:len(plan) - n_tail]:
            for a, b in ch:
                done[a:b] = True
        assert all(done[w[0]:w[-1] + 1].all() for w in inside)
        assert plan_ref.cpp_chunk_plan(first, last, chunk, True, 8, wins, 0) == [((a, b),) for a, b in plan_ref.cpp_chunk_schedule(first, last, chunk, True, 8)]
    rng = np.random.default_rng(0)
    d = rng.uniform(0.5, 6.0, (48, 64))
    d[5, 7] = 0.0
    px = np.stack([rng.integers(0, 64, 200), rng.integers(0, 48, 200)], 1).astype(np.float64)
    px[0] = [7, 5]
    full = offline.depth_at(offline.depth_image(d, 1, np.float64), px, 64, 48)
    assert np.array_equal(full, d[px[:, 1].astype(int), px[:, 0].astype(int)]) and full[0] == 0.0
    f32 = offline.depth_at(offline.depth_image(d, 1, np.float32), px, 64, 48)
    assert np.array_equal(f32, d.astype(np.float32)[px[:, 1].astype(int), px[:, 0].astype(int)].astype(np.float64))
    q = offline.depth_image(d, 4, np.uint16)                       # quarter resolution, TUM scale
    assert q.shape == (12, 16) and q.dtype == np.uint16
    got = offline.depth_at(q, px, 64, 48)
    ref = np.rint(d[4 * (px[:, 1].astype(int) // 4), 4 * (px[:, 0].astype(int) // 4)] * 5000.0) / 5000.0
    assert np.allclose(got, ref, rtol=0, atol=1e-12)
    assert np.abs(got - full).max() < 6.0                           # (a coarse prior: neighbouring samples of a random map differ)


def test_offline_chunk_plan_properties_random():
    """the C++ driver's chunk_plan (through plan_ref.cpp_chunk_plan -> ygz_offline_plan_range) on random sequences / shards / window shapes: every frame of the shard in exactly one range, ranges of a chunk sorted
    and apart, no frame twice among a chunk's frames and halo frames (they share a lane's slots), at most `chunk` frames per chunk, every window
    that ends inside the shard complete before the first deferred chunk, and shards of all ranks together cover the sequence"""
    rng = np.random.default_rng(123)
    for _ in range(300):
        kf_stride = int(rng.choice([2, 4, 8, 8, 8, 16]))
        window_kfs = int(rng.integers(2, 9))
        n_total = int(rng.integers(kf_stride * window_kfs + 1, 1500))
        world = int(rng.choice([1, 1, 2, 3, 4, 8]))
        chunk = int(rng.choice([16, 24, 32, 64, 128]))
        defer = int(rng.choice([0, 1, 2, 5, 13, 40]))
        ramp = bool(rng.integers(0, 2))
        wins = offline.ba_windows(n_total, kf_stride, window_kfs)
        covered = np.zeros(n_total, int)
        for rank in range(world):
            first, count, _halo = ydist.shard_frames(n_total, rank, world)
            if count <= 0:
                continue
            last = first + count
            plan = plan_ref.cpp_chunk_plan(first, last, chunk, ramp, kf_stride, wins, defer)
            seen = np.zeros(n_total, int)
            for ch in plan:
                assert len(ch) >= 1 and 0 < sum(b - a for a, b in ch) <= chunk, (ch, chunk)
                assert all(first <= a < b <= last for a, b in ch) and all(x[1] < y[0] for x, y in zip(ch, ch[1:])), ch
                frames = [f for a, b in ch for f in (range(a - 1, b) if a > 0 else range(a, b))]
                assert len(set(frames)) == len(frames), ch                          # halo frames included
                for a, b in ch:
                    seen[a:b] += 1
            assert np.all(seen[first:last] == 1) and seen.sum() == count
            covered += seen
            inside = [w for w in wins if w[0] >= first and w[-1] < last]
            plain = plan_ref.cpp_chunk_plan(first, last, chunk, ramp, kf_stride, wins, 0)
            if plan != plain:                                                       # some gaps are deferred: find the first chunk made of gap frames only
                in_window = np.zeros(n_total, bool)
                for w in wins:
                    in_window[w[0]:w[-1] + 1] = True
                k = len(plan)
                while k > 0 and not any(in_window[a:b].any() for a, b in plan[k - 1]):
                    k -= 1
                done = np.zeros(n_total, bool)
                for ch in plan[:k]:
                    for a, b in ch:
                        done[a:b] = True
                assert k < len(plan) and all(done[w[0]:w[-1] + 1].all() for w in inside)
        assert np.all(covered == 1)


def test_offline_chain_matches_the_interpreter_form():
    rng = np.random.default_rng(5)
    T_rel = np.stack([synth.se3_exp(rng.normal(0, 0.05, 6)) for _ in range(300)])
    a, b = offline.chain(T_rel), offline.chain_py(T_rel)
    assert np.array_equal(a[0], offline.I7) and np.allclose(a, b, rtol=0, atol=1e-13)
    assert offline.chain(T_rel[:1]).shape == (1, 7) and offline.chain(np.zeros((0, 7))).shape == (0, 7)
