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
from ygz_slam_amd import synth


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
