"""Multi-GPU host logic: one process per GPU, frames sharded in contiguous chunks, no data-path collective.

The hot path partitions by frame (extraction) and by consecutive frame pair (matching / LK / alignment), so
rank r owns frames [start, start+count) plus a one-frame halo (its first pair needs the predecessor).  The only
exchange is the local-BA window (map points + keyframe poses, ~50 KB), broadcast from its owner once per BA
round over RCCL (backend "nccl" on ROCm; "gloo" in the CPU tests) and an all-gather of per-shard trajectories
at the end of a run."""
import numpy as np


def shard_frames(n_total, rank, world):
    """Contiguous shard of rank: (first_frame, n_frames, halo) -- halo = 1 if a predecessor frame of another
    shard must be processed too (every shard but the first)."""
    base, rem = divmod(n_total, world)
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count, (1 if start > 0 and count > 0 else 0)


def shard_pairs(n_total, rank, world):
    """(cur, ref) global frame indices of the consecutive pairs this rank tracks: every frame but frame 0 is the
    `cur` of exactly one pair, owned by the rank that owns that frame."""
    start, count, _ = shard_frames(n_total, rank, world)
    return [(i, i - 1) for i in range(max(start, 1), start + count)]


def broadcast_map(points, poses, src=0, device=None):
    """RCCL/gloo broadcast of the BA window state (points [P,3], poses [K,6]) from `src`; returns numpy copies."""
    import torch
    import torch.distributed as dist
    P, K = points.shape[0], poses.shape[0]
    buf = torch.from_numpy(np.concatenate([np.ascontiguousarray(points, np.float64).ravel(),
                                           np.ascontiguousarray(poses, np.float64).ravel()]))
    if device is not None:
        buf = buf.to(device)
    dist.broadcast(buf, src=src)
    out = buf.cpu().numpy()
    return out[:P * 3].reshape(P, 3).copy(), out[P * 3:].reshape(K, 6).copy()


def gather_trajectories(local_poses, n_total, rank, world, device=None):
    """all-gather of per-shard poses [count,7] into the full [n_total,7] trajectory (shards may be ragged)."""
    import torch
    import torch.distributed as dist
    counts = [shard_frames(n_total, r, world)[1] for r in range(world)]
    mx = max(counts)
    pad = np.zeros((mx, 7))
    pad[:len(local_poses)] = local_poses
    t = torch.from_numpy(pad)
    if device is not None:
        t = t.to(device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return np.concatenate([o.cpu().numpy()[:c] for o, c in zip(outs, counts)], 0)


def all_gather_rows(local, counts, pg=None, via_host=False):
    """One all-gather of ragged row blocks: rank r contributes counts[r] rows (`local`, a torch tensor [counts[rank], ...] on any
    device); returns a tensor [world, max(counts), ...] on local's device whose block r holds rank r's rows (the tail of a short
    block is padding).  Fixed shapes, so this is a single collective (RCCL all_gather_into_tensor on device buffers; with
    via_host=True -- gloo, the CPU tests and the two-ranks-on-one-GPU tests -- the same call on host copies)."""
    import torch
    import torch.distributed as dist
    world = len(counts)
    mx = max(max(counts), 1)
    lbuf = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device="cpu" if via_host else local.device)
    if local.shape[0]:
        lbuf[:local.shape[0]] = local
    gbuf = torch.empty((world,) + tuple(lbuf.shape), dtype=local.dtype, device=lbuf.device)
    if via_host:
        dist.all_gather(list(gbuf.unbind(0)), lbuf, group=pg)
        return gbuf.to(local.device)
    dist.all_gather_into_tensor(gbuf, lbuf, group=pg)
    return gbuf
