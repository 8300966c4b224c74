#!/usr/bin/env python3
"""bench.py -- frames/sec of the per-frame hot path (extract + match + LK + local-BA linearise),
640x480, ~1000 ORB keypoints per frame, on N MI355X (one process per GPU).

A "step" = one pass of the whole hot path over one resident batch of B synthetic frames per GPU:
  InitFrame (BGR->gray + pyramid) -> grid FAST/ORB extraction -> cross-checked 256-bit Hamming match
  of every frame against its predecessor -> pyramidal LK of the predecessor's keypoints ->
  FindDirectProjection/Align2D of the predecessor's features -> sparse image alignment ->
  one local-BA Jacobian/JtJ build (10 keyframes x 2000 points) per frame.
Inputs are resident in HBM before the timed region; results stay in HBM.  value = frames of all ranks /
max-over-ranks time.  See DESIGN.md (measurement) for the byte accounting behind `roofline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W, H, LEVELS = 640, 480, 3


def build_inputs(batch, rank):
    from ygz_slam_amd import synth
    tex, m = synth.make_texture(1 + rank, W, H)
    poses = synth.trajectory(batch, 11 + rank, 0.25)
    poses[0] = [0, 0, 0, 1, 0, 0, 0]
    frames, depths = [], []
    for i in range(batch):
        im, d = synth.render(tex, m, poses[i], W, H, 1.0, 1000 * (rank + 1) + i)
        frames.append(synth.gray_to_bgr(im, i))
        depths.append(d)
    ba = synth.ba_window(10, 2000, seed=7)
    return np.stack(frames), poses, np.stack(depths), ba


class Pipeline:
    """Drives the C ABI for one GPU.  Everything it needs is uploaded in setup()."""

    def __init__(self, batch, device, rank, stream=None, overlap=True):
        self.overlap = overlap
        from ygz_slam_amd import _lib
        self.lib = _lib
        self.B = batch
        self.ctx = _lib.HipContext(width=W, height=H, levels=LEVELS, max_frames=batch, device=device, stream=stream)
        self.frames, self.poses, self.depths, self.ba = build_inputs(batch, rank)

    def setup(self):
        c = self.ctx
        for s in range(self.B):
            c.upload_bgr(s, self.frames[s])
        c.build_pyramid(0, self.B, from_bgr=True)
        c.detect(0, self.B)
        self.kps = [c.get_keypoints(s) for s in range(self.B)]
        self.kp_depth = [np.array([self.depths[s][int(p[1]), int(p[0])] for p in k["px"]]) for s, k in enumerate(self.kps)]
        for s in range(self.B):                               # Feature::_depth / _mappoint of the (deterministic) keypoints
            c.set_keypoint_depths(s, self.kp_depth[s], np.ones(len(self.kp_depth[s]), np.uint8))
        self.q = list(range(self.B))                          # current frame of pair i
        self.t = [(i - 1) % self.B for i in range(self.B)]    # its predecessor
        c.match_slots(self.q, self.t, 1)
        c.track_begin(self.q, self.t, self.poses[self.q], self.poses[self.t], predict=True)
        f = self.ba
        self.ba_dims = [c.ba_upload(w, f["poses"], f["fixed"], f["points"], f["edge_pose"], f["edge_point"], f["obs"])
                        for w in range(self.B)]
        c.set_overlap(self.overlap)
        c.synchronize()

    def step(self):
        """the whole hot path over the resident batch: 8 ABI calls, no host<->device copies, no syncs"""
        c = self.ctx
        # with overlap enabled the ABI forks sparse alignment, BA and the matcher onto side streams: the latency-bound sparse
        # alignment and the HBM / FP64-bound BA build then share the CUs with the VALU-bound extractor, LK and matcher.
        # (Issuing the BA build -- it depends on no image -- before the extractor was measured slower: 3.64 against 3.55 ms.)
        c.build_pyramid(0, self.B, from_bgr=True)             # A1  InitFrame
        c.detect(0, self.B)                                   # A2-A7 FeatureDetector::Detect
        c.track_reload(True)                                  # track sets from the fresh keypoints
        c.track_sparse_align()                                # L3  SparseImgAlign::run
        c.ba_linearize_resident(0, self.B)                    # B1-B5 one Jacobian/JtJ build per frame
        c.match_slots_again(1)                                # M1-M3 BFMatcher(crossCheck) vs predecessor
        c.track_direct()                                      # L1-L2 FindDirectProjection / Align2D (side stream, beside LK)
        c.track_klt()                                         # L4  Tracker::TrackKLT

    def stage_times(self, reps=3):
        """per-stage HIP-event times (ms per batch), outside the timed region"""
        c = self.ctx
        out = {}

        def t(name, fn):
            fn(); c.synchronize()
            acc = 0.0
            for _ in range(reps):
                c.timer_begin(); fn(); acc += c.timer_end()
            out[name] = acc / reps
        t("gray_pyramid", lambda: c.build_pyramid(0, self.B, from_bgr=True))
        t("detect_describe", lambda: c.detect(0, self.B))
        t("hamming_crosscheck", lambda: c.match_slots_again(1))
        t("track_load", lambda: c.track_reload(True))
        t("klt", c.track_klt)
        t("direct_projection", c.track_direct)
        t("sparse_align", c.track_sparse_align)
        t("ba_linearize", lambda: c.ba_linearize_resident(0, self.B))
        return out


def cpu_baseline(pipe, budget_s=12.0):
    """The oracle (kind 'port': our scalar restatement of the reference path, 1 core) on a bounded sample
    of the same workload: whole frames of this rank's batch until ~budget_s of CPU time is used."""
    from oracle.pyoracle import Oracle
    try:
        o = Oracle(variant="o3")        # reference flags (-O3 -march=native), built on this host
    except Exception:
        o = Oracle()
    f = pipe.ba

    last = {}

    def one_frame(i, keep=False):
        p = (i - 1) % pipe.B
        lv = o.pyramid(o.bgr2gray(pipe.frames[i]), LEVELS)
        k = o.detect(lv)
        kp_ = pipe.kps[p]
        lvp = last.get(p) if keep else None                       # sequential run: the predecessor's pyramid is kept, as in the reference
        if lvp is None:
            lvp = o.pyramid(o.bgr2gray(pipe.frames[p]), LEVELS)
        if keep:
            last.clear(); last[i] = lv
        o.bf_match(k["desc"], kp_["desc"], 1)
        pts = kp_["px"].astype(np.float32)
        o.klt_track(lvp[0], lv[0], pts, pts)
        o.find_direct_projection_n(lvp, pipe.poses[p], lv, pipe.poses[i], kp_["px"], pipe.kp_depth[p], kp_["level"], kp_["px"])
        o.sparse_align(lvp, pipe.poses[p], lv, pipe.poses[p], kp_["px"], pipe.kp_depth[p], np.ones(len(pipe.kp_depth[p]), np.uint8))
        o.ba_linearize(f["poses"], f["fixed"], f["points"], f["edge_pose"], f["edge_point"], f["obs"])

    n, t0 = 0, time.perf_counter()
    while n < pipe.B and (time.perf_counter() - t0) < budget_s:
        one_frame(n, keep=True)
        n += 1
    dt = time.perf_counter() - t0
    res = {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%d whole frames of the same batch (oracle/, gcc -O3, single thread)" % n}
    # the whole host -- the same frames handed to one thread per core (the C calls release the GIL)
    try:
        from concurrent.futures import ThreadPoolExecutor
        cores = max(1, min(len(os.sched_getaffinity(0)), 128))
        if cores > 1:
            m = int(min(pipe.B, max(cores, (n / dt) * cores * 6.0)))           # ~6 s if it scaled perfectly
            t1 = time.perf_counter()
            with ThreadPoolExecutor(cores) as ex:
                list(ex.map(one_frame, range(m)))
            dt2 = time.perf_counter() - t1
            res["all_cores"] = {"value": m / dt2, "unit": "frames/s", "cores": cores,
                                "sample": "%d whole frames, one thread per host core" % m}
    except Exception as e:       # the 1-core number is the contract; the whole-host number is extra
        res["all_cores"] = {"error": str(e)}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=512, help="frames resident per GPU per step (5.6 GB of HBM at VGA)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="run every stage on one stream")
    ap.add_argument("--probe", default="k_klt", help="kernel timed with HIP events for the roofline leg")
    ap.add_argument("--double-buffer", action="store_true",
                    help="two resident batches per GPU, consecutive steps alternate between them so that the latency-bound tail of "
                         "one step overlaps the extraction of the next (+4.5 %% frames/s; off by default because the probed kernel "
                         "then shares the GPU with the other batch and its launch duration no longer measures the kernel alone)")
    ap.add_argument("--size", default="vga", choices=["vga", "720p"],
                    help="vga = BASELINE.json's metric (640x480, the default); 720p = its configs[4] frame size (1280x720, --batch 128 per GPU)")
    a = ap.parse_args()
    global W, H
    if a.size == "720p":
        W, H = 1280, 720

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # test hook for a 1-GPU box (the N > 1 control flow without RCCL): YGZ_BENCH_ONE_DEVICE=1 puts every rank on device 0
        # and uses gloo; RCCL refuses two ranks on one device
        if os.environ.get("YGZ_BENCH_ONE_DEVICE") == "1":
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world)
    else:
        torch.cuda.set_device(local_rank)

    # one HIP stream shared by torch (RCCL broadcast) and the ABI context, so the exchange is ordered with the kernels
    # --double-buffer: two resident batches per GPU, each behind its own ABI context and streams: step k runs on batch k % 2,
    # so the next step's extraction does not wait for the latency-bound tail (sparse alignment, BA) of this one.  Every step is
    # still one full pass of the hot path over one batch of a.batch frames.
    n_buf = 2 if a.double_buffer else 1
    streams = [torch.cuda.Stream() for _ in range(n_buf)]
    torch.cuda.set_stream(streams[0])
    pipes = [Pipeline(a.batch, local_rank, rank, stream=streams[b].cuda_stream, overlap=not a.no_overlap) for b in range(n_buf)]
    for q in pipes:
        q.setup()
    pipe = pipes[0]
    n_pts = pipe.ba["points"].size
    map_buf = torch.from_numpy(np.concatenate([pipe.ba["points"].ravel(), pipe.ba["poses"].ravel()])).cuda()
    step_no = [0]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        for q in pipes:
            q.ctx.synchronize()

    def one_step():
        q = pipes[step_no[0] % n_buf]
        step_no[0] += 1
        if dist is not None:                 # the path's only exchange: map points + keyframe poses of the shared BA window
            with torch.cuda.stream(streams[pipes.index(q)]):     # the stream the batch's kernels are ordered on
                dist.broadcast(map_buf, src=0)                   # RCCL over xGMI, ~50 KB, once per BA round
            q.ctx.ba_set_state_device(0, map_buf.data_ptr() + 8 * n_pts, map_buf.data_ptr())
        q.step()

    for _ in range(a.warmup):
        one_step()
    barrier()
    probe_kernel = a.probe
    pipe.ctx.probe_begin(probe_kernel, 8 * (a.steps + 1) * 8)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one_step()
    barrier()
    dt = time.perf_counter() - t0
    probe_ms, probe_n = pipe.ctx.probe_end()
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        frames = a.batch * a.steps * world
        stages = pipe.stage_times()
        n_kp = float(np.mean([len(k["level"]) for k in pipe.kps]))
        # dominant kernel (see profiles/): algorithmic bytes per launch (DESIGN.md "Measurement") / HIP-event duration
        n_px = sum((W >> L) * (H >> L) for L in range(LEVELS))
        alg = {"k_klt": a.batch * n_kp * 5 * 2 * 23 * 23,                  # 23x23 B window, 2 images, 5 levels per point
               "k_sparse_align": a.batch * n_kp * 3 * 80.0,               # ~80 B per feature-level
               "k_fast_select": a.batch * (n_px + 8 * 3000 + 100 * 3000),
               "k_find_direct_projection": a.batch * n_kp * 220.0,
               "k_hamming_nn": a.batch * 72000.0 * 0.5}.get(probe_kernel, 0.0)
        avg_s = (probe_ms / max(probe_n, 1)) * 1e-3
        ach = alg / avg_s / 1e9 if avg_s > 0 else 0.0
        # HBM bytes per launch from the PMC passes of this same command (tools/collect_profiles.sh -> profiles/traffic.json)
        traffic = None
        tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            # the 21x21 LK launches are k_klt3 (three points per wavefront) under the library's "k_klt" timer id
            tname = {"k_klt": "k_klt3"}.get(probe_kernel, probe_kernel)
            tname = tname if tname in tj.get("kernels", {}) else probe_kernel
            if tj.get("batch") == a.batch and tname in tj.get("kernels", {}):
                traffic = tj["kernels"][tname]["hbm_bytes"]
        roofline = {"bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": traffic,
                    "kernel": {"k_klt": "k_klt3"}.get(probe_kernel, probe_kernel), "launches": probe_n, "avg_launch_us": avg_s * 1e6, "algorithmic_bytes_per_launch": alg}
        res = {"metric": "frames/sec (extract+match+LK+local-BA), %dx%d, %d ORB kpts" % (W, H, int(round(n_kp, -3)) if n_kp >= 500 else int(n_kp)),
               "value": frames / dt, "unit": "frames/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/f32/f64", "data": "synthetic",
               "config": {"workload": "2x%dx%d-pair pipeline: ORB extract + 256-bit Hamming BF cross-check + KLT 21x21x5 + "
                                      "FindDirectProjection + SparseImgAlign + local-BA 10x2000 linearise, per frame" % (W, H),
                          "frames_per_gpu_per_step": a.batch, "resident_batches_per_gpu": n_buf, "keypoints_per_frame": n_kp,
                          "parallelism": "frames sharded x%d" % world},
               "stage_ms_per_batch": stages, "roofline": roofline}
        if not a.no_cpu_baseline and world == 1:          # reported on rank 0 at N = 1 only
            res["cpu_baseline"] = cpu_baseline(pipe)
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
