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

# the offline run with gray frames keeps 4 tracking lanes + the BA context = 5 HIP streams busy; the runtime maps streams onto 4 hardware
# queues unless told otherwise, and streams that share a queue serialise.  Read when the runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

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

    def __init__(self, batch, device, rank, stream=None, overlap=True, inputs=None):
        self.overlap = overlap
        from ygz_slam_amd import _lib
        self.lib = _lib
        self.B = batch
        self.ctx = _lib.HipContext(width=W, height=H, levels=LEVELS, max_frames=batch, device=device, stream=stream)
        self.frames, self.poses, self.depths, self.ba = inputs if inputs is not None else build_inputs(batch, rank)
        self.from_bgr = True
        # LK's working images (framed copies + Scharr images) right behind the pyramid, on a side stream beside the extractor: measured -1 % of the
        # step in round 5 (5.07 -> 5.005 ms) and adopted in round 6; YGZ_BENCH_EARLY_IMAGES=0 gives the round-5 order back
        self.early_images = bool(overlap) and os.environ.get("YGZ_BENCH_EARLY_IMAGES", "1") != "0"

    def setup_stream(self, upload):
        """stream mode: the batch lives in page-locked host memory and crosses PCIe every step; so do the results"""
        _lib, B, c = self.lib, self.B, self.ctx
        self.upload = upload
        self.from_bgr = upload != "gray"
        for old in ("pin", "sum_pin"):
            if hasattr(self, old):
                getattr(self, old).free()
        for v in getattr(self, "kp_pin", {}).values():
            v.free()
        if upload == "gray":                                  # the caller converts on the host side of the ABI
            self.pin = _lib.PinnedArray((B, H, W), np.uint8)
            for s in range(B):
                self.pin.array[s] = c.download_level(s, 0)
        else:
            self.pin = _lib.PinnedArray((B, H, W, 3), np.uint8)
            self.pin.array[:] = self.frames
        cells = c.cells
        self.kp_pin = {k: _lib.PinnedArray(sh, dt) for k, sh, dt in (("px", (B, cells, 2), np.float64), ("level", (B, cells), np.int32),
                       ("score", (B, cells), np.float32), ("angle", (B, cells), np.float32), ("desc", (B, cells, 32), np.uint8),
                       ("count", (B,), np.int32))}
        self.kp_out = {k: v.array for k, v in self.kp_pin.items()}
        self.sum_pin = _lib.PinnedArray((B, _lib.SUMMARY_FIELDS), np.float64)
        self.h2d_bytes = self.pin.nbytes
        self.d2h_bytes = sum(v.nbytes for v in self.kp_pin.values()) + self.sum_pin.nbytes
        self.consumed = 0.0
        self.in_flight = False

    def stream_step(self, after=None):
        """after: the pipeline whose upload was enqueued last -- this one's upload queues behind it (first-in-first-out at the full PCIe
        rate instead of sharing the link: the kernels of the earlier batch start sooner)"""
        c = self.ctx
        if self.in_flight:                                    # the step issued on this buffer two steps ago: wait, then the host reads its results
            c.synchronize()
            self.consume()
        if after is not None and after is not self:
            c.wait_mark(after.ctx)
        if self.upload == "gray":
            c.upload_gray_batch(0, self.pin.array, wait=False)
        else:
            c.upload_bgr_batch(0, self.pin.array, wait=False)
        c.mark()
        self.step()
        c.match_postfilter()
        c.get_keypoints_batch(0, self.B, out=self.kp_out, wait=False)
        c.track_get_summary(out=self.sum_pin.array, wait=False)
        self.in_flight = True

    def consume(self):
        """host-side use of a finished step: keypoint counts, a checksum of the descriptors that exist, tracked / matched totals"""
        cnt = self.kp_out["count"]
        S = self.sum_pin.array
        self.consumed += float(cnt.sum()) + float(S[:, 16].sum()) + float(S[:, 19].sum()) + float(S[:, 20].sum()) + float(S[:, 7].sum())
        self.consumed += float(self.kp_out["desc"][0, :int(cnt[0])].sum())
        self.last_counts = (int(cnt.sum()), int(S[:, 16].sum()), int(S[:, 19].sum()), int(S[:, 20].sum()))

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
        c.build_pyramid(0, self.B, from_bgr=self.from_bgr)    # A1  InitFrame
        if self.early_images:
            c.track_klt_prepare()                             # LK's Scharr images on a side stream, beside the extractor (they depend on the pyramid only)
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

    one_frame(0, keep=True)                                  # warm-up (page-in, first-touch), not timed
    n, t0, per = 0, time.perf_counter(), []
    while n < pipe.B and ((time.perf_counter() - t0) < budget_s or n < 30):      # BASELINE.md protocol: >= 30 repetitions after warm-up
        t1 = time.perf_counter()
        one_frame((n + 1) % pipe.B, keep=True)
        per.append(time.perf_counter() - t1)
        n += 1
    dt = time.perf_counter() - t0
    per = np.array(per) * 1e3
    res = {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%d whole frames of the same batch (oracle/, gcc -O3 -march=native, single thread), one warm-up frame before" % n,
           "per_frame_ms": {"median": float(np.median(per)), "p10": float(np.percentile(per, 10)), "p90": float(np.percentile(per, 90)), "reps": n},
           "what_it_is": "scalar restatement of the reference path (no SSE2 FAST / OpenCV SIMD as the reference's libraries have): "
                         "a lower bound of what the reference would do on this host, not a tuned CPU implementation"}
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


def valu_roofline(pipe, a, probe_kernel, probe_avg_s, n_kp):
    """VALU issue roofline of k_klt3 (the probed launches of the timed region, and alone) and the matrix-core roofline of the
    matcher k_hamming_f4 (probed here, outside the timed region): wave64 VALU instructions per launch (profiles/valu_counts.json: SQ_INSTS_VALU of a counter pass, scaled to this batch and
    keypoint count) / launch time / SIMDs, against the issue ceiling of the kernel's opcode mix (profiles/valu_mix.json from
    tools/valu_mix.py + the per-opcode rates measured by tools/ubench/valu_peak, profiles/r02_valu_peak.txt)."""
    base = os.path.join(ROOT, "profiles")
    try:
        counts = json.load(open(os.path.join(base, "valu_counts.json")))
        mix = json.load(open(os.path.join(base, "valu_mix.json")))
    except Exception as e:
        return {"error": "profiles/valu_counts.json / valu_mix.json not readable: %s" % e}
    import torch
    n_simd = 4 * torch.cuda.get_device_properties(0).multi_processor_count
    kp_ref = counts["keypoints_per_frame"]
    from ygz_slam_amd.srchash import kernel_source_hash
    counts_stale = counts.get("kernel_source_hash") != kernel_source_hash()
    out = {"unit": "G wave64 VALU instructions / s / SIMD", "simds": n_simd, "counts_collected_on_other_kernel_sources": counts_stale, "peak_full_rate_measured": mix["peak_full_rate"],
           "peak_half_rate_measured": mix["peak_half_rate"], "peak_guide": mix["peak_guide_2cycles_2p4GHz"], "kernels": {}}

    def entry(name, per_unit, scale, avg_s, launches_note):
        instr = per_unit * a.batch * scale
        ach = instr / n_simd / avg_s / 1e9 if avg_s > 0 else 0.0
        m = mix["kernels"][name]
        return {"valu_instr_per_launch": instr, "avg_launch_us": avg_s * 1e6, "achieved": ach, "half_rate_share_of_the_hot_loops": m["half_rate_share"],
                "issue_ceiling_of_the_mix": m["issue_ceiling_Ginstr_per_s_per_SIMD"], "frac": ach / m["issue_ceiling_Ginstr_per_s_per_SIMD"],
                "frac_of_full_rate_peak": ach / mix["peak_full_rate"], "frac_of_guide_peak": ach / mix["peak_guide_2cycles_2p4GHz"], "timed": launches_note}
    if probe_kernel == "k_klt":
        out["kernels"]["k_klt3"] = entry("k_klt3", counts["kernels"]["k_klt3"]["valu_per_unit"], n_kp / kp_ref, probe_avg_s, "HIP events inside the timed region")
    c = pipe.ctx
    c.synchronize()
    if a.profile_part == "step":
        return out
    if probe_kernel == "k_klt":                               # the same launch with the GPU to itself (in the step it shares the CUs with the side streams)
        c.probe_begin("k_klt", 64)
        for _ in range(3):
            c.track_klt()
        ms, n = c.probe_end()
        if n:
            e = out["kernels"]["k_klt3"]
            alone = e["valu_instr_per_launch"] / n_simd / (ms / n * 1e-3) / 1e9
            e["alone"] = {"avg_launch_us": ms / n * 1e3, "achieved": alone, "frac": alone / e["issue_ceiling_of_the_mix"],
                          "timed": "HIP events, 3 runs of the LK stage alone after the timed region (%d launches)" % n}
    # the matcher runs on the matrix cores (k_hamming_f4: FP4 block-scaled MFMA, K = 64 per instruction): multiply-adds of the 0 / +-1
    # expansion, 2 x 256 x |A| x |B| operations per direction
    c.probe_begin("k_hamming_nn", 64)
    for _ in range(3):
        c.match_slots_again(1)
    ms, n = c.probe_end()
    if n:
        ops = 2.0 * 256.0 * n_kp * n_kp * a.batch            # per launch: every pair of the batch, one direction
        t = ms / n * 1e-3
        form = "k_hamming_f4 (FP4 32x32x64, block scale 2^6)"
        out["mfma"] = {"kernel": form, "bound": "mfma", "unit": "TOP/s", "ops_per_launch": ops, "avg_launch_us": t * 1e6,
                       "achieved": ops / t / 1e12, "peak": 10000.0, "peak_int8": 5000.0, "peak_measured_here_fp4": 7630.0, "peak_measured_here_int8": 4392.0,
                       "frac": ops / t / 1e12 / 10000.0,
                       "frac_of_measured_fp4_rate": ops / t / 1e12 / 7630.0,
                       "note": "algorithmic ops (unpadded |A| x |B| x 256 x 2).  peak = dense FP4 MFMA of MI355X_MICROARCH.md (~10 PF; its micro-benchmark reaches "
                               "9099, tools/ubench/mfma_f4_probe 7630 with four chains per wavefront: profiles/r03_mfma_f4_probe.txt); peak_int8 = the 5 POP/s the "
                               "int8 form of rounds 1-2 was priced against (north_star's matcher target: 0.60 of it).  The kernel is VALU-bound: per tile of 32 "
                               "columns 8 MFMAs against 67 VALU instructions (32 running minima, 20 bit -> nibble, 10 for the C operand), and a SIMD hides about "
                               "four VALU instructions behind one MFMA (profiles/r02_mfma_valu_mix.txt)",
                       "timed": "HIP events, 3 runs of the matcher stage alone after the timed region (%d launches)" % n}
    return out



def kernel_table(pipe, a, n_kp):
    """`roofline_valu.step_kernels`: EVERY kernel of the step, timed with HIP events around each of its launches -- inside two extra steps (beside
    whatever shares the GPU with it there) and with its stage issued alone --, with the algorithmic bytes of SURVEY 8d, the HBM bytes of the counter
    passes (profiles/traffic.json) and the wave64 VALU instructions of the SQ pass (profiles/valu_counts.json), so that each fraction DESIGN.md
    section 4 quotes can be recomputed from this line and profiles/ alone.  Runs after the timed region."""
    import torch
    c = pipe.ctx
    base = os.path.join(ROOT, "profiles")
    try:
        counts = json.load(open(os.path.join(base, "valu_counts.json")))
    except Exception:
        counts = {"kernels": {}, "keypoints_per_frame": n_kp}
    try:
        tj = json.load(open(os.path.join(base, "traffic.json")))
    except Exception:
        tj = {"kernels": {}, "batch": 0}
    n_simd = 4 * torch.cuda.get_device_properties(0).multi_processor_count
    B = a.batch
    n_px = sum((W >> L) * (H >> L) for L in range(LEVELS))
    klt_px = 0
    w, h = W, H
    for _ in range(5):                                        # LK's five levels (maxLevel 4)
        klt_px += w * h; w, h = (w + 1) // 2, (h + 1) // 2
    edges = float(len(pipe.ba["obs"]))
    # probe id -> (name in the profiler tables, stage call that issues it, algorithmic bytes per STEP of B frames or None)
    rows = [("k_bgr2gray", "k_bgr2gray16", lambda: c.build_pyramid(0, B, from_bgr=True), B * 4.0 * W * H, "cvtColor: 3 B read + 1 B written per pixel"),
            ("k_pyr_down", "k_pyr_down", lambda: c.build_pyramid(0, B, from_bgr=True), B * 480000.0 * (W * H) / 307200.0, "SURVEY 8d: 480 000 B per VGA frame"),
            ("k_fast_select", "k_fast_select", lambda: c.detect(0, B), B * (n_px + 8 * 3000 + 100 * 3000), "every pyramid pixel once + 108 B per NMS corner"),
            ("k_compact", "k_compact", lambda: c.detect(0, B), None, None),
            ("k_describe", "k_describe", lambda: c.detect(0, B), B * n_kp * 997.0, "961 B window + 36 B written per keypoint"),
            ("k_track_load", "k_track_load", lambda: c.track_reload(True), None, None),
            ("k_sparse_align", "k_sparse_align2", c.track_sparse_align, B * n_kp * 3 * 80.0, "80 B per feature and level per linearisation (one linearisation counted)"),
            ("k_ba_pose_prep", "k_ba_pose_prep", lambda: c.ba_linearize_resident(0, B), None, None),
            ("k_ba_points", "k_ba_points", lambda: c.ba_linearize_resident(0, B), B * edges * 170.0, "170 B per edge"),
            ("k_ba_final", "k_ba_final", lambda: c.ba_linearize_resident(0, B), None, None),
            ("k_hamming_nn", "k_hamming_f4", lambda: c.match_slots_again(1), B * 72000.0 * 0.5, "72 000 B per cross-checked frame pair, two launches"),
            ("k_match_finalize", "k_match_finalize", lambda: c.match_slots_again(1), None, None),
            ("k_find_direct_projection", "k_find_direct_projection", c.track_direct, B * n_kp * 220.0, "220 B per candidate"),
            ("k_scharr", "k_scharr", c.track_klt, B * klt_px * 5.0, "1 B read + 4 B written per pixel of LK's five levels"),
            ("k_klt", "k_klt3", c.track_klt, B * n_kp * 5 * 2 * 23 * 23, "23 x 23 B window, 2 images, 5 levels per point")]
    kp_scale = n_kp / max(counts.get("keypoints_per_frame", n_kp), 1.0)
    out = {}
    for kid, name, stage, alg, alg_note in rows:
        try:
            c.synchronize()
            c.probe_begin(kid, 256)
            for _ in range(2):
                pipe.step()
            ms_in, n_in = c.probe_end()
            c.probe_begin(kid, 256)
            for _ in range(2):
                stage()
            ms_al, n_al = c.probe_end()
        except Exception as e:
            out[name] = {"error": repr(e)}
            continue
        if not n_in:
            continue
        per_step = n_in / 2.0                                  # launches per step
        t_in = ms_in / 2.0 * 1e-3                              # seconds of this kernel per step, inside the step
        t_al = ms_al / 2.0 * 1e-3 if n_al else None
        e = {"launches_per_step": per_step, "ms_in_step": t_in * 1e3, "ms_alone": None if t_al is None else t_al * 1e3}
        if alg:
            e["algorithmic_bytes_per_step"] = alg
            e["algorithmic_bytes_are"] = alg_note
            e["hbm_frac_in_step"] = alg / t_in / 8e12
            if t_al:
                e["hbm_frac_alone"] = alg / t_al / 8e12
        tk = tj.get("kernels", {}).get(name)
        if tk and tj.get("batch") == B:
            e["counter_hbm_bytes_per_step"] = tk["hbm_bytes"] * per_step
        vk = counts.get("kernels", {}).get(name)
        if vk:
            scale = kp_scale if name in ("k_describe", "k_sparse_align2", "k_hamming_f4", "k_find_direct_projection", "k_klt3") else 1.0
            instr = vk["valu_per_unit"] * B * scale * per_step      # per_unit = mean per dispatch / frames of the counter pass
            e["valu_wave64_instr_per_step"] = instr
            e["valu_frac_of_guide_peak_in_step"] = instr / n_simd / t_in / 1e9 / 1.2
            if t_al:
                e["valu_frac_of_guide_peak_alone"] = instr / n_simd / t_al / 1e9 / 1.2
        out[name] = e
    c.synchronize()
    return out


# ---------------------------------------------------------------------------------------------- surface mode: the drop-in path, one frame at a time
SURF_KF_STRIDE, SURF_LOCAL_KFS = 8, 3                         # keyframe every 8th frame; LocalMapping.local_keyframes: 3 (config/default.yaml)


def _surface_local(n_kfs):
    """indices of the local keyframes: keyframe 0 + the newest SURF_LOCAL_KFS - 1 (tests/cpp/bench_surface.cpp says why)"""
    return sorted(set([0] + list(range(max(0, n_kfs - (SURF_LOCAL_KFS - 1)), n_kfs))))


def surface_sequence(n):
    """n VGA frames of a smooth synthetic sequence + the depth images of its keyframes + ground-truth poses (frame 0 = world)"""
    from ygz_slam_amd import synth
    seq = synth.Sequence(n, W, H, seed=21, step=0.05)
    bgr = np.stack([seq.frame(i) for i in range(n)])
    kfd = np.stack([seq.depth(i).astype(np.float32) for i in range(0, n, SURF_KF_STRIDE)])
    return bgr, kfd, seq.poses.copy()


def surface_gpu(bgr, kfd, caller=0):
    """tests/cpp/bench_surface.cpp through ctypes: the reference-shaped loop over the ygz:: class surfaces (libygz_host.so), one frame at a time.
    caller 0: TrackLocalMap through the batch method Matcher::ProjectMapPoints; 1: through reference-named methods only (FindCandidates + one
    Matcher::FindDirectProjection per candidate, LocalMapping.cpp:47-120); 2: as 1 without the speculative launch; 3: as 1, every call verified."""
    import ctypes as C
    from ygz_slam_amd import _lib
    _lib.load()
    lib = C.CDLL(os.path.join(ROOT, "tests", "cpp", "libbench_surface.so"))
    n = len(bgr)
    nk = len(kfd)
    ms = np.zeros(n); T = np.zeros((n, 7)); cnt = np.zeros((n, 4), np.int32); ba = np.zeros((nk, 4)); stage = np.zeros(8); memo = np.zeros(9)
    p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    rc = lib.ygz_bench_surface2(p(bgr, C.c_uint8), p(kfd, C.c_float), n, W, H, SURF_KF_STRIDE, SURF_LOCAL_KFS, int(caller), p(ms, C.c_double), p(T, C.c_double),
                                p(cnt, C.c_int32), p(ba, C.c_double), p(stage, C.c_double), p(memo, C.c_double))
    if rc != 0:
        raise RuntimeError("ygz_bench_surface failed")
    names = ("InitFrame", "SparseImageAlignment", "ProjectMapPoints", "OptimizeCurrentPoseOnly", "Detect", "keyframe bookkeeping", "LocalBAG2O", "delete frame")
    return dict(ms=ms, T=T, counts=cnt, ba=ba, stage_ms_per_frame={k: float(v) / n for k, v in zip(names, stage)},
                memo=dict(zip(("hits", "single", "launches", "speculated", "calls", "mismatches", "find_candidates_ms", "fdp_calls_ms", "speculate_ms"),
                              [int(x) for x in memo[:6]] + [float(x) for x in memo[6:]])))


def surface_cpu(bgr, kfd, budget_s=14.0):
    """The SAME loop on the oracle (scalar restatement, one core): InitFrame -> SparseImageAlignment -> FindCandidates + ProjectMapPoints ->
    OptimizeCurrentPoseOnly -> Detect, keyframe + LocalBAG2O every 8th frame; stops after budget_s (whole frames)."""
    from oracle.pyoracle import Oracle
    from ygz_slam_amd import offline as off, synth
    try:
        o = Oracle(variant="o3")
    except Exception:
        o = Oracle()
    prm = o.default_params(W, H, LEVELS)
    fx, fy, cx, cy = synth.FX, synth.FY, synth.CX, synth.CY
    cols = -(-W // prm.cell_size)
    mp_pos = np.zeros((0, 3)); mp_nobs = np.zeros(0, np.int64)
    kfs, ref = [], None
    ms, Ts, cnts, bas = [], [], [], []
    t_start = time.perf_counter()
    for i in range(len(bgr)):
        t0 = time.perf_counter()
        lv = o.pyramid(o.bgr2gray(bgr[i]), LEVELS)
        cur = dict(lv=lv, T=off.I7.copy(), px=np.zeros((0, 2)), level=np.zeros(0, np.int32), mp=np.zeros(0, np.int64), depth=np.zeros(0), bad=np.zeros(0, bool))
        n_sa = n_proj = n_inl = 0
        if ref is not None:
            has = ref["mp"] >= 0
            n_sa = int(has.sum())
            _, T, _ = o.sparse_align(ref["lv"], ref["T"], lv, ref["T"], ref["px"], ref["depth"], has.astype(np.uint8))
            loc = [kfs[j] for j in _surface_local(len(kfs))]
            ids = [k["mp"][(k["mp"] >= 0) & ~k["bad"]] for k in loc]
            pts = np.unique(np.concatenate(ids)) if ids else np.zeros(0, np.int64)
            if len(pts):
                cp, ck, cpx, cl = [], [], [], []
                for j, k in enumerate(loc):
                    m = (k["mp"] >= 0) & ~k["bad"]
                    cp.append(np.searchsorted(pts, k["mp"][m])); ck.append(np.full(int(m.sum()), j)); cpx.append(k["px"][m]); cl.append(k["level"][m])
                cp, ck, cpx, cl = np.concatenate(cp), np.concatenate(ck), np.concatenate(cpx), np.concatenate(cl)
                order = np.lexsort((ck, cp))
                n_proj, in_view, _, match, px_m, lvl_m = o.track_local_map([k["lv"] for k in loc], [k["T"] for k in loc], lv, T, mp_pos[pts], None,
                                                                           cp[order], ck[order], cpx[order], cl[order])
                g = match >= 0
                cur.update(px=px_m[g], level=lvl_m[g].astype(np.int32), mp=pts[g].astype(np.int64))
            if len(cur["px"]):
                th = o.se3_log(T)
                pose, bad, depth, n_inl, _ = o.optimize_current_pose_only(np.concatenate([T[4:], th[3:]]), cur["px"], mp_pos[cur["mp"]])
                T = np.concatenate([o.se3_exp(np.concatenate([np.zeros(3), pose[3:]]))[:4], pose[:3]])
                cur["mp"] = np.where(bad.astype(bool), -1, cur["mp"])
                cur["depth"] = np.where(bad.astype(bool), -1.0, depth)
                cur["bad"] = bad.astype(bool)
            cur["T"] = T
        occ = np.zeros(prm.image_height // prm.cell_size * cols + cols, np.uint8)[: -(-prm.image_height // prm.cell_size) * cols]
        if len(cur["px"]):
            occ[(cur["px"][:, 1] // prm.cell_size).astype(int) * cols + (cur["px"][:, 0] // prm.cell_size).astype(int)] = 1
        kp = o.detect(lv, prm, occ if ref is not None else None)
        nd = len(kp)
        cur["px"] = np.concatenate([cur["px"], np.stack([kp["px"], kp["py"]], 1).astype(np.float64)])
        cur["level"] = np.concatenate([cur["level"], kp["level"].astype(np.int32)])
        cur["mp"] = np.concatenate([cur["mp"], np.full(nd, -1, np.int64)])
        cur["depth"] = np.concatenate([cur["depth"], np.full(nd, -1.0)])
        cur["bad"] = np.concatenate([cur["bad"], np.zeros(nd, bool)])
        if i % SURF_KF_STRIDE == 0:
            D = kfd[i // SURF_KF_STRIDE]
            old = (cur["mp"] >= 0) & ~cur["bad"]
            mp_nobs[cur["mp"][old]] += 1
            d = D[cur["px"][:, 1].astype(int), cur["px"][:, 0].astype(int)].astype(np.float64)
            new = np.nonzero((cur["mp"] < 0) & (d > 0))[0]
            pc = np.stack([(cur["px"][new, 0] - cx) * d[new] / fx, (cur["px"][new, 1] - cy) * d[new] / fy, d[new]], 1)
            cur["mp"][new] = len(mp_pos) + np.arange(len(new)); cur["depth"][new] = d[new]
            mp_pos = np.concatenate([mp_pos, off.se3_act(off.se3_inv(cur["T"]), pc)]); mp_nobs = np.concatenate([mp_nobs, np.ones(len(new), np.int64)])
            kfs.append(cur)
            loc = _surface_local(len(kfs))
            ba_rec = [0, 0, 0.0, 0.0]
            ids = np.unique(np.concatenate([kfs[j]["mp"][(kfs[j]["mp"] >= 0) & ~kfs[j]["bad"]] for j in loc]))
            ids = ids[mp_nobs[ids] >= 2]
            if len(loc) >= 2 and len(ids):
                tb = time.perf_counter()
                pose_of, poses, fixed, ep, el, ob, who = {}, [], [], [], [], [], []
                for j in loc:                                  # local keyframes first (keyframe 0 constant, BA.cpp:404), then whoever else sees the points
                    pose_of[j] = len(poses); poses.append(off.se3_log_g2o(kfs[j]["T"])); fixed.append(1 if j == 0 else 0)
                for j, k in enumerate(kfs):
                    m = (k["mp"] >= 0) & ~k["bad"] & np.isin(k["mp"], ids)
                    if not m.any():
                        continue
                    if j not in pose_of:
                        pose_of[j] = len(poses); poses.append(off.se3_log_g2o(k["T"])); fixed.append(1)
                    f = np.nonzero(m)[0]
                    ep.append(np.full(len(f), pose_of[j])); el.append(np.searchsorted(ids, k["mp"][f])); ob.append(k["px"][f]); who += [(j, f)]
                ep, el, ob = np.concatenate(ep), np.concatenate(el), np.concatenate(ob)
                order = np.lexsort((ep, el))
                P2, X2, st = o.g2o_lm(np.stack(poses), np.array(fixed, np.uint8), mp_pos[ids], ep[order], el[order], ob[order])
                lin = o.ba_linearize(P2, np.array(fixed, np.uint8), X2, ep, el, ob)             # the inlier test of BA.cpp:503-515
                e2 = (lin["err"] ** 2).sum(1)
                a = 0
                for j, f in who:
                    kfs[j]["bad"][f[e2[a:a + len(f)] > 5.991]] = True; a += len(f)
                for j in loc:
                    if j != 0:
                        kfs[j]["T"] = off.se3_exp_g2o(P2[pose_of[j]])
                mp_pos[ids] = X2
                ba_rec = [int(st.get("iterations", 0)), len(ids), float(st.get("chi2_final", 0.0)), (time.perf_counter() - tb) * 1e3]
            bas.append(ba_rec)
        Ts.append(cur["T"].copy()); cnts.append((n_sa, n_proj, n_inl, len(cur["px"])))
        ref = cur
        ms.append((time.perf_counter() - t0) * 1e3)
        if time.perf_counter() - t_start > budget_s and i >= 2 * SURF_KF_STRIDE and (i + 1) % SURF_KF_STRIDE == 0:
            break
    return dict(ms=np.array(ms), T=np.stack(Ts), counts=np.array(cnts, np.int32), ba=np.array(bas))


def surface_block(n_frames=204, cpu_budget_s=14.0, want_cpu=True):
    """The `surface` block of the driver line: frames/s of the DROP-IN path -- the reference-shaped loop, one frame at a time, through
    ygz::Frame / FeatureDetector / Matcher / ba:: (test/test_vo_track.cpp:100-113 -> VisualOdometry::AddFrame) -- and the oracle on the same loop."""
    from ygz_slam_amd import offline as off
    bgr, kfd, gt = surface_sequence(n_frames)
    surface_gpu(bgr[:3 * SURF_KF_STRIDE], kfd[:3])                      # warm-up: context, allocations, first launches
    g = surface_gpu(bgr, kfd)
    gt0 = np.stack([off.se3_mul(gt[i], off.se3_inv(gt[0])) for i in range(n_frames)])
    ms = g["ms"][1:]
    kf = np.arange(1, n_frames) % SURF_KF_STRIDE == 0
    out = {"what": "one frame at a time through the class surfaces (libygz_host.so): Frame::InitFrame -> Matcher::SparseImageAlignment -> "
                   "Matcher::ProjectMapPoints (FindCandidates + FindDirectProjection) -> ba::OptimizeCurrentPoseOnly -> FeatureDetector::Detect; "
                   "keyframe + ba::LocalBAG2O (keyframe 0 + the newest %d keyframes) every %dth frame; host clock around every frame, every surface call synchronous"
                   % (SURF_LOCAL_KFS - 1, SURF_KF_STRIDE),
           "frames": int(n_frames), "size": "%dx%d" % (W, H), "frames_per_s": float((n_frames - 1) / (ms.sum() * 1e-3)),
           "ms_per_frame": {"median": float(np.median(ms)), "p10": float(np.percentile(ms, 10)), "p90": float(np.percentile(ms, 90)),
                            "median_plain_frame": float(np.median(ms[~kf])), "median_keyframe": float(np.median(ms[kf]))},
           "host_ms_per_frame_by_surface_call": g["stage_ms_per_frame"],
           "local_ba_ms_median": float(np.median(g["ba"][g["ba"][:, 3] > 0, 3])) if (g["ba"][:, 3] > 0).any() else None,
           "mean_map_points_aligned_projected_inliers_features": [float(v) for v in g["counts"][1:].mean(0)],
           "max_abs_pose_error_vs_ground_truth": float(np.abs(g["T"] - gt0).max())}
    # the same loop with TrackLocalMap written the way the reference's UNCHANGED caller performs it -- LocalMapping::FindCandidates + one
    # Matcher::FindDirectProjection per candidate (src/Module/LocalMapping.cpp:47-120), reference-named methods only -- and, on fewer frames, what
    # those calls cost as one n = 1 launch each (the speculative launch behind FindDirectProjection switched off)
    u = surface_gpu(bgr, kfd, caller=1)
    ums = u["ms"][1:]
    n1 = min(n_frames, 5 * SURF_KF_STRIDE + 1)
    s1 = surface_gpu(bgr[:n1], kfd[:(n1 + SURF_KF_STRIDE - 1) // SURF_KF_STRIDE], caller=2)
    out["unchanged"] = {"what": "TrackLocalMap as src/Module/LocalMapping.cpp:47-120 has it: FindCandidates, then Matcher::FindDirectProjection(ref, curr, MapPoint*, px, level) "
                                "once per candidate; the first call of a frame that misses runs one speculative launch over the candidates of the keyframes in use and "
                                "the calls that follow are answered from it when their inputs are bit-equal (ygz_host.cpp: FdpMemo)",
                        "frames_per_s": float((n_frames - 1) / (ums.sum() * 1e-3)),
                        "ms_per_frame": {"median": float(np.median(ums)), "median_plain_frame": float(np.median(ums[~kf])), "median_keyframe": float(np.median(ums[kf]))},
                        "host_ms_per_frame_by_surface_call": u["stage_ms_per_frame"],
                        "find_direct_projection_calls_per_frame": u["memo"]["calls"] / float(n_frames - 1),
                        "answered_from_the_speculative_launch": u["memo"]["hits"], "n1_launches": u["memo"]["single"],
                        "speculative_launches_per_frame": u["memo"]["launches"] / float(n_frames - 1),
                        "candidates_evaluated_per_frame": u["memo"]["speculated"] / float(n_frames - 1),
                        "mean_map_points_aligned_projected_inliers_features": [float(v) for v in u["counts"][1:].mean(0)],
                        "max_abs_pose_error_vs_ground_truth": float(np.abs(u["T"] - gt0).max()),
                        "every_call_its_own_launch": {"frames": int(n1), "frames_per_s": float((n1 - 1) / (s1["ms"][1:].sum() * 1e-3)),
                                                      "ms_per_frame_median": float(np.median(s1["ms"][1:])),
                                                      "ProjectMapPoints_ms_per_frame": s1["stage_ms_per_frame"]["ProjectMapPoints"]}}
    d4 = surface_gpu(bgr[:n1], kfd[:(n1 + SURF_KF_STRIDE - 1) // SURF_KF_STRIDE], caller=4)["memo"]
    out["unchanged"]["TrackLocalMap_ms_per_frame_where"] = {
        "FindCandidates (the caller's std::map<Feature*, Vector2d>)": d4["find_candidates_ms"] / (n1 - 1),
        "inside the FindDirectProjection calls": d4["fdp_calls_ms"] / (n1 - 1),
        "of which the speculative launch (gather, launch, table)": d4["speculate_ms"] / (n1 - 1),
        "note": "a separate run of %d frames with the host clock around every call (~0.1 ms per frame of clock reads included)" % n1}
    if want_cpu:
        c = surface_cpu(bgr, kfd, cpu_budget_s)
        m = len(c["ms"])
        cms = c["ms"][1:]
        out["cpu_oracle_same_loop"] = {"frames": m, "frames_per_s": float((m - 1) / (cms.sum() * 1e-3)), "cores": 1, "kind": "port",
                                       "ms_per_frame_median": float(np.median(cms)),
                                       "mean_map_points_aligned_projected_inliers_features": [float(v) for v in c["counts"][1:].mean(0)],
                                       "max_abs_pose_error_vs_ground_truth": float(np.abs(c["T"] - gt0[:m]).max()),
                                       "max_abs_pose_difference_gpu_vs_oracle": float(np.abs(c["T"] - g["T"][:m]).max())}
        out["vs_cpu_1core"] = out["frames_per_s"] / out["cpu_oracle_same_loop"]["frames_per_s"]
        out["unchanged"]["vs_cpu_1core"] = out["unchanged"]["frames_per_s"] / out["cpu_oracle_same_loop"]["frames_per_s"]
    return out


# ---------------------------------------------------------------------------------------------- offline mode (configs[4])
_SEQ = None
OFF_W, OFF_H = 1280, 720                                      # BASELINE configs[4] frame size
DEPTH_DIV, DEPTH_SCALE = 4, 1.0 / 5000.0                      # the depth image that crosses PCIe: quarter resolution, uint16 (TUM RGB-D scale)


def _render_one(i):
    return i, _SEQ.frame(i).copy(), _SEQ.depth(i).astype(np.float32)


def offline_render(n_frames, rank, world):
    """this rank's frames of the synthetic 1280x720 sequence, rendered on the host cores with a fork-based pool -- call BEFORE the
    GPU runtime is touched"""
    global _SEQ
    from ygz_slam_amd import synth, dist as ydist
    start, count, halo = ydist.shard_frames(n_frames, rank, world)
    need = list(range(start - halo, start + count))
    import multiprocessing as mp
    _SEQ = synth.Sequence(n_frames, OFF_W, OFF_H, seed=11, step=0.02)
    workers = max(1, min(len(os.sched_getaffinity(0)) // max(1, min(world, 8)), 32))
    t_r = time.perf_counter()
    with mp.get_context("fork").Pool(workers) as pool:
        rendered = pool.map(_render_one, need, chunksize=max(1, len(need) // (4 * workers)))
    return dict(need=need, rendered=rendered, render_s=time.perf_counter() - t_r, count=count, n_frames=n_frames)


def cpu_offline_baseline(bgr, dimg, vo, budget_s=12.0):
    """the oracle's composition of the offline per-pair path (what tests/test_gpu_offline.py checks the GPU against) on a
    bounded sample of this rank's pairs, one core"""
    from oracle.pyoracle import Oracle
    try:
        o = Oracle(variant="o3")
    except Exception:
        o = Oracle()
    I7 = np.array([0, 0, 0, 1.0, 0, 0, 0])
    prm = o.default_params(OFF_W, OFF_H, LEVELS)
    n, t0, prev = 0, time.perf_counter(), None
    for k in range(len(bgr)):
        gray = bgr[k] if bgr[k].ndim == 2 else o.bgr2gray(bgr[k])
        lv = o.pyramid(gray, LEVELS)
        kp = o.detect(lv, prm)
        px = np.stack([kp["px"], kp["py"]], axis=1).astype(np.float64)
        dep = vo.depth_at(dimg[k], px)
        if prev is not None:
            plv, pkp, ppx, pdep = prev
            idx, dist_, _ = o.bf_match(kp["desc"], pkp["desc"], 1)
            o.good_match_filter(idx, dist_)
            pts = ppx.astype(np.float32)
            o.klt_track(plv[0], lv[0], pts, pts)
            _, T, _ = o.sparse_align(plv, I7, lv, I7, ppx, pdep, (pdep > 0).astype(np.uint8))
            pw, pred, cand = o.track_candidates(I7, T, ppx, pdep, OFF_W, OFF_H)
            ci = np.nonzero(cand)[0]
            ok, pxo, _ = o.find_direct_projection_n(plv, I7, lv, T, ppx[ci], pdep[ci], pkp["level"][ci], pred[ci])
            th = o.se3_log(T)
            o.optimize_current_pose_only(np.concatenate([T[4:], th[3:]]), pxo[ok], pw[ci][ok])
            n += 1
        prev = (lv, kp, px, dep)
        if time.perf_counter() - t0 > budget_s and n > 0:
            break
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d consecutive %dx%d frame pairs of the sequence (oracle/, gcc -O3, single thread; the BA round is not included)" % (n, OFF_W, OFF_H)}


def offline_run(R, rank, world, local_rank, dist, upload="bgr", steps=3, warmup=1, chunk=128, probe="k_klt", one_dev=False, cpu_baseline_s=0.0,
                overlap=False, lm_group=None, lanes=None, defer=None, bg_budget=0, bg_spread=True):
    """BASELINE configs[4] on the frames offline_render produced: one sequence sharded over the ranks (strong scaling).  A step = one
    complete offline run; the timed region holds every upload, kernel, result copy, collective and the BA round.  Returns the result
    dict on rank 0 (None elsewhere)."""
    import torch
    from ygz_slam_amd import _lib, offline
    W_, H_ = OFF_W, OFF_H
    need, n_frames, count = R["need"], R["n_frames"], R["count"]
    gray_in = upload == "gray"                                # the caller hands gray frames over (cv::cvtColor's fixed-point weights, on the host)
    chunk = min(chunk, max(32, -(-count // 4)))               # a shard is cut into >= 4 chunks: uploads, kernels and the BA windows of a rank overlap
    if lanes is None:
        # BGR frames: the run is PCIe-bound, three lanes keep the link busy (four: 57.1 against 56.0 ms of tracking per 1024 frames);
        # gray frames: kernel-bound, a fourth lane fills more of the GPU (38.6 against 40.1 ms)
        lanes = 4 if gray_in else 3
    pin = _lib.PinnedArray((len(need), H_, W_) if gray_in else (len(need), H_, W_, 3), np.uint8)
    dpin = _lib.PinnedArray((len(need), H_ // DEPTH_DIV, W_ // DEPTH_DIV), np.uint16)
    # the DRIVER is C++ (ygz_slam_amd/host/ygz_offline.cpp in libygz_host.so; collectives = RCCL called from there): this function only fills
    # page-locked buffers, calls ygz_offline_run through the binding and prints what came back
    vo = offline.OfflineVO(W_, H_, n_frames, rank=rank, world=world, device=local_rank, chunk=chunk, kf_stride=8, window_kfs=8, max_points=2000,
                           exchange_on_device=not one_dev, depth_div=DEPTH_DIV, depth_dtype=np.uint16, depth_scale=DEPTH_SCALE, overlap=overlap, lm_group=lm_group, lanes=lanes,
                           gray=gray_in, defer_gaps=defer, bg_team_budget=bg_budget, bg_team_spread=bg_spread)
    for k, (i, b, d) in enumerate(R["rendered"]):
        assert i == need[k]
        if gray_in:
            b32 = b.astype(np.int32)
            pin.array[k] = ((b32[..., 0] * 1868 + b32[..., 1] * 9617 + b32[..., 2] * 4899 + 8192) >> 14).astype(np.uint8)
        else:
            pin.array[k] = b
        dpin.array[k] = vo.depth_image(d)
    base = need[0]

    def block(frames):
        i0 = frames[0] - base
        return pin.array[i0:i0 + len(frames)], dpin.array[i0:i0 + len(frames)]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        for c in vo.lanes:
            c.synchronize()
        vo.ba.synchronize()

    res = None
    for _ in range(warmup):
        res = vo.run(None, None, block)
    barrier()
    vo.ctx.probe_begin(probe, 64 * (steps + 1))
    barrier()
    t0 = time.perf_counter()
    phases = []
    for _ in range(steps):
        t_run = time.perf_counter()
        res = vo.run(None, None, block)
        phases.append(dict(vo.timing))
    barrier()
    dt = time.perf_counter() - t0
    probe_ms, probe_n = vo.ctx.probe_end()
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    out = None
    if rank == 0:
        recs = res["records"]
        n_kp = float(np.mean([r["n_kp"] for r in recs.values()]))
        pairs = [r for r in recs.values() if "T_rel" in r]
        gt = np.stack([offline.se3_mul(_SEQ.poses[i], offline.se3_inv(_SEQ.poses[0])) for i in range(n_frames)])
        traj_err = float(np.abs(res["trajectory"] - gt).max())
        alg = {"k_klt": min(count, chunk) * n_kp * 5 * 2 * 23 * 23}.get(probe, 0.0)       # per launch = per chunk
        avg_s = (probe_ms / max(probe_n, 1)) * 1e-3
        ach = alg / avg_s / 1e9 if avg_s > 0 else 0.0
        frame_bytes = (1 if gray_in else 3) * W_ * H_ + 2 * (W_ // DEPTH_DIV) * (H_ // DEPTH_DIV)
        med = {k: float(np.median([p[k] for p in phases])) for k in phases[0]}
        out = {"metric": "frames/sec (extract+match+LK+local-BA), %dx%d offline VO, %d frames sharded over the GPUs" % (W_, H_, n_frames),
               "value": n_frames * steps / dt, "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": warmup,
               "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "u8/f32/f64", "data": "synthetic",
               "config": {"workload": "BASELINE configs[4]: %d synthetic %dx%d frames, contiguous shards with a one-frame halo, per pair ORB extract + "
                                      "BF cross-check match + good-match filter + KLT 21x21x5 + SparseImgAlign + FindCandidates/FindDirectProjection + "
                                      "pose-only BA; BA round: windows of 8 keyframes (stride 8) x <= 2000 points built on the device (observations: the map points "
                                      "projected into the keyframes and refined by FindDirectProjection), 20 LM iterations resident + the chi2 > 5.991 inlier test, "
                                      "pipelined behind the tracking chunks; map exchange + trajectory all-gather; H2D of every frame "
                                      "(+ a quarter-resolution uint16 depth image) and D2H of the results inside the timed region"
                                      % (n_frames, W_, H_),
                          "frames_total": n_frames, "frames_per_gpu": count, "chunk": chunk, "keypoints_per_frame": n_kp,
                          "parallelism": "frames sharded x%d" % world, "frames_cross_pcie_as": upload, "lanes": len(vo.lanes), "lm_launches_per_run": vo.lm_launches,
                          "host_driver": "C++ (libygz_host.so: ygz_offline_run), one ABI call per run", "exchange_backend": vo.backend,
                          "h2d_bytes_per_frame": frame_bytes, "h2d_GBps": frame_bytes * count * steps / dt / 1e9},
               "phases_ms": med, "render_s_outside_timed_region": R["render_s"],
               "result_check": {"pairs": len(pairs), "mean_pose_only_inliers": float(np.mean([r["po_inliers"] for r in pairs])),
                                "max_abs_trajectory_error_vs_ground_truth": traj_err,
                                "ba_windows": len(res["windows"]),
                                "ba_window_sizes_K_P_E": [list(res["built"][i]) for i in sorted(res["built"])[:4]],
                                "ba_chi2_initial_final": [[float(w["stats"][0]), float(w["stats"][1])] for w in res["windows"][:4]],
                                # observations by direct projection (LocalMapping.cpp:82-120); the inlier test of BA.cpp:503-515 after optimize(20)
                                "ba_observations": vo.obs_mode,
                                "ba_lm_iterations_trials": [[w["lm"]["iterations"], w["lm"]["trials"]] for w in res["windows"][:4]],
                                "ba_degenerate_windows": [i for i, w in enumerate(res["windows"]) if w["lm"]["degenerate"]],
                                "ba_edges_outliers_chi2_chi2inliers": [[w["inliers"]["edges"], w["inliers"]["outliers"], w["inliers"]["chi2"], w["inliers"]["chi2_inliers"]]
                                                                       for w in res["windows"][:4]],
                                "ba_mean_chi2_per_inlier_edge_px2": float(np.mean([w["inliers"]["chi2_inliers"] / max(1, w["inliers"]["edges"] - w["inliers"]["outliers"])
                                                                                   for w in res["windows"]])),
                                "ba_outlier_share": float(sum(w["inliers"]["outliers"] for w in res["windows"]) / max(1, sum(w["inliers"]["edges"] for w in res["windows"]))),
                                # keyframe poses relative to their window's anchor against the ground truth, before (chained tracking) and after the BA round
                                "keyframe_pose_error_vs_ground_truth": {k: float(v.mean()) for k, v in offline.window_pose_errors(res["windows"], res["trajectory"], gt).items()}},
               "roofline": {"bound": "valu", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": None,
                            "kernel": {"k_klt": "k_klt3"}.get(probe, probe), "launches": probe_n, "avg_launch_us": avg_s * 1e6,
                            "algorithmic_bytes_per_launch": alg,
                            "note": "HBM fraction of the dominant kernel (lane 0's launches); it is VALU-issue bound, see roofline_valu in the default mode"}}
        if cpu_baseline_s > 0 and world == 1:
            out["cpu_baseline"] = cpu_offline_baseline(pin.array, dpin.array, vo, cpu_baseline_s)
    vo.close()
    pin.free(); dpin.free()
    return out


def main_offline(a):
    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    R = offline_render(a.frames, rank, world)                 # before the GPU runtime is touched (fork-based pool)
    import torch
    dist = None
    one_dev = os.environ.get("YGZ_BENCH_ONE_DEVICE") == "1"
    if one_dev:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if one_dev else "nccl", rank=rank, world_size=world)
    chunk = a.batch if a.batch != 512 else 128
    out = offline_run(R, rank, world, local_rank, dist, upload=a.upload, steps=a.steps, warmup=a.warmup, chunk=chunk, probe=a.probe, one_dev=one_dev,
                      cpu_baseline_s=0.0 if a.no_cpu_baseline else 12.0, overlap=a.lane_overlap, lm_group=a.lm_group, lanes=a.lanes, defer=a.defer, bg_budget=a.bg_budget, bg_spread=not a.bg_compact)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="frames resident per GPU per step (5.6 GB of HBM at VGA)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="run every stage on one stream")
    ap.add_argument("--probe", default="k_klt", help="kernel timed with HIP events for the roofline leg")
    ap.add_argument("--double-buffer", action="store_true",
                    help="two resident batches per GPU, consecutive steps alternate between them so that the latency-bound tail of "
                         "one step overlaps the extraction of the next (+4.5 %% frames/s; off by default because the probed kernel "
                         "then shares the GPU with the other batch and its launch duration no longer measures the kernel alone)")
    ap.add_argument("--mode", default="step", choices=["step", "stream", "offline", "surface"],
                    help="step (default): the hot path over a batch resident in HBM = BASELINE.json's metric; stream: the same step with the "
                         "frames coming from page-locked host memory every step (double-buffered H2D under compute) and the keypoints + per-pair "
                         "results copied back and read by the host; offline: BASELINE configs[4], a --frames long 1280x720 sequence sharded over "
                         "the ranks (strong scaling) through ygz_slam_amd/offline.py, uploads, result copies, collectives and the BA round included")
    ap.add_argument("--frames", type=int, default=1024, help="offline mode: length of the sequence (all ranks together)")
    ap.add_argument("--offline-frames", type=int, default=1024, help="length of the configs[4] sequence measured for the `offline` block of the default line")
    ap.add_argument("--lane-overlap", action="store_true", help="offline mode: side streams inside each tracking lane (measured slower: the two lanes "
                                                               "and the BA context already fill the GPU and the hardware queues)")
    ap.add_argument("--lanes", type=int, default=None, help="offline mode: tracking contexts that take the chunks in turn (default: 3 with BGR frames, 4 with gray frames)")
    ap.add_argument("--lm-group", type=int, default=None, help="offline mode: BA windows per resident-LM launch")
    ap.add_argument("--defer", type=int, default=None, help="offline mode: keyframe-free gaps behind the last windows of a shard processed at the very end (ygz_offline_params::defer_gaps; default: the driver's rule)")
    ap.add_argument("--bg-compact", action="store_true", help="offline mode: resident-LM launches that chunks still follow keep the compact placement (one XCD per window) instead of spreading over the XCDs")
    ap.add_argument("--bg-budget", type=int, default=0, help="offline mode: workgroups a resident-LM launch may hold while tracking chunks follow (0: library default)")
    ap.add_argument("--profile-part", default=None, choices=["step", "alone"],
                    help="for rocprofv3 runs (tools/collect_r05_profiles.sh): 'step' = the warm-up and timed steps only (no stage-alone timings, no "
                         "extra blocks), so that a kernel table of the run holds IN-STEP launches only; 'alone' = every stage run by itself only")
    ap.add_argument("--no-extras", action="store_true", help="default mode: skip the `offline` and `stream` blocks (the timed region is the same either way)")
    ap.add_argument("--upload", default="bgr", choices=["bgr", "gray"], help="stream mode: what crosses PCIe per frame (3 or 1 byte per pixel)")
    ap.add_argument("--size", default=None, choices=["vga", "720p"],
                    help="vga = BASELINE.json's metric (640x480, the default); 720p = its configs[4] frame size (1280x720, --batch 128 per GPU)")
    a = ap.parse_args()
    global W, H
    if a.size is None:
        a.size = "720p" if a.mode == "offline" else "vga"
    if a.size == "720p":
        W, H = 1280, 720
    if a.mode == "offline":
        return main_offline(a)
    if a.mode == "surface":                                    # the drop-in path alone (the `surface` block of the default line)
        print(json.dumps({"surface": surface_block(n_frames=a.frames if a.frames != 1024 else 204, want_cpu=not a.no_cpu_baseline)}))
        return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # extra blocks of the default line (measured after the timed region, so that the driver's record carries them): BASELINE configs[4]
    # (the offline run, sharded over the same ranks) and, at one GPU, the transfer-inclusive stream mode
    extras = a.mode == "step" and a.size == "vga" and not a.no_extras and a.profile_part is None and os.environ.get("YGZ_BENCH_EXTRAS", "1") != "0"
    R_off = offline_render(a.offline_frames, rank, world) if extras else None      # host cores, before the GPU runtime is touched
    import torch
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

    # Step mode shards by frame: every rank runs the same step on its own resident batch -- replicas, no data-path collective (SURVEY 8e:
    # frames are independent units; the BA windows of the step are per-frame copies of one 10 x 2000 window).  The exchange the path does
    # have -- refined BA-window states from their owners to every rank, trajectory all-gather -- belongs to the offline run and is
    # measured there (the `offline` block of this line, `--mode offline`).
    # --double-buffer: two resident batches per GPU, each behind its own ABI context and streams: step k runs on batch k % 2,
    # so the next step's extraction does not wait for the latency-bound tail (sparse alignment, BA) of this one.  Every step is
    # still one full pass of the hot path over one batch of a.batch frames.
    n_buf = 2 if (a.double_buffer or a.mode == "stream") else 1
    streams = [torch.cuda.Stream() for _ in range(n_buf)]
    torch.cuda.set_stream(streams[0])
    pipes = [Pipeline(a.batch, local_rank, rank, stream=streams[b].cuda_stream, overlap=not a.no_overlap) for b in range(n_buf)]
    for q in pipes:
        q.setup()
        if a.mode == "stream":
            q.setup_stream(a.upload)
    pipe = pipes[0]
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
        if a.mode == "stream":
            q.stream_step(pipes[(step_no[0] - 2) % n_buf] if step_no[0] > 1 else None)
        else:
            q.step()

    if a.profile_part == "alone":                             # a profiler run of the stages by themselves: nothing of the step in its kernel table but one warm-up pass
        one_step(); barrier()
        st = pipe.stage_times(reps=5)
        if rank == 0:
            print(json.dumps({"profile_part": "alone", "stage_ms_per_batch": st, "frames_per_gpu_per_step": a.batch}))
        return
    for _ in range(a.warmup):
        one_step()
    barrier()
    probe_kernel = a.probe
    pipe.ctx.probe_begin(probe_kernel, 8 * (a.steps + 1) * 8)
    barrier()
    # per-step device times (median / p10 / p90): an event on the pipeline's stream at the start of every step and one
    # behind the last (after the side streams have joined); step k = the interval between events k and k + 1.  One resident batch only: with two,
    # consecutive steps run on different streams.
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)] if n_buf == 1 and a.mode == "step" else None
    t0 = time.perf_counter()
    for k in range(a.steps):
        if step_ev is not None:
            step_ev[k].record(streams[0])
        one_step()
    if step_ev is not None:
        pipe.ctx.join()
        step_ev[a.steps].record(streams[0])
    barrier()
    if a.mode == "stream":                                   # the last steps' results are read inside the timed region too
        for q in pipes:
            if q.in_flight:
                q.consume(); q.in_flight = False
    dt = time.perf_counter() - t0
    probe_ms, probe_n = pipe.ctx.probe_end()
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # who took part: one line per rank (device index, UUID, name) gathered to rank 0, so that a record of an N > 1 run shows N ranks on N
    # distinct devices and the backend the collectives ran on
    props = torch.cuda.get_device_properties(local_rank)
    me = {"rank": rank, "local_rank": local_rank, "device": props.name, "uuid": str(getattr(props, "uuid", "")), "pid": os.getpid()}
    ranks = [me]
    if dist is not None:
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
    if rank == 0:
        frames = a.batch * a.steps * world
        stages = pipe.stage_times() if a.profile_part is None else {}
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
        traffic, traffic_stale = None, None
        tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            # the 21x21 LK launches are k_klt3 (three points per wavefront) under the library's "k_klt" timer id
            tname = {"k_klt": "k_klt3"}.get(probe_kernel, probe_kernel)
            tname = tname if tname in tj.get("kernels", {}) else probe_kernel
            if tj.get("batch") == a.batch and tname in tj.get("kernels", {}):
                traffic = tj["kernels"][tname]["hbm_bytes"]
            from ygz_slam_amd.srchash import kernel_source_hash
            traffic_stale = tj.get("kernel_source_hash") != kernel_source_hash()     # the counters were collected on other kernel sources
        # The limiter of the dominant kernel is VALU issue, not HBM (its counter traffic is BELOW the algorithmic bytes: the window re-reads
        # hit L1 / L2).  `roofline` keeps the HBM view the contract asks for (achieved / peak / frac in GB/s, traffic from the PMC passes) and
        # names the real bound; `roofline_valu` prices the same launches against the measured VALU issue ceiling of the kernel's opcode mix.
        roofline = {"bound": "valu", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": traffic,
                    "kernel": {"k_klt": "k_klt3"}.get(probe_kernel, probe_kernel), "launches": probe_n, "avg_launch_us": avg_s * 1e6, "algorithmic_bytes_per_launch": alg,
                    "traffic_collected_on_other_kernel_sources": traffic_stale,
                    "note": "frac is the HBM fraction (algorithmic bytes / launch time / 8 TB/s); the kernel is VALU-issue bound, see roofline_valu"}
        roofline_valu = valu_roofline(pipe, a, probe_kernel, avg_s, n_kp)
        if a.profile_part is None and a.mode == "step" and isinstance(roofline_valu, dict):
            try:
                roofline_valu["step_kernels"] = kernel_table(pipe, a, n_kp)
                roofline_valu["step_kernels_note"] = ("every kernel of the step: HIP events around each of its launches in two extra steps (ms_in_step: beside the side "
                                                      "streams) and with its stage issued alone (ms_alone); hbm_frac = algorithmic bytes (SURVEY 8d) / time / 8 TB/s; "
                                                      "counter bytes from profiles/traffic.json, VALU instructions from profiles/valu_counts.json (guide peak: 1.2 G wave64 "
                                                      "instructions / s / SIMD).  The events serialise nothing but add ~2 us per launch")
            except Exception as e:
                roofline_valu["step_kernels"] = {"error": repr(e)}
        alone = roofline_valu.get("kernels", {}).get(roofline["kernel"], {}).get("alone") if isinstance(roofline_valu, dict) else None
        if alone:                                             # the same kernel with the GPU to itself (in the step its launch shares the CUs with three side streams)
            roofline["avg_launch_us_alone"] = alone["avg_launch_us"]
            roofline["achieved_alone"] = alg / (alone["avg_launch_us"] * 1e-6) / 1e9
            roofline["frac_alone"] = roofline["achieved_alone"] / 8000.0
            roofline["how"] = ("frac: HIP events around every launch of the kernel inside the timed steps (profiles/r05_step_kernel_stats.md holds the same launches); "
                               "frac_alone: 3 runs of the LK stage by itself after the timed region (profiles/r05_alone_kernel_stats.md)")
        step_ms = None
        if step_ev is not None:
            per = np.array([step_ev[k].elapsed_time(step_ev[k + 1]) for k in range(a.steps)])
            step_ms = {"median": float(np.median(per)), "p10": float(np.percentile(per, 10)), "p90": float(np.percentile(per, 90)),
                       "min": float(per.min()), "max": float(per.max()), "steps": int(a.steps),
                       "how": "HIP events on the pipeline's stream at the start of every step of the timed region and behind the last one (side streams joined); "
                              "rank 0's device"}
        res = {"metric": "frames/sec (extract+match+LK+local-BA), %dx%d, %d ORB kpts; frames resident in HBM (PCIe-inclusive rate: value_with_transfers)%s"
                         % (W, H, int(round(n_kp, -3)) if n_kp >= 500 else int(n_kp),
                            "" if world == 1 else "; %d independent replicas, one per GPU (weak scaling, no data-path collective) -- the sharded configs[4] run is the `offline` block (strong scaling)" % world),
               "value": frames / dt, "unit": "frames/s", "value_with_transfers": None, "step_ms": step_ms,
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/f32/f64", "data": "synthetic",
               "config": {"workload": "2x%dx%d-pair pipeline: ORB extract + 256-bit Hamming BF cross-check + KLT 21x21x5 + "
                                      "FindDirectProjection + SparseImgAlign + local-BA 10x2000 linearise, per frame" % (W, H),
                          "frames_per_gpu_per_step": a.batch, "resident_batches_per_gpu": n_buf, "keypoints_per_frame": n_kp,
                          "parallelism": "frames sharded x%d (replicas: no data-path collective in this mode; see the `offline` block)" % world},
               "ranks": {"world": world, "backend": (dist.get_backend() if dist is not None else None), "distinct_devices": len({r["uuid"] or r["local_rank"] for r in ranks}),
                         "per_rank": ranks},
               "stage_ms_per_batch": stages, "roofline": roofline, "roofline_valu": roofline_valu}
        if a.mode == "stream":
            res["metric"] += ", frames streamed from host memory"
            res["stream"] = {"upload": a.upload, "h2d_bytes_per_step": pipe.h2d_bytes, "d2h_bytes_per_step": pipe.d2h_bytes,
                             "h2d_GBps": pipe.h2d_bytes / (dt / a.steps) / 1e9, "d2h_GBps": pipe.d2h_bytes / (dt / a.steps) / 1e9,
                             "host_read_totals_last_step": list(getattr(pipes[(step_no[0] - 1) % n_buf], "last_counts", ())),
                             "note": "every step: H2D of the batch from page-locked memory, the 8-call hot path + good-match filter, D2H of all keypoint "
                                     "fields and the per-pair summary, host reads them; two batches in flight (upload of one under the kernels of the other)"}
        if not a.no_cpu_baseline and world == 1:          # reported on rank 0 at N = 1 only
            res["cpu_baseline"] = cpu_baseline(pipe)
    else:
        res = None
    if extras:
        # a collective of the extra blocks that never completes must not cost the line of the timed region: after the deadline rank 0
        # prints what it has and every rank leaves
        import threading
        done = threading.Event()

        def watchdog():
            if not done.wait(float(os.environ.get("YGZ_BENCH_EXTRAS_TIMEOUT", "300"))):
                if rank == 0:
                    res["extras_error"] = "timeout"
                    print(json.dumps(res), flush=True)
                os._exit(0)
        threading.Thread(target=watchdog, daemon=True).start()
        one_dev = os.environ.get("YGZ_BENCH_ONE_DEVICE") == "1"
        try:
            off = offline_run(R_off, rank, world, local_rank, dist, upload="bgr", steps=2, warmup=1, one_dev=one_dev)
            # (three lanes here: this process still holds the streams of the step pipeline, and the runtime deals hardware queues to
            # streams in creation order -- with a fourth lane two of them ended up on one queue: 19.6 k instead of 23.3 k standalone)
            off_g = offline_run(R_off, rank, world, local_rank, dist, upload="gray", steps=2, warmup=1, one_dev=one_dev, lanes=3)
            if rank == 0:
                keep = ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "config", "phases_ms", "result_check")
                res["offline"] = {k: off[k] for k in keep}
                res["offline"]["gray"] = {k: off_g[k] for k in ("value", "ms_per_step", "phases_ms")}
                res["offline"]["gray"]["h2d_bytes_per_frame"] = off_g["config"]["h2d_bytes_per_frame"]
        except Exception as e:                              # the step-mode line stands on its own
            if rank == 0:
                res["offline"] = {"error": repr(e)}
        R_off = None
        if world == 1:
            try:
                res["stream"] = stream_block(pipe, a, local_rank, rank)
                res["value_with_transfers"] = res["stream"]["bgr"]["value"]       # every step: H2D of the BGR frames, the same hot path, D2H of the results
            except Exception as e:
                res["stream"] = {"error": repr(e)}
            try:                                            # the drop-in path: one frame at a time through the class surfaces, and the oracle on the same loop
                res["surface"] = surface_block(want_cpu=not a.no_cpu_baseline)
            except Exception as e:
                res["surface"] = {"error": repr(e)}
        done.set()
    if rank == 0:
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def stream_block(pipe, a, local_rank, rank, steps=18, warmup=3):
    """the transfer-inclusive mode (bench.py --mode stream) measured after the default timed region: every step uploads its batch
    from page-locked memory (BGR, then gray), runs the same hot path + good-match filter and copies all keypoint fields and the
    per-pair summary back; three batches in flight, each behind its own context (the upload of the next batch is queued before the
    current one ends, so PCIe never waits for the host), the stages of a batch on one stream (side streams of three contexts would
    share the hardware queues)"""
    import torch
    n_buf, side = 3, False
    streams = [torch.cuda.Stream() for _ in range(n_buf - 1)]
    pipes = [pipe]
    for st in streams:
        q = Pipeline(a.batch, local_rank, rank, stream=st.cuda_stream, overlap=side, inputs=(pipe.frames, pipe.poses, pipe.depths, pipe.ba))
        q.setup()
        pipes.append(q)
    pipe.ctx.set_overlap(side)
    out = {"steps": steps, "warmup": warmup, "frames_per_step": a.batch, "batches_in_flight": n_buf,
           "note": "every step: H2D of the batch from page-locked memory, the 8-call hot path + good-match filter, D2H of all keypoint fields "
                   "and the per-pair summary, host reads them"}
    for upload in ("bgr", "gray"):
        for q in pipes:
            q.setup_stream(upload)
        k = 0
        fifo = True
        for _ in range(warmup):
            pipes[k % n_buf].stream_step(pipes[(k - 1) % n_buf] if fifo and k else None); k += 1
        for q in pipes:
            q.ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            pipes[k % n_buf].stream_step(pipes[(k - 1) % n_buf] if fifo else None); k += 1
        for q in pipes:
            q.ctx.synchronize()
            if q.in_flight:
                q.consume(); q.in_flight = False
        dt = time.perf_counter() - t0
        out[upload] = {"value": a.batch * steps / dt, "unit": "frames/s", "ms_per_step": dt / steps * 1e3,
                       "h2d_bytes_per_step": pipe.h2d_bytes, "d2h_bytes_per_step": pipe.d2h_bytes,
                       "h2d_GBps": pipe.h2d_bytes / (dt / steps) / 1e9, "d2h_GBps": pipe.d2h_bytes / (dt / steps) / 1e9}
    for q in pipes[1:]:
        q.ctx.close()
    pipe.ctx.set_overlap(not a.no_overlap)
    return out


if __name__ == "__main__":
    main()
