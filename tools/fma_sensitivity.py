#!/usr/bin/env python3
"""(CPU) What "bit-exact against the oracle" means against a stock GCC build of ygz-slam.

The parity oracle (and the HIP kernels) evaluate every floating-point expression as written: -ffp-contract=off.  The reference's
own build (CMakeLists.txt:15: g++ -std=c++11 -march=native -O3) leaves GCC's default -ffp-contract=fast on, so on an x86 host
with FMA every a*b+c of the reference's OWN sources (FeatureDetector.cpp:550-552 rotated BRIEF coordinates, CVUtils.cpp:271-276
Align2D sums, SparseImageAlign.cpp, the BA Jacobians ...) may be one fused operation.  This tool runs the same inputs through the
oracle compiled both ways (oracle/Makefile: libygz_oracle.so vs libygz_oracle_contract.so) and counts what differs, stage by stage.
It cannot speak for the third-party libraries (OpenCV, libfast, g2o, ceres): their binaries carry their own build flags.

usage: python tools/fma_sensitivity.py [--frames N] [--out profiles/r03_fma_sensitivity.md]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
I7 = np.array([0, 0, 0, 1.0, 0, 0, 0])


def popcount_rows(a, b):
    return np.unpackbits(np.bitwise_xor(a, b), axis=1).sum(axis=1)


def measure(n_frames=4, w=640, h=480, seed=1):
    from oracle.pyoracle import Oracle
    from ygz_slam_amd import synth
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fixtures
    A, B = Oracle(), Oracle(variant="contract")
    tex, m = synth.make_texture(seed, w, h)
    poses = synth.trajectory(n_frames, 11, 0.25)
    poses[0] = I7
    R = {k: 0 for k in ("frames", "keypoints", "keypoints_differ", "angles_differ", "angles_max_abs_diff_deg", "descriptors_differ", "descriptor_bits_flipped",
                        "descriptor_bits_total", "match_pairs", "match_idx_differ", "klt_points", "klt_status_differ", "klt_track_max_abs_diff_px",
                        "fdp_candidates", "fdp_flag_differ", "fdp_level_differ", "fdp_px_differ", "fdp_px_max_abs_diff", "sparse_align_pairs",
                        "sparse_align_iteration_counts_differ", "sparse_align_n_meas_differ", "sparse_align_pose_max_abs_diff")}
    R["angles_max_abs_diff_deg"] = R["klt_track_max_abs_diff_px"] = R["fdp_px_max_abs_diff"] = R["sparse_align_pose_max_abs_diff"] = 0.0
    prev = None
    for i in range(n_frames):
        img, depth = synth.render(tex, m, poses[i], w, h, 1.0, 1000 + i)
        out = []
        for o in (A, B):
            lv = o.pyramid(img, 3)
            out.append((lv, o.detect(lv)))
        (lvA, kA), (lvB, kB) = out
        R["frames"] += 1
        assert all(np.array_equal(x, y) for x, y in zip(lvA, lvB))        # integer pyramid
        R["keypoints"] += len(kA)
        same_set = len(kA) == len(kB) and np.array_equal(kA["px"], kB["px"]) and np.array_equal(kA["py"], kB["py"]) and np.array_equal(kA["level"], kB["level"])
        if not same_set:
            R["keypoints_differ"] += len(set(zip(kA["px"], kA["py"], kA["level"])) ^ set(zip(kB["px"], kB["py"], kB["level"])))
        else:
            da = np.abs(kA["angle"] - kB["angle"])
            R["angles_differ"] += int((da != 0).sum()); R["angles_max_abs_diff_deg"] = max(R["angles_max_abs_diff_deg"], float(da.max()))
            bits = popcount_rows(kA["desc"], kB["desc"])
            R["descriptors_differ"] += int((bits != 0).sum()); R["descriptor_bits_flipped"] += int(bits.sum()); R["descriptor_bits_total"] += 256 * len(kA)
        if prev is not None and same_set:
            plv, pk, pdepth, ppose = prev
            ia, _, _ = A.bf_match(kA["desc"], pk["desc"], 1)
            ib, _, _ = A.bf_match(kB["desc"], pk["desc"], 1)                # the matcher is integer: only its INPUT differs
            R["match_pairs"] += len(ia); R["match_idx_differ"] += int((ia != ib).sum())
            px = np.stack([pk["px"], pk["py"]], 1).astype(np.float64)
            pts = px.astype(np.float32)
            (ta, sa, _), (tb, sb, _) = A.klt_track(plv[0], lvA[0], pts, pts), B.klt_track(plv[0], lvA[0], pts, pts)
            R["klt_points"] += len(sa); R["klt_status_differ"] += int((sa != sb).sum())
            mm = (sa != 0) & (sb != 0)
            if mm.any():
                R["klt_track_max_abs_diff_px"] = max(R["klt_track_max_abs_diff_px"], float(np.abs(ta[mm] - tb[mm]).max()))
            dep = pdepth[px[:, 1].astype(int), px[:, 0].astype(int)]
            fa = A.find_direct_projection_n(plv, ppose, lvA, poses[i], px, dep, pk["level"], px)
            fb = B.find_direct_projection_n(plv, ppose, lvA, poses[i], px, dep, pk["level"], px)
            R["fdp_candidates"] += len(dep); R["fdp_flag_differ"] += int((fa[0] != fb[0]).sum()); R["fdp_level_differ"] += int((fa[2] != fb[2]).sum())
            both = fa[0] & fb[0]
            dpx = np.abs(fa[1][both] - fb[1][both])
            R["fdp_px_differ"] += int((dpx.max(1) != 0).sum()) if len(dpx) else 0
            R["fdp_px_max_abs_diff"] = max(R["fdp_px_max_abs_diff"], float(dpx.max()) if len(dpx) else 0.0)
            hm = np.ones(len(dep), np.uint8)
            na, Ta, sta = A.sparse_align(plv, ppose, lvA, ppose, px, dep, hm)
            nb, Tb, stb = B.sparse_align(plv, ppose, lvA, ppose, px, dep, hm)
            R["sparse_align_pairs"] += 1
            R["sparse_align_iteration_counts_differ"] += int(list(sta.iters_per_level)[:3] != list(stb.iters_per_level)[:3])
            R["sparse_align_n_meas_differ"] += int(na != nb)
            R["sparse_align_pose_max_abs_diff"] = max(R["sparse_align_pose_max_abs_diff"], float(np.abs(Ta - Tb).max()))
        prev = (lvA, kA, depth, poses[i])
    f = fixtures.ba_fixture_test_local_ba(noise=True)
    ra = A.ba_linearize(f["poses"], f["fixed"], f["points"], f["edge_pose"], f["edge_point"], f["obs"])
    rb = B.ba_linearize(f["poses"], f["fixed"], f["points"], f["edge_pose"], f["edge_point"], f["obs"])
    rel = lambda x, y: float(np.abs(x - y).max() / max(np.abs(x).max(), 1e-300))
    R["ba_err_max_rel_diff"] = rel(ra["err"], rb["err"]); R["ba_Hpp_max_rel_diff"] = rel(ra["Hpp"], rb["Hpp"]); R["ba_Hpl_max_rel_diff"] = rel(ra["Hpl"], rb["Hpl"])
    w10 = synth.ba_window(10, 2000, seed=7)
    ga = A.g2o_lm(w10["poses"], w10["fixed"], w10["points"], w10["edge_pose"], w10["edge_point"], w10["obs"], max_iterations=10)
    gb = B.g2o_lm(w10["poses"], w10["fixed"], w10["points"], w10["edge_pose"], w10["edge_point"], w10["obs"], max_iterations=10)
    R["g2o_lm_10x2000_iterations"] = [int(ga[2]["iterations"]), int(gb[2]["iterations"])]
    R["g2o_lm_10x2000_chi2_final_rel_diff"] = abs(ga[2]["chi2_final"] - gb[2]["chi2_final"]) / ga[2]["chi2_final"]
    R["g2o_lm_10x2000_pose_max_abs_diff"] = float(np.abs(ga[0] - gb[0]).max())
    return R


def table(R):
    pct = lambda a, b: "%d of %d (%.3f %%)" % (a, b, 100.0 * a / max(b, 1))
    rows = [("pyramid (cv::pyrDown, integer)", "identical"),
            ("keypoint set: FAST-10 + NMS (integer), grid selection on the float Shi-Tomasi score", pct(R["keypoints_differ"], R["keypoints"])),
            ("IC_Angle (float moments -> fastAtan2 polynomial): angles that differ", pct(R["angles_differ"], R["keypoints"]) + ", max %.2e deg" % R["angles_max_abs_diff_deg"]),
            ("rotated BRIEF (FeatureDetector.cpp:550-571): descriptors with a flipped bit", pct(R["descriptors_differ"], R["keypoints"])),
            ("  ... bits flipped", pct(R["descriptor_bits_flipped"], R["descriptor_bits_total"])),
            ("BFMatcher(crossCheck) on those descriptors: query rows whose match index changes", pct(R["match_idx_differ"], R["match_pairs"])),
            ("calcOpticalFlowPyrLK restatement: status bytes that differ", pct(R["klt_status_differ"], R["klt_points"]) + ", tracks max |diff| %.2e px" % R["klt_track_max_abs_diff_px"]),
            ("FindDirectProjection / Align2D (CVUtils.cpp:186-318): success flags that differ", pct(R["fdp_flag_differ"], R["fdp_candidates"])),
            ("  ... search levels that differ / refined pixels that differ", "%d / %s, max |diff| %.2e px" % (R["fdp_level_differ"], pct(R["fdp_px_differ"], R["fdp_candidates"]), R["fdp_px_max_abs_diff"])),
            ("SparseImgAlign: pairs whose Gauss-Newton iteration counts per level differ", pct(R["sparse_align_iteration_counts_differ"], R["sparse_align_pairs"]) +
             ", n_meas differs in %d, pose max |diff| %.2e" % (R["sparse_align_n_meas_differ"], R["sparse_align_pose_max_abs_diff"])),
            ("BA linearisation (test_local_ba fixture): max relative difference of err / Hpp / Hpl", "%.1e / %.1e / %.1e" % (R["ba_err_max_rel_diff"], R["ba_Hpp_max_rel_diff"], R["ba_Hpl_max_rel_diff"])),
            ("g2o LM restatement, 10 x 2000 window: iterations, final chi2 rel. diff, pose max |diff|",
             "%s, %.1e, %.1e" % (R["g2o_lm_10x2000_iterations"], R["g2o_lm_10x2000_chi2_final_rel_diff"], R["g2o_lm_10x2000_pose_max_abs_diff"]))]
    return "\n".join(["| stage | -ffp-contract=fast vs off (%d VGA frames, %d keypoints) |" % (R["frames"], R["keypoints"]), "|---|---|"] +
                     ["| %s | %s |" % r for r in rows])


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    R = measure(a.frames)
    t = table(R)
    print(t)
    if a.out:
        import subprocess
        flags = subprocess.run(["gcc", "-march=native", "-Q", "--help=target"], capture_output=True, text=True).stdout
        fma = [l.strip() for l in flags.splitlines() if l.strip().startswith(("-mfma ", "-mavx2 ", "-march="))]
        open(a.out, "w").write("# FMA-contraction sensitivity of the restated path (tools/fma_sensitivity.py, CPU)\n\n"
                               "host: gcc %s, %s\n\n%s\n\nraw: `%s`\n" % (subprocess.run(["gcc", "-dumpversion"], capture_output=True, text=True).stdout.strip(),
                                                                    "; ".join(fma), t, json.dumps(R)))
