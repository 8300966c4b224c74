#!/usr/bin/env python3
"""How much of k_klt3's wavefront time is the 'any of 3 live' wait?  Runs the oracle's LK on the bench workload's first frame
pairs with the iteration tap (oracle/klt.c yo_klt_iter_log) and compares, per pyramid level, the mismatch evaluations a
3-points-per-wavefront grouping executes (max over the triple) with the evaluations the points need (mean).  CPU only."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle.pyoracle import Oracle

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
o = Oracle()
frames, poses, depths, _ = bench.build_inputs(2 * n_pairs, 0)
prm = o.default_params(bench.W, bench.H, bench.LEVELS)
tot = np.zeros(5); need = np.zeros(5); srt = np.zeros(5); n_pts = 0
for k in range(n_pairs):
    g0, g1 = o.bgr2gray(frames[2 * k]), o.bgr2gray(frames[2 * k + 1])
    kp = o.detect(o.pyramid(g0, bench.LEVELS), prm)
    pts = np.stack([kp["px"], kp["py"]], 1).astype(np.float32)
    n = len(pts)
    log = np.zeros((n, 8), np.int32)
    C.c_void_p.in_dll(o.lib, "yo_klt_iter_log").value = log.ctypes.data
    o.klt_track(g0, g1, pts, pts)
    C.c_void_p.in_dll(o.lib, "yo_klt_iter_log").value = None
    it = log[:, :5].astype(np.float64)
    pad = (-n) % 3
    itp = np.concatenate([it, np.zeros((pad, 5))])
    tot += itp.reshape(-1, 3, 5).max(1).sum(0)
    need += it.sum(0) / 3.0
    key = it.sum(1)                                           # oracle-knowledge grouping: points sorted by their total count
    its = np.concatenate([it[np.argsort(key)], np.zeros((pad, 5))])
    srt += its.reshape(-1, 3, 5).max(1).sum(0)
    n_pts += n
print("points %d; per level (0..4):" % n_pts)
print(" mean evaluations per point      ", np.round(3 * need / n_pts, 2))
print(" executed per wavefront (max of 3)", np.round(3 * tot / n_pts, 2), " -> waste %.1f %%" % (100 * (1 - need.sum() / tot.sum())))
print(" with sorted grouping            ", np.round(3 * srt / n_pts, 2), " -> waste %.1f %%" % (100 * (1 - need.sum() / srt.sum())))
