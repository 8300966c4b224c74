#!/usr/bin/env python3
"""Run the resident step of bench.py once on a small batch and dump what its kernels produced (sparse-alignment poses and iteration
counts, LK tracks, direct projections, matches) to an .npz -- two runs under one of the library's switches (YGZ_SA_THREADS, YGZ_HAMMING_VALU, ...: tests/test_gpu_switches.py) or with two builds of
libygz_hip.so must give identical files: `tools/step_dump.py --compare a.npz b.npz`.
usage: tools/step_dump.py out.npz [--batch 16] [--size vga|720p]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("out", nargs="?")
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--size", default="vga")
ap.add_argument("--compare", nargs=2)
a = ap.parse_args()
if a.compare:
    x, y = np.load(a.compare[0]), np.load(a.compare[1])
    bad = [k for k in x.files if not np.array_equal(x[k], y[k], equal_nan=True)]
    print("compare %s %s: %s" % (a.compare[0], a.compare[1], "IDENTICAL (%d arrays)" % len(x.files) if not bad else "DIFFERENT: %s" % bad))
    sys.exit(1 if bad else 0)
import bench
if a.size == "720p":
    bench.W, bench.H = 1280, 720
p = bench.Pipeline(a.batch, 0, 0)
p.setup()
c = p.ctx
for _ in range(2):          # the second step runs with the tracker's buffers in place (fused framed copies)
    p.step(); c.synchronize()
out = {}
for i in range(a.batch):
    n_meas, T, iters = c.track_get_pose(i)
    out["sa_T_%d" % i] = np.asarray(T); out["sa_n_%d" % i] = np.asarray([n_meas] + list(iters))
    k = c.track_get_klt(i)
    for j, v in enumerate(k if isinstance(k, (tuple, list)) else [k]):
        out["klt_%d_%d" % (i, j)] = np.asarray(v)
    d = c.track_get_direct(i)
    for j, v in enumerate(d if isinstance(d, (tuple, list)) else [d]):
        out["fdp_%d_%d" % (i, j)] = np.asarray(v)
    idx, dist = c.get_matches(i)
    out["m_idx_%d" % i] = idx; out["m_dist_%d" % i] = dist
np.savez(a.out, **out)
print("wrote %s: %d arrays" % (a.out, len(out)))
