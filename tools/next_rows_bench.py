#!/usr/bin/env python3
"""Kernel timings (HIP events) of the SURVEY 8f rows that are not part of bench.py's step: BoW transform and BoW-guided
matching, LocalMapping::FindCandidates + ProjectMapPoints, DepthFromTriangulation, pose-only BA, the resident LM loop."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ygz_slam_amd import synth
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import fixtures

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
p = bench.Pipeline(B, 0, 0)
p.setup()
c = p.ctx
c.synchronize()

def timed(name, kernel, fn, reps=3):
    fn(); c.synchronize()
    best = 1e9
    for _ in range(reps):
        c.probe_begin(kernel, 64); fn(); c.synchronize()
        ms, n = c.probe_end()
        best = min(best, ms)
    print("%-46s %-26s %8.3f ms" % (name, kernel, best))
    return best

# ---- BoW: vocabulary k = 10, L = 4 (10^4 words, DBoW3 binary format), Frame::ComputeBoW of every frame, then the two matchers
voc = fixtures.synthetic_vocabulary(k=10, L=4, seed=5)
c.vocab_load(voc)
timed("ComputeBoW, %d frames x ~970 features" % B, "k_bow_transform", lambda: c.compute_bow(0, B, 4))
s1 = list(range(B)); s2 = [(i - 1) % B for i in range(B)]
timed("SearchByBoW, %d frame pairs" % B, "k_bow_match", lambda: c.search_by_bow_slots(s1, s2, 0))
E = np.array([0.0, -0.02, 0.01, 0.02, 0.0, -1.0, -0.01, 1.0, 0.0])
timed("SearchForTriangulation, %d frame pairs" % B, "k_bow_match", lambda: c.search_by_bow_slots(s1, s2, 1, E12=np.tile(E, (B, 1))))

# ---- LocalMapping::FindCandidates + ProjectMapPoints: map points from 3 keyframes, candidates in every keyframe
rng = np.random.default_rng(1)
kf = [0, 1, 2]; cur = 3
pos, cp, ck, cx, cl = [], [], [], [], []
for k in kf:
    kp, dep = p.kps[k], p.kp_depth[k]
    Twc = None
    import oracle.pyoracle as po
    o = po.Oracle()
    Twc = o.se3_inv(p.poses[k]); R = synth.quat_to_R(Twc[:4])
    for i in range(len(dep)):
        x, y = kp["px"][i]
        pc_ = np.array([(x - synth.CX) / synth.FX * dep[i], (y - synth.CY) / synth.FY * dep[i], dep[i]])
        pw = R @ pc_ + Twc[4:]
        pidx = len(pos); pos.append(pw)
        for q in kf:
            pr, z = synth.project(p.poses[q], pw[None])
            if 20 < pr[0][0] < 620 and 20 < pr[0][1] < 460:
                cp.append(pidx); ck.append(q); cx.append(pr[0]); cl.append(int(kp["level"][i]) if q == k else 0)
pos = np.array(pos); cx = np.array(cx)
print("local map: %d points, %d candidates, %d keyframes" % (len(pos), len(cp), len(kf)))
T = [p.poses[k] for k in kf]
t0 = time.perf_counter(); r = c.track_local_map(cur, p.poses[cur], kf, T, pos, None, cp, ck, cx, cl); dt = (time.perf_counter() - t0) * 1e3
timed("FindCandidates + ProjectMapPoints (call incl. copies %.2f ms, %d matched)" % (dt, r[0]), "k_lmap_match",
      lambda: c.track_local_map(cur, p.poses[cur], kf, T, pos, None, cp, ck, cx, cl))

# ---- DepthFromTriangulation: 10^6 ray pairs
n = 1000000
f1 = rng.normal(size=(n, 3)); f1[:, 2] = np.abs(f1[:, 2]) + 1; f1 /= np.linalg.norm(f1, axis=1, keepdims=True)
f2 = f1 + rng.normal(scale=0.01, size=(n, 3)); f2 /= np.linalg.norm(f2, axis=1, keepdims=True)
Tsr = synth.se3_exp([0.1, 0.02, -0.01, 0.01, -0.02, 0.005])
timed("DepthFromTriangulation, 10^6 ray pairs", "k_depth_from_triangulation", lambda: c.depth_from_triangulation(Tsr, f1, f2))
c.close()
