"""Stage timing of the batched ba::OptimizeCurrentPoseOnly kernel (k_pose_only_ba): 256 frames x 1000 features."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ygz_slam_amd import synth, _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import fixtures

ctx = _lib.HipContext(max_frames=1)
fr = [fixtures.pose_only_fixture(n=1000, seed=100 + i, outlier_frac=0.1) for i in range(16)] * 16
off = np.concatenate([[0], np.cumsum([len(f["px"]) for f in fr])]).astype(np.int32)
px = np.concatenate([f["px"] for f in fr]); pw = np.concatenate([f["pw"] for f in fr]); en = np.stack([f["entry"] for f in fr])
for k in range(4):
    ctx.probe_begin("k_pose_only_ba", 4)
    t = time.perf_counter(); r = ctx.optimize_pose_only(off, px, pw, en); dt = time.perf_counter() - t
    ms, nl = ctx.probe_end()
    print("pose-only BA, %d frames x 1000 features: kernel %.3f ms, call incl. copies %.3f ms; inliers %s rounds %s"
          % (len(fr), ms / max(nl, 1), dt * 1e3, r[3][:3], r[4][:3]))
ctx.close()
