"""A/B aid (run on the GPU box): SparseImgAlign of the library at argv[1] -- a single-frame call on ~1000 features (the surface loop's shape) and the
batched call on two pairs -- prints the poses, the iterations per level, a hash of the outputs and the kernel time (HIP events), so that two builds
(or YGZ_SA_THREADS=256 against the default) can be compared.  python tools/ab_sparse_align.py ygz_slam_amd/libygz_hip.so"""
import hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ygz_slam_amd import _lib, synth
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
seq = synth.Sequence(3, 640, 480, seed=5, step=0.3)
ctx = _lib.HipContext(width=640, height=480, levels=3, max_frames=4)
for s in range(3):
    ctx.upload_bgr(s, seq.frame(s))
ctx.build_pyramid(0, 3, from_bgr=True); ctx.detect(0, 3)
h = hashlib.sha256()
I7 = np.array([0, 0, 0, 1.0, 0, 0, 0])
kp0 = ctx.get_keypoints(0)
d0 = seq.depth(0)[kp0["px"][:, 1].astype(int), kp0["px"][:, 0].astype(int)].astype(np.float64)
nm, Tc, its = ctx.sparse_align(0, I7, 1, I7, kp0["px"], d0, np.ones(len(d0), np.uint8))
ctx.probe_begin("k_sparse_align", 512)
for _ in range(100):
    nm, Tc, its = ctx.sparse_align(0, I7, 1, I7, kp0["px"], d0, np.ones(len(d0), np.uint8))
kms, kn = ctx.probe_end()
h.update(Tc.tobytes()); h.update(np.array(its).tobytes())
for s in range(3):
    kp = ctx.get_keypoints(s)
    d = seq.depth(s)[kp["px"][:, 1].astype(int), kp["px"][:, 0].astype(int)].astype(np.float64)
    ctx.set_keypoint_depths(s, d, np.ones(len(d), np.uint8))
ctx.track_begin([1, 2], [0, 1], np.tile(I7, (2, 1)), np.tile(I7, (2, 1)), predict=False)
ctx.track_sparse_align()
out = [ctx.track_get_pose(p) for p in range(2)]
for o in out:
    h.update(np.asarray(o[1]).tobytes()); h.update(np.array(o[2]).tobytes())
print(os.path.relpath(_lib.LIB_PATH, ROOT), "features", len(d0), "n_meas", nm, "iterations per level", its, "pose", np.array2string(Tc, precision=17),
      "| pairs:", [tuple(o[2]) for o in out], "| hash", h.hexdigest()[:16], "| single-frame kernel %.1f us x %d" % (kms * 1e3 / max(kn, 1), kn))
ctx.close()
