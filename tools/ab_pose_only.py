"""A/B aid (run on the GPU box): the pose-only BA of the library at argv[1] on the frames of tests/test_gpu_parity.py::test_optimize_pose_only_batch plus
a surface-loop sized frame -- prints a hash of every output and the mean launch time, so that two builds can be compared bit for bit.
python tools/ab_pose_only.py ygz_slam_amd/prev/libygz_hip.so; python tools/ab_pose_only.py ygz_slam_amd/libygz_hip.so"""
import hashlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ygz_slam_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import fixtures
frames = [fixtures.pose_only_fixture(n=400, seed=3), fixtures.pose_only_fixture(n=1000, seed=5, outlier_frac=0.3),
          fixtures.pose_only_fixture(n=37, seed=6, outlier_frac=0.0), fixtures.pose_only_fixture(n=12, seed=4, outlier_frac=0.0),
          dict(entry=np.zeros(6), px=np.zeros((0, 2)), pw=np.zeros((0, 3))), fixtures.pose_only_fixture(n=257, seed=7),
          fixtures.pose_only_fixture(n=1003, seed=9, outlier_frac=0.02)]
frames[3]["entry"] = frames[3]["entry"] + np.array([0.5, 0.5, 0, 0, 0, 0])
frames[5]["pw"][11, 2] = -2.0
off = np.concatenate([[0], np.cumsum([len(f["px"]) for f in frames])]).astype(np.int32)
ctx = _lib.HipContext(width=640, height=480, levels=3, max_frames=1)
args = (off, np.concatenate([f["px"] for f in frames]), np.concatenate([f["pw"] for f in frames]), np.stack([f["entry"] for f in frames]))
out = ctx.optimize_pose_only(*args)
h = hashlib.sha256()
for a in out:
    h.update(np.ascontiguousarray(a).tobytes())
one = ([0, 1003], frames[6]["px"], frames[6]["pw"], frames[6]["entry"][None])
ctx.optimize_pose_only(*one)
ctx.probe_begin("k_pose_only_ba", 512)
t0 = time.perf_counter()
for _ in range(300):
    o1 = ctx.optimize_pose_only(*one)
dt = (time.perf_counter() - t0) / 300
kms, kn = ctx.probe_end()
for a in o1:
    h.update(np.ascontiguousarray(a).tobytes())
print(os.path.relpath(_lib.LIB_PATH, ROOT), h.hexdigest()[:16], "rounds", list(out[4]), "single frame of 1003 features: %.1f us per call (host clock), kernel %.1f us x %d (HIP events)" % (dt * 1e6, kms * 1e3 / max(kn, 1), kn))
ctx.close()
