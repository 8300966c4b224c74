"""The A/B switches of the library select a second implementation of the same result (the VALU matcher instead of the int8-MFMA one,
one workgroup per window instead of a team in the resident LM, the sparse-alignment scratch in HBM instead of LDS).  They are read
once per process, so each setting runs in its own interpreter; every output must equal the default path's bit for bit."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

SCRIPT = r'''
import os, pickle, sys
import numpy as np
sys.path.insert(0, %r)
from ygz_slam_amd import _lib, synth
out = {}
seq = synth.Sequence(3, 640, 480, seed=5, step=0.3)
ctx = _lib.HipContext(width=640, height=480, levels=3, max_frames=4)
for s in range(3):
    ctx.upload_bgr(s, seq.frame(s))
ctx.build_pyramid(0, 3, from_bgr=True); ctx.detect(0, 3)
ctx.match_slots([1, 2], [0, 1], 1)
out["match"] = [tuple(a.tobytes() for a in ctx.get_matches(p)) for p in range(2)]
for s in range(3):
    kp = ctx.get_keypoints(s)
    d = seq.depth(s)[kp["px"][:, 1].astype(int), kp["px"][:, 0].astype(int)].astype(np.float64)
    ctx.set_keypoint_depths(s, d, np.ones(len(d), np.uint8))
I7 = np.array([0, 0, 0, 1.0, 0, 0, 0])
ctx.track_begin([1, 2], [0, 1], np.tile(I7, (2, 1)), np.tile(I7, (2, 1)), predict=False)
ctx.track_sparse_align()
out["sa"] = [(ctx.track_get_pose(p)[0], ctx.track_get_pose(p)[1].tobytes(), tuple(ctx.track_get_pose(p)[2])) for p in range(2)]
w = synth.ba_window(6, 400, seed=3)
ctx.ba_upload(0, w["poses"], w["fixed"], w["points"], w["edge_pose"], w["edge_point"], w["obs"])
st = ctx.ba_optimize_resident(0, 1, iterations=10)[0]
pg, tg = ctx.ba_get_state(0, len(w["poses"]), len(w["points"]))
out["lm"] = (st.iterations, st.lm_trials, st.chi2_final, pg.tobytes(), tg.tobytes())
ctx.close()
pickle.dump(out, open(sys.argv[1], "wb"))
''' % ROOT


def _run(tmp_path, name, env):
    path = str(tmp_path / (name + ".pkl"))
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", SCRIPT, path], capture_output=True, text=True, timeout=600, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    return pickle.load(open(path, "rb"))


def test_alternate_paths_give_identical_results(tmp_path):
    ref = _run(tmp_path, "default", {})
    assert ref["lm"][2] > 0 and len(ref["match"]) == 2
    for name, env, keys in (("valu_matcher", {"YGZ_HAMMING_VALU": "1"}, ("match",)),
                            ("matcher_int8_mfma", {"YGZ_HAMMING_FORM": "1"}, ("match",)),
                            ("matcher_int8_mfma_shared_b", {"YGZ_HAMMING_WG": "1"}, ("match",)),
                            ("wave_priority", {"YGZ_WAVE_PRIO": "15"}, ("match", "sa", "lm")),
                            ("lm_single_workgroup", {"YGZ_BA_LM_TEAM": "1"}, ("lm",)),
                            ("sa_scratch_in_hbm", {"YGZ_SA_LDS": "0"}, ("sa",)),
                            ("sa_scratch_split", {"YGZ_SA_LDS": "256"}, ("sa",))):
        got = _run(tmp_path, name, env)
        for k in keys:
            assert got[k] == ref[k], (name, k)
