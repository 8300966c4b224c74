import sys, numpy as np, time
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ygz_slam_amd import synth, _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import fixtures
from oracle import pyoracle
oracle = pyoracle.Oracle()
wins = [fixtures.ba_fixture_test_local_ba(noise=True, seed=5), synth.ba_window(6, 300, seed=5), synth.ba_window(10, 2000, seed=7),
        synth.ba_window(4, 50, seed=9), synth.ba_window(8, 700, seed=3, sort_by_point=False)]
ctx = _lib.HipContext(max_frames=1)
for i, w in enumerate(wins):
    ctx.ba_upload(i, w["poses"], w["fixed"], w["points"], w["edge_pose"], w["edge_point"], w["obs"])
t = time.perf_counter(); stats = ctx.ba_optimize_resident(0, len(wins), iterations=20); print("resident ms", (time.perf_counter() - t) * 1e3)
for i, w in enumerate(wins):
    po, pt, so = oracle.g2o_lm(w["poses"], w["fixed"], w["points"], w["edge_pose"], w["edge_point"], w["obs"], max_iterations=20)
    pg, tg = ctx.ba_get_state(i, len(w["poses"]), len(w["points"]))
    st = stats[i]
    print(i, (st.iterations, st.lm_trials), (so["iterations"], so["lm_trials"]), st.chi2_initial, so["chi2_initial"], st.chi2_final, so["chi2_final"],
          np.abs(pg - po).max(), np.abs(tg - pt).max(), st.lambda_final, so["lambda_final"])
t = time.perf_counter(); ph, th, sh = ctx.ba_optimize(wins[2]["poses"], wins[2]["fixed"], wins[2]["points"], wins[2]["edge_pose"], wins[2]["edge_point"], wins[2]["obs"]); print("host-loop ms (10x2000)", (time.perf_counter() - t) * 1e3, sh.iterations, sh.lm_trials, sh.chi2_final)

# throughput: 256 windows of 10 x 2000 in one launch
N = 256
base = [synth.ba_window(10, 2000, seed=100 + i) for i in range(8)]
for i in range(N):
    w = base[i % 8]
    ctx.ba_upload(i, w["poses"], w["fixed"], w["points"], w["edge_pose"], w["edge_point"], w["obs"])
for rep in range(2):
    for i in range(N):
        w = base[i % 8]
        ctx.ba_set_state(i, w["poses"], w["points"])
    ctx.synchronize()
    t = time.perf_counter(); st = ctx.ba_optimize_resident(0, N, iterations=20); dt = time.perf_counter() - t
    print("resident LM, %d windows x (10 x 2000): %.1f ms -> %.0f windows/s; trials %s" % (N, dt * 1e3, N / dt, [s.lm_trials for s in st[:8]]))
