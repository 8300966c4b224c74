import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ygz_slam_amd import synth, _lib
ctx = _lib.HipContext(max_frames=1)
for n in (8, 2, 1):
    wins = [synth.ba_window(8, 2000, seed=100 + i) for i in range(n)]
    for i, w in enumerate(wins):
        ctx.ba_upload(i, w["poses"], w["fixed"], w["points"], w["edge_pose"], w["edge_point"], w["obs"])
    for rep in range(2):
        for i, w in enumerate(wins): ctx.ba_set_state(i, w["poses"], w["points"])
        ctx.synchronize()
        t = time.perf_counter(); st = ctx.ba_optimize_resident(0, n, iterations=20); print("%d windows: %.2f ms" % (n, (time.perf_counter() - t) * 1e3), [(s.iterations, s.lm_trials) for s in st][:4])
