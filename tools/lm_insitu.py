#!/usr/bin/env python3
"""The resident LM on the windows an offline run built (direct-projection observations), timed ALONE after the run with HIP events: how much of
the run's BA tail is the kernel itself and how much the company it keeps.  usage: tools/lm_insitu.py [--frames 256]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, default=256); a = ap.parse_args()
R = bench.offline_render(a.frames, 0, 1)
from ygz_slam_amd import _lib, offline
W_, H_ = bench.OFF_W, bench.OFF_H
vo = offline.OfflineVO(W_, H_, a.frames, chunk=64, kf_stride=8, window_kfs=8, max_points=2000, depth_div=bench.DEPTH_DIV, depth_dtype=np.uint16,
                       depth_scale=bench.DEPTH_SCALE, lanes=3)
pin = _lib.PinnedArray((a.frames, H_, W_, 3), np.uint8); dpin = _lib.PinnedArray((a.frames, H_ // bench.DEPTH_DIV, W_ // bench.DEPTH_DIV), np.uint16)
for k, (i, b, d) in enumerate(R["rendered"]):
    pin.array[k] = b; dpin.array[k] = vo.depth_image(d)
block = lambda fr: (pin.array[fr[0]:fr[0] + len(fr)], dpin.array[fr[0]:fr[0] + len(fr)])
res = vo.run(None, None, block)
n = len(vo.mine)
print("windows", n, "sizes", [res["built"][i] for i in range(n)], "iterations/trials", [(w["lm"]["iterations"], w["lm"]["trials"]) for w in res["windows"]])
for budget in (0, 256, 64):
    for cnt in (n, 1):
        ts = []
        for rep in range(3):
            vo._ba_launch(list(range(cnt)), optimize=False)           # rebuild the graphs (the loop updates the points in place)
            vo.ba.ba_set_team_budget(budget)
            vo.ba.synchronize()
            vo.ba.timer_begin(); vo.ba.ba_optimize_resident(0, cnt, 20, want_stats=False); ts.append(vo.ba.timer_end())
        print("budget %3d: %d windows in one launch: %s ms" % (budget, cnt, " ".join("%.2f" % t for t in ts)))
if os.environ.get("YGZ_LM_DEBUG"):
    vo._ba_launch([0], optimize=False); vo.ba.ba_optimize_resident(0, 1, 20)
vo.close()
