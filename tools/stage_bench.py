#!/usr/bin/env python3
"""Run ONE stage of the resident pipeline repeatedly (for rocprofv3 --pmc passes and A/B timing).
usage: tools/stage_bench.py <stage> [--batch 256] [--reps 5]
stages: pyramid detect match load klt direct sparse ba all"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

ap = argparse.ArgumentParser()
ap.add_argument("stage")
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--probe", default="", help="comma-separated kernel names: HIP-event time of each kernel's launches in one run of the stage")
a = ap.parse_args()
p = bench.Pipeline(a.batch, 0, 0)
p.setup()
c = p.ctx
fn = {"pyramid": lambda: c.build_pyramid(0, p.B, from_bgr=True), "detect": lambda: c.detect(0, p.B),
      "match": lambda: c.match_slots_again(1), "load": lambda: c.track_reload(True), "klt": c.track_klt,
      "direct": c.track_direct, "sparse": c.track_sparse_align, "ba": lambda: c.ba_linearize_resident(0, p.B),
      "all": p.step}[a.stage]
p.step(); c.synchronize()
fn(); c.synchronize()
ts = []
for _ in range(a.reps):
    c.timer_begin(); fn(); ts.append(c.timer_end())
print("%s batch %d: %s ms  (min %.3f)" % (a.stage, a.batch, " ".join("%.3f" % t for t in ts), min(ts)))
for name in [x for x in a.probe.split(",") if x]:
    c.probe_begin(name); fn(); c.synchronize()
    ms, n = c.probe_end()
    print("  %s: %d launches, %.3f ms total" % (name, n, ms))
