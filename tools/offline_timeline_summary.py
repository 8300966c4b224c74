#!/usr/bin/env python3
"""Summarise gpurun_out/<tag>_timeline.tsv (tools/offline_timeline.sh) into the markdown kept under profiles/.

  python tools/offline_timeline_summary.py gpurun_out/r04_final_timeline.tsv > profiles/r04_offline_timeline_summary.md

The table covers the LAST run of the bench (the last timed step): the frame uploads (copies of >= 16 MB), the kernels that end
after the last upload, and every resident-LM launch of that run.
"""
import sys

path = sys.argv[1]
title = sys.argv[2] if len(sys.argv) > 2 else "bench.py --mode offline --frames 1024"
K, C = [], []
for line in open(path):
    p = line.rstrip("\n").split("\t")
    if len(p) < 5:
        continue
    rec = (p[1], float(p[2]) / 1e3, float(p[3]) / 1e3, p[4])          # ms
    (K if p[0] == "K" else C).append(rec)

# a run ends with k_ba_pack (the window states packed for the exchange / download): the last run lies between the last two of them
packs = sorted(k[2] for k in K if k[0].startswith("k_ba_pack"))
t_end = packs[-1] if packs else max(k[2] for k in K)
t_begin = packs[-2] if len(packs) > 1 else min(k[1] for k in K)
up = [c for c in C if t_begin < c[1] < t_end and int(c[0].split(":")[1]) >= 2 << 20 and (int(c[0].split(":")[1]) % 2764800 == 0 or int(c[0].split(":")[1]) % 921600 == 0)]
up = [c for c in up if int(c[0].split(":")[1]) % (1280 * 720) == 0 and int(c[0].split(":")[1]) >= 1280 * 720 * 2]      # frame uploads (BGR or gray), not depth images
t0 = up[0][1]
t_last_upload = up[-1][2]
kern = [k for k in K if t0 - 0.5 <= k[1] <= t_end + 0.5]
end = max(max(k[2] for k in kern), max(c[2] for c in C if t0 <= c[1] <= t_end + 0.5))

print("# Device timeline of the last timed run of `%s` (1 GPU)\n" % title)
print("`tools/offline_timeline.sh <tag> ...` (rocprofv3 --kernel-trace --memory-copy-trace; times in ms from the first upload of the run; "
      "under the profiler), summarised by `tools/offline_timeline_summary.py`.\n")
print("## Frame uploads (one hipMemcpyAsync per chunk, chained first-in-first-out)\n")
print("| chunk | MB | start | end | GB/s |\n|---|---|---|---|---|")
for i, c in enumerate(up):
    nb = int(c[0].split(":")[1])
    print("| %d | %.1f | %.2f | %.2f | %.1f |" % (i, nb / 1e6, c[1] - t0, c[2] - t0, nb / 1e6 / max(c[2] - c[1], 1e-9)))
busy = sum(c[2] - c[1] for c in up)
print("\nLink busy %.2f ms of the %.2f ms from the first to the end of the last upload.\n" % (busy, t_last_upload - t0))
print("## Kernels that end after the last upload\n")
print("| kernel | start | end | stream |\n|---|---|---|---|")
for k in sorted((k for k in kern if k[2] > t_last_upload), key=lambda k: k[1]):
    print("| %s | %.2f | %.2f | %s |" % (k[0], k[1] - t0, k[2] - t0, k[3]))
lm = [k for k in kern if k[0].startswith("k_ba_lm_team")]
print("\nResident LM launches of the run: " + "; ".join("%.2f -> %.2f (%.2f ms)" % (k[1] - t0, k[2] - t0, k[2] - k[1]) for k in lm) + ".")
print("End of the run: %.2f ms." % (end - t0))
