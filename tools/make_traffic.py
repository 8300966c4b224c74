#!/usr/bin/env python3
"""pmc_fetch.md + pmc_write.md (tools/pmc_summary.py tables) -> traffic.json: HBM bytes per dispatch and kernel.

rocprofv3's FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction (MI355X_MICROARCH.md, "HBM"): FETCH_SIZE tallies the
128-byte read requests of a wide streaming read at 64 B, i.e. reports half the bytes -> doubled here; WRITE_SIZE is taken
as reported (uncalibrated).  Usage: make_traffic.py pmc_fetch.md pmc_write.md out.json [frames per step of the profiled command]"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ygz_slam_amd.srchash import kernel_source_hash


def table(path, col):
    out = {}
    for ln in open(path):
        c = [x.strip() for x in ln.strip().strip("|").split("|")]
        if len(c) >= 3 and c[0] not in ("kernel", "---") and not set(c[0]) <= set("-"):
            try:
                out[c[0].replace("void ", "").split("<")[0]] = float(c[2])
            except ValueError:
                pass
    return out


f, w = table(sys.argv[1], "FETCH_SIZE"), table(sys.argv[2], "WRITE_SIZE")
res = {"note": "HBM bytes per dispatch: fetch = 2 x FETCH_SIZE KiB x 1024 (gfx950 correction), write = WRITE_SIZE KiB x 1024; "
               "bench.py --steps 5 --warmup 2", "batch": int(sys.argv[4]) if len(sys.argv) > 4 else 256,
       "kernel_source_hash": kernel_source_hash(),      # bench.py compares it with the sources it runs: a stale table is flagged, not used silently
       "kernels": {}}
# For which kernels the x 2 has been checked against a known byte count: the streaming kernels (k_bgr2gray16 reads exactly 3 B/px and writes 1 B/px
# + the framed copy: 471.9 MB of fetch per 512 VGA frames, the counter x 2 gives 471.9).  Gather kernels issue 4 ... 16-byte requests per lane that
# are tallied at their sector size, so the x 2 may overstate them: their fetch figure is an UPPER BOUND (true value in [fetch / 2, fetch]).
STREAMING = {"k_bgr2gray16", "k_pyr_down", "k_scharr", "k_klt_frame", "k_klt_pad", "k_ba_points", "k_ba_final", "k_ba_pose_prep", "k_kf_put_img", "k_kf_put", "k_compact"}
for k in sorted(set(f) | set(w)):
    fb, wb = 2.0 * f.get(k, 0.0) * 1024.0, w.get(k, 0.0) * 1024.0
    res["kernels"][k] = {"fetch_bytes": fb, "write_bytes": wb, "hbm_bytes": fb + wb,
                         "fetch_x2": "validated (streaming reads)" if k in STREAMING else "upper bound (gather kernel: true fetch in [fetch_bytes / 2, fetch_bytes])"}
json.dump(res, open(sys.argv[3], "w"), indent=1)
print(json.dumps(res["kernels"].get("k_klt3", res["kernels"].get("k_klt", {}))))
