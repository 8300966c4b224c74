#!/usr/bin/env python3
"""gpurun_out/sq_counters.md (tools/pmc_one_pass.sh: one rocprofv3 --pmc pass of SQ counters over `tools/stage_bench.py ba --reps 2`)
-> profiles/valu_counts.json: wave64 VALU instructions per dispatch and per unit (frame pair) for every kernel of the step.
bench.py's roofline_valu block scales these to its batch.  Usage: make_valu_counts.py sq_counters.md out.json batch keypoints_per_frame"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ygz_slam_amd.srchash import kernel_source_hash

src, dst, batch, n_kp = sys.argv[1], sys.argv[2], int(sys.argv[3]), float(sys.argv[4])
rows, head = {}, None
for ln in open(src):
    c = [x.strip() for x in ln.strip().strip("|").split("|")]
    if len(c) < 3 or set(c[0]) <= set("-"):
        continue
    if c[0] == "kernel":
        head = c
        continue
    rows[c[0].replace("void ", "").split("<")[0]] = {h: float(v) for h, v in zip(head[1:], c[1:])}
out = {"note": "SQ_INSTS_VALU (wave64 instructions, mean per dispatch) from one rocprofv3 --pmc pass over tools/stage_bench.py ba --reps 2 at batch %d "
               "(tools/pmc_one_pass.sh, raw table: profiles/r04_sq_counters_raw.md); per_unit = per frame pair of ~%.0f keypoints; cycles = "
               "GRBM_GUI_ACTIVE / 8 XCDs.  The matcher is dispatched twice per cross-checked match (query->train, train->query)." % (batch, n_kp),
       "batch": batch, "keypoints_per_frame": n_kp, "kernel_source_hash": kernel_source_hash(), "kernels": {}}
for k, v in sorted(rows.items()):
    if not k.startswith("k_"):
        continue
    out["kernels"][k] = {"valu_per_dispatch": v["SQ_INSTS_VALU"], "valu_per_unit": v["SQ_INSTS_VALU"] / batch,
                         "cycles_per_dispatch": v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: out["kernels"][k]["valu_per_unit"] for k in ("k_klt3", "k_hamming_mfma", "k_hamming_f4") if k in out["kernels"]}))
