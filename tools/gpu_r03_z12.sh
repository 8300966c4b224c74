#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r03z
run() { tag=$1; shift; timeout 300 "$@" > gpurun_out/r03z/$tag.json 2> gpurun_out/r03z/$tag.err; python - gpurun_out/r03z/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-20s %9.1f frames/s  ms %.3f  %s" % (sys.argv[2], d["value"], d["ms_per_step"], {k: round(v,2) for k,v in d.get("phases_ms",{}).items()}))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
OFF="python bench.py --mode offline --steps 5 --warmup 2 --no-cpu-baseline --frames 1024 --upload gray"
run gray_l4 $OFF --lanes 4
run gray_l5 $OFF --lanes 5
run gray_l6 $OFF --lanes 6
GPU_MAX_HW_QUEUES=16 run gray_l6_q16 $OFF --lanes 6
run gray_l4_c64 $OFF --lanes 4 --batch 64
run gray_l6_c64 $OFF --lanes 6 --batch 64
