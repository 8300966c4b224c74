#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03c
export TMPDIR=/tmp
run() { tag=$1; shift; timeout 300 "$@" > gpurun_out/r03c/$tag.json 2> gpurun_out/r03c/$tag.err; python - gpurun_out/r03c/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s %9.1f %s/s  ms_per_step %.2f  %s" % (sys.argv[2], d["value"], "frames", d["ms_per_step"], {k: round(v,1) for k,v in d.get("phases_ms",{}).items()} or {k: round(v,2) for k,v in d.get("stage_ms_per_batch",{}).items()}))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
OFF="python bench.py --mode offline --frames 1024 --steps 3 --warmup 1 --no-cpu-baseline"
run off_default $OFF
run off_noov $OFF --no-overlap
GPU_MAX_HW_QUEUES=16 run off_q16 $OFF
GPU_MAX_HW_QUEUES=16 run off_q16_noov $OFF --no-overlap
GPU_MAX_HW_QUEUES=16 run off_q16_c64 $OFF --batch 64
GPU_MAX_HW_QUEUES=16 run off_q16_gray $OFF --upload gray
GPU_MAX_HW_QUEUES=16 run off_q16_gray_c64 $OFF --upload gray --batch 64
run off_noov_gray $OFF --no-overlap --upload gray
STEP="python bench.py --no-cpu-baseline --no-extras"
run step_default $STEP
GPU_MAX_HW_QUEUES=8 run step_q8 $STEP
GPU_MAX_HW_QUEUES=16 run step_q16 $STEP
GPU_MAX_HW_QUEUES=16 run step_q16_db $STEP --double-buffer
tail -3 gpurun_out/r03c/*.err | cut -c1-300 | tail -30
