#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03d
export TMPDIR=/tmp
run() { tag=$1; shift; timeout 300 "$@" > gpurun_out/r03d/$tag.json 2> gpurun_out/r03d/$tag.err; python - gpurun_out/r03d/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s %9.1f %s/s  ms_per_step %.2f  %s" % (sys.argv[2], d["value"], "frames", d["ms_per_step"], {k: round(v,1) for k,v in d.get("phases_ms",{}).items()} or {k: round(v,2) for k,v in d.get("stage_ms_per_batch",{}).items()}))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
OFF="python bench.py --mode offline --frames 1024 --steps 3 --warmup 1 --no-cpu-baseline"
run g4 $OFF
run g8 $OFF --lm-group 8
run g16 $OFF --lm-group 16
run g2 $OFF --lm-group 2
run g4_gray $OFF --upload gray
run g8_gray $OFF --upload gray --lm-group 8
run g16_gray $OFF --upload gray --lm-group 16
GPU_MAX_HW_QUEUES=16 run g4_q16 $OFF
GPU_MAX_HW_QUEUES=16 run g4_q16_ov $OFF --lane-overlap
GPU_MAX_HW_QUEUES=16 run g4_q16_gray $OFF --upload gray
GPU_MAX_HW_QUEUES=16 run g4_q16_ov_gray $OFF --upload gray --lane-overlap
run g4_c256 $OFF --batch 256
tools/offline_timeline.sh r03d_off --mode offline --frames 1024 --steps 2 --warmup 1 --no-cpu-baseline
tools/offline_timeline.sh r03d_off_gray --mode offline --frames 1024 --steps 2 --warmup 1 --no-cpu-baseline --upload gray
for f in gpurun_out/r03d/*.err; do echo $f; tail -n 3 $f | cut -c1-300; done | tail -40
