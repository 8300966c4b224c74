#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03e
export TMPDIR=/tmp
run() { tag=$1; shift; timeout 300 "$@" > gpurun_out/r03e/$tag.json 2> gpurun_out/r03e/$tag.err; python - gpurun_out/r03e/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s %9.1f %s/s  ms_per_step %.2f  %s" % (sys.argv[2], d["value"], "frames", d["ms_per_step"], {k: round(v,1) for k,v in d.get("phases_ms",{}).items()} or {k: round(v,2) for k,v in d.get("stage_ms_per_batch",{}).items()}))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
YGZ_LM_DEBUG=1 timeout 120 python tools/lm_phase_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-900
OFF="python bench.py --mode offline --frames 1024 --steps 3 --warmup 1 --no-cpu-baseline"
run l3 $OFF
run l2 $OFF --lanes 2
run l4 $OFF --lanes 4
GPU_MAX_HW_QUEUES=8 run l3_q8 $OFF
GPU_MAX_HW_QUEUES=8 run l4_q8 $OFF --lanes 4
run l3_gray $OFF --upload gray
run l3_gray_g8 $OFF --upload gray --lm-group 8
GPU_MAX_HW_QUEUES=8 run l3_q8_gray $OFF --upload gray
run l3_c64 $OFF --batch 64 
run l3_c96 $OFF --batch 96 
timeout 1500 python -m pytest tests -q -m gpu --no-header -rf -x > gpurun_out/r03e/pytest_gpu.txt 2>&1
tail -5 gpurun_out/r03e/pytest_gpu.txt
