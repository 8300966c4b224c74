#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03v
export TMPDIR=/tmp
run() { tag=$1; shift; timeout 300 "$@" > gpurun_out/r03v/$tag.json 2> gpurun_out/r03v/$tag.err; python - gpurun_out/r03v/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s %9.1f %s/s  ms_per_step %.3f  %s" % (sys.argv[2], d["value"], "frames", d["ms_per_step"], {k: round(v,2) for k,v in d.get("phases_ms",{}).items()} or {k: round(v,2) for k,v in d.get("stage_ms_per_batch",{}).items()}))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
OFF="python bench.py --mode offline --steps 5 --warmup 2 --no-cpu-baseline"
run ramp $OFF --frames 1024
YGZ_OFF_RAMP=0 run noramp $OFF --frames 1024
run ramp_gray $OFF --frames 1024 --upload gray
YGZ_OFF_RAMP=0 run noramp_gray $OFF --frames 1024 --upload gray
run ramp_f512 $OFF --frames 512
run ramp_f256 $OFF --frames 256
run ramp_c256 $OFF --frames 1024 --batch 256
timeout 600 python -m pytest tests/test_gpu_offline.py -q -m gpu --no-header -rf -k "offline" 2>&1 | tail -2
