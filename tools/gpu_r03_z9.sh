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
OFF="python bench.py --mode offline --steps 5 --warmup 2 --no-cpu-baseline --frames 1024"
run a0 $OFF
YGZ_OFF_AHEAD=1 run a1 $OFF
YGZ_OFF_AHEAD=2 run a2 $OFF
YGZ_OFF_AHEAD=3 run a3 $OFF
YGZ_OFF_AHEAD=2 run a2_l2 $OFF --lanes 2
YGZ_OFF_AHEAD=2 run a2_gray $OFF --upload gray
YGZ_OFF_AHEAD=2 run a2_gray_l3 $OFF --upload gray --lanes 3
timeout 600 python -m pytest tests/test_gpu_offline.py -q -m gpu --no-header -rf -k "offline" 2>&1 | tail -2
YGZ_OFF_AHEAD=2 timeout 600 python -m pytest tests/test_gpu_offline.py -q -m gpu --no-header -rf -k "offline" 2>&1 | tail -2
