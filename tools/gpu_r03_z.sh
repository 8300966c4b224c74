#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r03z
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_surface.py -q -m gpu --no-header -rf -k "ba_optimize or resident_windows or team_size" 2>&1 | tail -3
YGZ_LM_DEBUG=1 timeout 120 python tools/lm_phase_probe.py 2>&1 | tail -3
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
run g8 $OFF
run g4 $OFF --lm-group 4
run g2 $OFF --lm-group 2
run g8_gray $OFF --upload gray
run g4_gray $OFF --lm-group 4 --upload gray
