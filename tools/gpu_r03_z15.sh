#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r03z
run() { tag=$1; shift; timeout 300 "$@" > gpurun_out/r03z/$tag.json 2> gpurun_out/r03z/$tag.err; python - gpurun_out/r03z/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-20s %9.1f frames/s  ms %.3f  %s %s" % (sys.argv[2], d["value"], d["ms_per_step"], {k: round(v,2) for k,v in d.get("phases_ms",{}).items()}, d["result_check"]["ba_chi2_initial_final"][0]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
OFF="python bench.py --mode offline --steps 5 --warmup 2 --no-cpu-baseline --frames 1024"
run base $OFF
YGZ_OFF_LAST_OVERLAP=1 run lastov $OFF
run base_b $OFF
YGZ_OFF_LAST_OVERLAP=1 run lastov_b $OFF
