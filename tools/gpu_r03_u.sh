#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03u
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --no-header -rf > gpurun_out/r03u/pytest_gpu.txt 2>&1
tail -4 gpurun_out/r03u/pytest_gpu.txt
run() { tag=$1; shift; timeout 300 "$@" > gpurun_out/r03u/$tag.json 2> gpurun_out/r03u/$tag.err; python - gpurun_out/r03u/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s %9.1f %s/s  ms_per_step %.3f  %s" % (sys.argv[2], d["value"], "frames", d["ms_per_step"], {k: round(v,2) for k,v in d.get("phases_ms",{}).items()} or {k: round(v,2) for k,v in d.get("stage_ms_per_batch",{}).items()}))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
OFF="python bench.py --mode offline --steps 5 --warmup 2 --no-cpu-baseline"
run f1024 $OFF --frames 1024
run f1024_g4 $OFF --frames 1024 --lm-group 4
run f1024_gray $OFF --frames 1024 --upload gray
run f512 $OFF --frames 512
run f256 $OFF --frames 256
run f128 $OFF --frames 128
python tools/lm_probe.py 2>&1 | grep -v amdgpu | tail -3 | cut -c1-300
