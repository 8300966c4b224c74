#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03f
export TMPDIR=/tmp
( time timeout 900 python bench.py > gpurun_out/r03f/bench_default.json 2> gpurun_out/r03f/bench_default.err ) 2>&1 | tail -4
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03f/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("extras_error"))
print(json.dumps(d.get("offline"))[:1500])
print(json.dumps(d.get("stream"))[:1200])
print(json.dumps(d.get("cpu_baseline"))[:600])
PY
tail -n 5 gpurun_out/r03f/bench_default.err | cut -c1-300
