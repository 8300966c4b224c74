#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03h
export TMPDIR=/tmp
run() { tag=$1; shift; timeout 300 "$@" > gpurun_out/r03h/$tag.json 2> gpurun_out/r03h/$tag.err; python - gpurun_out/r03h/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s %9.1f %s/s  ms_per_step %.3f  %s" % (sys.argv[2], d["value"], "frames", d["ms_per_step"], {k: round(v,2) for k,v in d.get("phases_ms",{}).items()} or {k: round(v,2) for k,v in d.get("stage_ms_per_batch",{}).items()}))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
STEP="python bench.py --no-cpu-baseline --no-extras"
run base $STEP
for m in 1 2 4 3 5 7; do YGZ_WAVE_PRIO=$m run prio$m $STEP; done
YGZ_KLT_PREP=e run early $STEP
for m in 1 5 7; do YGZ_KLT_PREP=e YGZ_WAVE_PRIO=$m run early_prio$m $STEP; done
YGZ_SA_LDS=832 run salds832 $STEP
YGZ_SA_LDS=768 run salds768 $STEP
YGZ_SA_LDS=832 YGZ_WAVE_PRIO=7 run salds832_prio7 $STEP
YGZ_SA_LDS=832 YGZ_WAVE_PRIO=7 YGZ_KLT_PREP=e run salds832_prio7_early $STEP
OFF="python bench.py --mode offline --frames 1024 --steps 3 --warmup 1 --no-cpu-baseline"
run off_base $OFF
YGZ_WAVE_PRIO=8 run off_prio8 $OFF
YGZ_WAVE_PRIO=9 run off_prio9 $OFF
YGZ_WAVE_PRIO=15 run off_prio15 $OFF
YGZ_WAVE_PRIO=8 run off_prio8_gray $OFF --upload gray
run off_base_gray $OFF --upload gray
