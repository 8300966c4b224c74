#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03k
export TMPDIR=/tmp
run() { tag=$1; shift; timeout 300 "$@" > gpurun_out/r03k/$tag.json 2> gpurun_out/r03k/$tag.err; python - gpurun_out/r03k/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s %9.1f %s/s  ms_per_step %.3f  %s" % (sys.argv[2], d["value"], "frames", d["ms_per_step"], {k: round(v,2) for k,v in d.get("phases_ms",{}).items()} or {k: round(v,2) for k,v in d.get("stage_ms_per_batch",{}).items()}))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
OFF="python bench.py --mode offline --frames 1024 --steps 3 --warmup 1 --no-cpu-baseline"
run fifo $OFF
YGZ_OFF_FIFO=0 run nofifo $OFF
run fifo_l2 $OFF --lanes 2
run fifo_l4 $OFF --lanes 4
run fifo_c64 $OFF --batch 64
run fifo_gray $OFF --upload gray
YGZ_OFF_FIFO=0 run nofifo_gray $OFF --upload gray
run fifo_gray_l2 $OFF --upload gray --lanes 2
# what a rank of an 8-GPU run does (128 frames, chunks of 32): one GPU, for the Amdahl table
run f128 python bench.py --mode offline --frames 128 --steps 5 --warmup 2 --no-cpu-baseline
run f256 python bench.py --mode offline --frames 256 --steps 5 --warmup 2 --no-cpu-baseline
run f512 python bench.py --mode offline --frames 512 --steps 4 --warmup 1 --no-cpu-baseline
tools/offline_timeline.sh r03k_off --mode offline --frames 1024 --steps 2 --warmup 1 --no-cpu-baseline
timeout 600 python -m pytest tests/test_gpu_offline.py -q -m gpu --no-header -rf -k "offline" 2>&1 | tail -3
