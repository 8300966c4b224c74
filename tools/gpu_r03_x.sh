#!/bin/bash
# step-mode schedule experiments: LK working images beside the extractor, descriptors on the matcher's stream, BA build first
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03x
export TMPDIR=/tmp
run() { tag=$1; shift; timeout 300 "$@" > gpurun_out/r03x/$tag.json 2> gpurun_out/r03x/$tag.err; python - gpurun_out/r03x/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s %9.1f frames/s  ms_per_step %.3f" % (sys.argv[2], d["value"], d["ms_per_step"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
STEP="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras"
run base $STEP
YGZ_BENCH_KLT_PREPARE=1 run prep $STEP
YGZ_DESCRIBE_ASIDE=1 run aside $STEP
YGZ_BENCH_KLT_PREPARE=1 YGZ_DESCRIBE_ASIDE=1 run prep_aside $STEP
YGZ_BENCH_KLT_PREPARE=1 YGZ_DESCRIBE_ASIDE=1 YGZ_BENCH_BA_EARLY=1 run prep_aside_baearly $STEP
YGZ_BENCH_KLT_PREPARE=1 YGZ_BENCH_BA_EARLY=1 run prep_baearly $STEP
run base2 $STEP
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu --no-header -rf -k "resident_batched" 2>&1 | tail -2
YGZ_DESCRIBE_ASIDE=1 timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_surface.py -q -m gpu --no-header -rf -x 2>&1 | tail -2
