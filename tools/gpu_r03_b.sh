#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03b
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_offline.py -q -m gpu --no-header -rf > gpurun_out/r03b/pytest_offline.txt 2>&1
tail -15 gpurun_out/r03b/pytest_offline.txt
YGZ_OFFLINE_TRACE=1 timeout 300 python bench.py --mode offline --frames 1024 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep "offline trace"
tools/offline_timeline.sh r03b_off --mode offline --frames 1024 --steps 2 --warmup 1 --no-cpu-baseline
