#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03i
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_offline.py -q -m gpu --no-header -rf -k "klt or depth_filter" > gpurun_out/r03i/pytest.txt 2>&1
tail -25 gpurun_out/r03i/pytest.txt | cut -c1-400
tools/offline_timeline.sh r03i_off --mode offline --frames 1024 --steps 2 --warmup 1 --no-cpu-baseline
tools/offline_timeline.sh r03i_off_gray --mode offline --frames 1024 --steps 2 --warmup 1 --no-cpu-baseline --upload gray
