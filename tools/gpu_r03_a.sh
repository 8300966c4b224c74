#!/bin/bash
# round-3 GPU call A: new offline path (tests), offline bench lines, H2D rate
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03a
export TMPDIR=/tmp
timeout 120 tools/ubench/h2d_rate > gpurun_out/r03a/h2d_rate.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_offline.py -q -x -m gpu --no-header -rf > gpurun_out/r03a/pytest_offline.txt 2>&1
tail -30 gpurun_out/r03a/pytest_offline.txt
timeout 300 python bench.py --mode offline --frames 1024 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03a/offline_bgr.json 2> gpurun_out/r03a/offline_bgr.err
timeout 300 python bench.py --mode offline --frames 1024 --steps 3 --warmup 1 --no-cpu-baseline --upload gray > gpurun_out/r03a/offline_gray.json 2> gpurun_out/r03a/offline_gray.err
tail -5 gpurun_out/r03a/offline_bgr.err; cat gpurun_out/r03a/offline_bgr.json | cut -c1-1500
cat gpurun_out/r03a/offline_gray.json | cut -c1-600
cat gpurun_out/r03a/h2d_rate.txt
