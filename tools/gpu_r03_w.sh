#!/bin/bash
# Round-3 closing batch (run through gpurun): full GPU suite, the three rocprofv3 passes of the default command, the SQ pass,
# the step timeline, the default bench line with its extra blocks, and the kernel statistics of the offline mode.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r03w
timeout 900 python -m pytest tests -q -m gpu --no-header -rf -x 2>&1 | tail -4
bash tools/collect_profiles.sh r03_v3 > gpurun_out/r03w/collect.log 2>&1; tail -3 gpurun_out/r03w/collect.log
bash tools/pmc_one_pass.sh > gpurun_out/r03w/sq.log 2>&1; tail -2 gpurun_out/r03w/sq.log | cut -c1-200
bash tools/timeline.sh > gpurun_out/r03w/timeline.log 2>&1; tail -2 gpurun_out/r03w/timeline.log | cut -c1-200
timeout 500 python bench.py > gpurun_out/r03w/bench_default.json 2> gpurun_out/r03w/bench_default.err; tail -c 1500 gpurun_out/r03w/bench_default.json
bash tools/stats_cmd.sh r03_offline1024 --mode offline --frames 1024 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -26
timeout 300 python bench.py --mode offline --frames 1024 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03w/off_f1024.json 2>/dev/null; cut -c1-400 gpurun_out/r03w/off_f1024.json
timeout 300 python bench.py --mode offline --frames 1024 --steps 5 --warmup 2 --no-cpu-baseline --upload gray > gpurun_out/r03w/off_f1024_gray.json 2>/dev/null; cut -c1-300 gpurun_out/r03w/off_f1024_gray.json
