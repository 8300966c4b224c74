#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the three rocprofv3 passes behind profiles/ and bench.py's roofline.traffic.
#   1. --kernel-trace --stats            -> per-kernel durations of the default bench command
#   2. --pmc FETCH_SIZE  (own pass)      -> HBM read traffic per dispatch
#   3. --pmc WRITE_SIZE  (own pass)      -> HBM write traffic per dispatch
# (FETCH_SIZE and WRITE_SIZE do not fit one pass; counters are never combined with sys/hip/hsa tracing.)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras"     # the timed region of the default command (the extra blocks follow it)
timeout 180 rocprofv3 --kernel-trace --stats -d $OUT/stats -- $CMD > $OUT/bench_stats.log 2>&1
timeout 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/bench_fetch.log 2>&1
timeout 180 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/bench_write.log 2>&1
DB=$(find $OUT/stats -name '*.db' | head -1)
[ -n "$DB" ] && python $R/tools/rocprof_summary.py $DB $OUT/kernel_stats.md > /dev/null
python $R/tools/pmc_summary.py $OUT/fetch > $OUT/pmc_fetch.md
python $R/tools/pmc_summary.py $OUT/write > $OUT/pmc_write.md
grep -h '^{' $OUT/bench_stats.log | tail -1 > $OUT/bench.json
BATCH=$(python -c "import json; print(json.load(open('$OUT/bench.json'))['config']['frames_per_gpu_per_step'])")
python $R/tools/make_traffic.py $OUT/pmc_fetch.md $OUT/pmc_write.md $OUT/traffic.json $BATCH
# keep the merge-back small: the raw traces stay on the box
rm -rf $OUT/stats $OUT/fetch $OUT/write
ls -la $OUT
