#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 --kernel-trace --stats over tools/stage_bench.py <stage> [args]; prints the per-kernel duration table.
R=${GRAFT_REPO_ROOT:-/root/repo}
STAGE=$1; shift
OUT=$R/gpurun_out/stats_$STAGE
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats -d $OUT/raw -- python $R/tools/stage_bench.py $STAGE "$@" > $OUT/log.txt 2>&1
DB=$(find $OUT/raw -name '*.db' | head -1)
python $R/tools/rocprof_summary.py $DB $OUT/kernel_stats.md | head -12
rm -rf $OUT/raw
