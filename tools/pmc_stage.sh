#!/bin/bash
# Run ON THE GPU BOX: one rocprofv3 --pmc pass (counters in $2...) over tools/stage_bench.py <stage>; prints the per-kernel table.
# usage: tools/pmc_stage.sh <stage> COUNTER [COUNTER ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
STAGE=$1; shift
OUT=$R/gpurun_out/pmc_$STAGE
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/raw -- python $R/tools/stage_bench.py $STAGE --reps 2 > $OUT/log.txt 2>&1
python $R/tools/pmc_summary.py $OUT/raw
rm -rf $OUT/raw
