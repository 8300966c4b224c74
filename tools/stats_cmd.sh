#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 --kernel-trace --stats over an arbitrary bench.py command line; per-kernel table -> gpurun_out/<tag>_kernel_stats.md
# usage: tools/stats_cmd.sh <tag> <bench.py arguments...>
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$R/gpurun_out/stats_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/raw -- python $R/bench.py "$@" > $OUT/log.txt 2>&1
DB=$(find $OUT/raw -name '*.db' | head -1)
python $R/tools/rocprof_summary.py $DB $R/gpurun_out/${TAG}_kernel_stats.md | head -24
grep -h '^{' $OUT/log.txt | tail -1 > $R/gpurun_out/${TAG}_bench.json
rm -rf $OUT/raw
