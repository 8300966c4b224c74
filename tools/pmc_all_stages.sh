#!/bin/bash
# Run ON THE GPU BOX: SQ occupancy / issue counters of every stage of the bench step, one rocprofv3 --pmc pass per stage.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/sq_counters.md
echo "# SQ counters per kernel (rocprofv3 --pmc, tools/stage_bench.py <stage> --reps 2, batch 256; mean per dispatch)" > $OUT
for st in detect match klt direct sparse ba; do
  echo "" >> $OUT; echo "## stage: $st" >> $OUT
  bash $R/tools/pmc_stage.sh $st SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE 2>&1 | grep -E "^\|" >> $OUT
done
cat $OUT
