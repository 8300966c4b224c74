#!/bin/bash
# Run ON THE GPU BOX: one SQ-counter pass over tools/stage_bench.py ba (its set-up step runs the whole pipeline once) -> gpurun_out/sq_counters.md
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/tools/pmc_stage.sh ba SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE 2>&1 | grep -E "^\|" > $R/gpurun_out/sq_counters.md
cat $R/gpurun_out/sq_counters.md | cut -c1-160
