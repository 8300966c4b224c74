#!/bin/bash
# Run ON THE GPU BOX: sparse-alignment stage alone at two batch sizes + the step bench (A/B of a kernel change)
R=${GRAFT_REPO_ROOT:-/root/repo}
python $R/tools/stage_bench.py sparse --batch 256 --reps 5
python $R/tools/stage_bench.py sparse --batch 512 --reps 5
python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step %8.0f frames/s  %.3f ms/step' % (d['value'], d['ms_per_step']))"
