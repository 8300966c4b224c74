#!/bin/bash
# Run ON THE GPU BOX: the step bench under the schedule switches (YGZ_AUX_PRIORITY, YGZ_KLT_PREP, --double-buffer); one line per variant.
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { local tag=$1; shift; env "$@" python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline ${EXTRA} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-34s %8.0f frames/s  %.3f ms/step' % ('$tag', d['value'], d['ms_per_step']))"; }
EXTRA=""
run "baseline" A=1
run "aux low priority" YGZ_AUX_PRIORITY=low
run "klt prep early" YGZ_KLT_PREP=early
run "prep early + aux low" YGZ_KLT_PREP=early YGZ_AUX_PRIORITY=low
EXTRA="--double-buffer"
run "double-buffer" A=1
run "double-buffer + aux low" YGZ_AUX_PRIORITY=low
run "double-buffer + prep early" YGZ_KLT_PREP=early
run "double-buffer + both" YGZ_KLT_PREP=early YGZ_AUX_PRIORITY=low
