#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OFF="python bench.py --mode offline --steps 3 --warmup 2 --no-cpu-baseline --frames 1024"
echo "== g8"; YGZ_OFFLINE_TRACE=1 timeout 300 $OFF 2>&1 >/dev/null | grep "offline trace" | head -12
echo "== s844"; YGZ_OFFLINE_TRACE=1 YGZ_OFF_LM_SCHED=8,4,4 timeout 300 $OFF 2>&1 >/dev/null | grep "offline trace" | head -12
