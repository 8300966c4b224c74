#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_surface.py -q -m gpu --no-header -rf -k "ba_optimize or resident_windows or team_size" 2>&1 | tail -5
YGZ_LM_DEBUG=1 timeout 120 python tools/lm_phase_probe.py 2>&1 | tail -4
timeout 120 python tools/lm_probe.py 2>&1 | tail -6
