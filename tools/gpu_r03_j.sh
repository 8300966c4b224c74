#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03j
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_offline.py -q -m gpu --no-header -rf -k "klt or depth_filter" > gpurun_out/r03j/pytest.txt 2>&1
grep -n "AssertionError\|passed\|failed" gpurun_out/r03j/pytest.txt | cut -c1-300 | head
