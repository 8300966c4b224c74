#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for rg in 2 3 4; do for b in 512 256; do echo -n "RG=$rg batch $b: "; YGZ_HAMMING_RG=$rg timeout 120 python tools/stage_bench.py match --batch $b 2>&1 | grep -v amdgpu | tail -1; done; done
for rg in 3 4; do YGZ_HAMMING_RG=$rg timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu --no-header -k "hamming or match or bf" 2>&1 | tail -1; done
