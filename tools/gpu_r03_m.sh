#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_switches.py -q -m gpu --no-header -rf 2>&1 | tail -8
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_offline.py -q -m gpu --no-header -rf -k "hamming or match or bf or bow" 2>&1 | tail -5
for v in 0 1; do for b in 512 256; do echo "FORM=$v batch $b"; YGZ_HAMMING_FORM=$v timeout 120 python tools/stage_bench.py match --batch $b 2>&1 | grep -v amdgpu | tail -1; done; done
for v in 0 1; do YGZ_HAMMING_FORM=$v timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('FORM=$v step', round(d['value']), d['ms_per_step'], d['roofline_valu'].get('mfma',{}).get('achieved'), d['roofline_valu'].get('mfma',{}).get('avg_launch_us'))"; done
