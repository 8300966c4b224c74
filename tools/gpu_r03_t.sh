#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
YGZ_LM_DEBUG=1 timeout 120 python tools/lm_phase_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-700
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_switches.py tests/test_gpu_surface.py -q -m gpu --no-header -rf -k "lm or resident or ba or alternate or surface" 2>&1 | tail -6
timeout 600 python -m pytest tests/test_gpu_offline.py -q -m gpu --no-header -rf -k "offline or device_built" 2>&1 | tail -3
python bench.py --mode offline --frames 1024 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phases_ms'].items()})"
