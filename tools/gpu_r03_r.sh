#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
cp ygz_slam_amd/libygz_hip.so /tmp/lib_backup.so
make -C ygz_slam_amd/csrc -B EXTRA=-DYGZ_SA_CAP2 -j16 > /dev/null 2>&1
for l in 1024 832 768 704 640 512; do echo -n "cap2 LDS=$l: "; YGZ_SA_LDS=$l python tools/stage_bench.py sparse --batch 512 --reps 5 2>&1 | tail -1; done
for l in 1024 832 768 640; do echo -n "cap2 LDS=$l step: "; YGZ_SA_LDS=$l python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['stage_ms_per_batch']['sparse_align'])"; done
cp /tmp/lib_backup.so ygz_slam_amd/libygz_hip.so
echo -n "base step: "; python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['stage_ms_per_batch']['sparse_align'])"
