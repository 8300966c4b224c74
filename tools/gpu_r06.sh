#!/bin/bash
# Run ON THE GPU BOX (through gpurun): tools/gpu_r06.sh <batch-name> -- the measurement batches of round 6, one case per batch.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
B=${1:-a}
OUT=gpurun_out/r06$B
mkdir -p $OUT
surf() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))['surface']; c=d.pop('cpu_oracle_same_loop',{}); d.pop('what'); u=d.pop('unchanged',{}); u.pop('what',None)
print("batch    ", json.dumps(d)); print("unchanged", json.dumps(u)); print("oracle   ", c.get('frames_per_s'))
PY
}
case $B in
a)  # the unchanged caller: per-candidate FindDirectProjection behind one speculative launch -- tests, then the surface block with `unchanged`
    timeout 900 python -m pytest tests/test_gpu_surface.py -x -q > $OUT/pytest_surface.log 2>&1; tail -15 $OUT/pytest_surface.log
    timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "per_candidate or given_angle or 10x2000 or track_local_map" > $OUT/pytest_new.log 2>&1; tail -15 $OUT/pytest_new.log
    timeout 400 python bench.py --mode surface > $OUT/surface.json 2> $OUT/surface.err; tail -3 $OUT/surface.err; surf $OUT/surface.json
    ;;
p)  # pose-only BA with the lane's features in registers: parity, then the surface loop (OptimizeCurrentPoseOnly per frame) and its kernel table
    timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_surface.py tests/test_gpu_offline.py -x -q -k "pose_only or surface or handover or unchanged" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
    timeout 400 python bench.py --mode surface --no-cpu-baseline > $OUT/surface.json 2> $OUT/surface.err; tail -3 $OUT/surface.err; surf $OUT/surface.json
    ;;
l)  # SparseImgAlign(LevenbergMarquardt): the computeResiduals primitive against the oracle, the class surface against the oracle's LM
    timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_surface.py tests/test_gpu_switches.py -x -q -k "sparse or class_surfaces or alternate" > $OUT/pytest.log 2>&1; grep -v amdgpu.ids $OUT/pytest.log | tail -25
    ;;
k)  # kernel time inside the surface loop (rocprofv3 kernel trace of bench.py --mode surface)
    bash tools/stats_cmd.sh r06_surface --mode surface --no-cpu-baseline
    cp gpurun_out/r06_surface_kernel_stats.md $OUT/; head -24 $OUT/r06_surface_kernel_stats.md
    ;;
u)  # ygz_hip_ba_upload keeps the slot's allocation and sends five packed regions: BA / LM / offline tests, then the host phases of LocalBAG2O
    timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_surface.py tests/test_gpu_offline.py tests/test_gpu_switches.py -x -q -k "ba or BA or lm or surface or unchanged or offline or window or ceres" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
    YGZ_HOST_TRACE=1 timeout 400 python bench.py --mode surface --no-cpu-baseline > $OUT/surface.json 2> $OUT/surface.err; grep -i "trace\|upload\|graph" $OUT/surface.err | tail -12; surf $OUT/surface.json
    ;;
f)  # LocalBAG2O's results in one transfer, the sparse alignment's result written to page-locked memory by the kernel: suite + host phases
    timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; grep -n "passed\|failed\|error" $OUT/pytest.log | tail -5
    YGZ_HOST_TRACE=1 timeout 400 python bench.py --mode surface --no-cpu-baseline > $OUT/surface.json 2> $OUT/surface.err
    grep "ms per call" $OUT/surface.err; surf $OUT/surface.json | cut -c1-420
    ;;
q)  # the speculative launch of the unchanged caller queued at the end of Matcher::SparseImageAlignment (collected at the first per-candidate call): tests, A/B
    timeout 900 python -m pytest tests/test_gpu_surface.py tests/test_gpu_parity.py -x -q -k "surface or unchanged or per_candidate or class_surfaces or loop" > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -3
    for pl in 1 0 1 0; do
        YGZ_FDP_PRELAUNCH=$pl timeout 400 python bench.py --mode surface --no-cpu-baseline > $OUT/surface_pl$pl.json 2> $OUT/surface_pl$pl.err
        echo "== YGZ_FDP_PRELAUNCH=$pl"; surf $OUT/surface_pl$pl.json | grep unchanged | cut -c1-700
    done
    ;;
h)  # the whole GPU suite + the default bench line
    timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; grep -n "passed\|failed\|error" $OUT/pytest.log | tail -5
    timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; python - $OUT/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.0f ms/step %.3f" % (d["value"], d["ms_per_step"]), d.get("step_ms"))
print("roofline", {k: d["roofline"][k] for k in ("achieved","frac","avg_launch_us") if k in d["roofline"]})
s=d.get("surface",{}); print("surface", s.get("frames_per_s"), s.get("vs_cpu_1core"), "unchanged", s.get("unchanged",{}).get("frames_per_s"), s.get("unchanged",{}).get("vs_cpu_1core"))
PY
    ;;
esac
