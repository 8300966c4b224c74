#!/bin/bash
# Run ON THE GPU BOX (through gpurun): tools/gpu_r05.sh <batch-name> -- the measurement batches of round 5, one case per batch.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
B=${1:-a}
OUT=gpurun_out/r05$B
mkdir -p $OUT
benchline() { tag=$1; shift; timeout 400 "$@" > $OUT/$tag.json 2> $OUT/$tag.err; python - $OUT/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ph=d.get("phases_ms",{})
    print("%-24s %9.1f frames/s  ms %.3f  %s" % (sys.argv[2], d["value"], d["ms_per_step"], " ".join("%s=%.2f"%(k,v) for k,v in ph.items())))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
OFF="python bench.py --mode offline --steps 5 --warmup 2 --no-cpu-baseline"
case $B in
a)  # the C++ offline driver: its tests, then configs[4] at the shard sizes of 1 / 2 / 4 / 8 ranks
    timeout 900 python -m pytest tests/test_gpu_offline.py -x -q > $OUT/pytest_offline.log 2>&1; tail -5 $OUT/pytest_offline.log
    benchline off1024 $OFF --frames 1024
    benchline off1024_gray $OFF --frames 1024 --upload gray
    benchline off512 $OFF --frames 512
    benchline off256 $OFF --frames 256
    benchline off128 $OFF --frames 128
    benchline off128_b $OFF --frames 128
    tail -3 $OUT/off1024.err
    ;;
b)  # plan sweep on short shards (what one rank of an 8 / 4-rank job sees): deferred gaps, LM launch grouping, chunk size
    for f in 128 256; do
      for d in 0 2 4; do for g in 0 1; do
        benchline off${f}_d${d}_g${g} $OFF --frames $f --defer $d --lm-group $g
      done; done
      # (the chunk sizes 64 / 48 / 24 of this sweep were set through a switch of the Python driver that went with it: ygz_offline_params::chunk now)
      benchline off${f}_d2_g1_bg16 $OFF --frames $f --defer 2 --lm-group 1 --bg-budget 16
      benchline off${f}_d2_g1_l4 $OFF --frames $f --defer 2 --lm-group 1 --lanes 4
    done
    ;;
c)  # device timelines of a 128-frame shard (plain plan / deferred gaps)
    bash tools/offline_timeline.sh r05_f128 --mode offline --frames 128 --steps 2 --warmup 1 --no-cpu-baseline
    python tools/offline_timeline_summary.py gpurun_out/r05_f128_timeline.tsv "bench.py --mode offline --frames 128" > $OUT/f128_summary.md
    bash tools/offline_timeline.sh r05_f128d2 --mode offline --frames 128 --steps 2 --warmup 1 --no-cpu-baseline --defer 2
    python tools/offline_timeline_summary.py gpurun_out/r05_f128d2_timeline.tsv "bench.py --mode offline --frames 128 --defer 2" > $OUT/f128d2_summary.md
    cp gpurun_out/r05_f128_timeline.tsv gpurun_out/r05_f128d2_timeline.tsv $OUT/
    ;;
d)  # the host's side of a 128 / 256-frame shard (no profiler), and more lanes on short shards
    YGZ_OFFLINE_TRACE=1 python bench.py --mode offline --steps 2 --warmup 1 --no-cpu-baseline --frames 128 2> $OUT/trace128.txt > /dev/null; grep "offline host" $OUT/trace128.txt | tail -24
    for f in 128 256 512; do for l in 3 4 5 6 8; do benchline off${f}_l$l $OFF --frames $f --lanes $l; done; done
    ;;
e)  # the drop-in path one frame at a time (class surfaces) against the oracle loop; then its numbers
    timeout 600 python -m pytest tests/test_gpu_surface.py -x -q > $OUT/pytest_surface.log 2>&1; tail -15 $OUT/pytest_surface.log
    timeout 300 python bench.py --mode surface > $OUT/surface.json 2> $OUT/surface.err; tail -3 $OUT/surface.err; python -c "
import json; d=json.load(open('$OUT/surface.json'))['surface']; c=d.pop('cpu_oracle_same_loop',{}); d.pop('what'); print(json.dumps(d)); print(c.get('frames_per_s'))"
    YGZ_BA_LM_TEAM=1 timeout 300 python bench.py --mode surface --no-cpu-baseline > $OUT/surface_team1.json 2> $OUT/surface_team1.err; python -c "
import json; d=json.load(open('$OUT/surface_team1.json'))['surface']; d.pop('what'); print('TEAM=1', json.dumps(d))"
    ;;
f)  # kernel time inside the surface loop (rocprofv3 kernel trace of bench.py --mode surface)
    bash tools/stats_cmd.sh r05_surface --mode surface --no-cpu-baseline
    cp gpurun_out/r05_surface_kernel_stats.md $OUT/
    ;;
g)  # the resident LM with the same-XCD barrier (no L2 write-back): parity, time per launch alone, the offline run, the surface loop
    timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_offline.py -x -q -k "lm or team or resident or ba or offline or window" > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
    for x in 0 1; do
      echo "== YGZ_LM_XCD_BARRIER=$x"
      YGZ_LM_XCD_BARRIER=$x YGZ_LM_DEBUG=1 python tools/lm_insitu.py --frames 256 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -9
      YGZ_LM_XCD_BARRIER=$x benchline off128_x$x $OFF --frames 128
      YGZ_LM_XCD_BARRIER=$x benchline off128_x${x}_d2 $OFF --frames 128 --defer 2
      YGZ_LM_XCD_BARRIER=$x benchline off1024_x$x $OFF --frames 1024
    done
    timeout 300 python bench.py --mode surface --no-cpu-baseline > $OUT/surface.json 2> $OUT/surface.err; python -c "
import json; d=json.load(open('$OUT/surface.json'))['surface']; d.pop('what'); print(json.dumps(d))"
    ;;
h)  # the whole GPU suite + the default bench line (after the pruning of the experiment forms)
    timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; grep -n "passed\|failed\|error" $OUT/pytest.log | tail -5
    timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; python - $OUT/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.0f ms/step %.3f" % (d["value"], d["ms_per_step"]), d.get("step_ms"))
print("roofline", {k: d["roofline"][k] for k in ("achieved","frac","avg_launch_us") if k in d["roofline"]})
print("offline", d["offline"].get("value"), d["offline"].get("ms_per_step"), d["offline"].get("config",{}).get("exchange_backend"))
print("surface", {k: d["surface"].get(k) for k in ("frames_per_s","vs_cpu_1core","ms_per_frame")})
print("stages", d["stage_ms_per_batch"])
PY
    ;;
i)  # does the resident LM with the same-XCD barrier still slow the tracking beside it?  (batch b measured + 2.4 ms with the full barrier)
    for f in 128 256; do
      benchline off${f}_g0 $OFF --frames $f
      benchline off${f}_g1 $OFF --frames $f --lm-group 1
      benchline off${f}_g1_d2 $OFF --frames $f --lm-group 1 --defer 2
      benchline off${f}_g1_bg32 $OFF --frames $f --lm-group 1 --bg-budget 32
      benchline off${f}_g1_bg64 $OFF --frames $f --lm-group 1 --bg-budget 64
    done
    ;;
j)  # device timeline of a 128-frame shard with the LM launched per window (what does the tracking do while the first LM runs?)
    bash tools/offline_timeline.sh r05_f128g1 --mode offline --frames 128 --steps 2 --warmup 1 --no-cpu-baseline --lm-group 1
    cp gpurun_out/r05_f128g1_timeline.tsv $OUT/
    YGZ_OFFLINE_TRACE=1 python bench.py --mode offline --steps 2 --warmup 1 --no-cpu-baseline --frames 128 --lm-group 1 2> $OUT/trace128g1.txt > /dev/null; grep "offline host" $OUT/trace128g1.txt | tail -16
    ;;
k)  # team placement: a resident-LM launch beside tracking chunks spread over the XCDs (4 CUs of each) instead of owning one XCD
    timeout 600 python -m pytest tests/test_gpu_offline.py -x -q -k "sharded or 128 or rccl or degenerate" > $OUT/pytest.log 2>&1; grep -n "passed\|failed" $OUT/pytest.log | tail -2
    for f in 128 256 512 1024; do
      benchline off${f}_default $OFF --frames $f
      benchline off${f}_compact $OFF --frames $f --bg-compact
    done
    for f in 128 256; do
      for d in 2 4; do
        benchline off${f}_d${d} $OFF --frames $f --defer $d
        benchline off${f}_d${d}_g1 $OFF --frames $f --defer $d --lm-group 1
        benchline off${f}_d${d}_bg16 $OFF --frames $f --defer $d --bg-budget 16
      done
      benchline off${f}_g1 $OFF --frames $f --lm-group 1
    done
    benchline off1024_g2 $OFF --frames 1024 --lm-group 2
    benchline off1024_g4 $OFF --frames 1024 --lm-group 4
    benchline off1024_d0 $OFF --frames 1024 --defer 0
    ;;
l)  # k_fast_select pass 1a: four pixels per lane, packed 16-bit compares -- parity (score / NMS maps, keypoints, both tie rules, odd sizes) and time
    timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_switches.py -x -q -k "fast or detect or pyramid or describe or pipeline or switch or alternate" > $OUT/pytest.log 2>&1; grep -n "passed\|failed" $OUT/pytest.log | tail -2
    python tools/stage_bench.py detect --batch 512 --reps 5 --probe k_fast_select,k_describe
    benchline step python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras
    ;;
m)  # smoke() with the offline driver, the placement invariance test, the surface loop again
    python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
    timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_surface.py -x -q -k "team_size or surface or loop" > $OUT/pytest.log 2>&1; grep -n "passed\|failed" $OUT/pytest.log | tail -2
    timeout 300 python bench.py --mode surface --no-cpu-baseline > $OUT/surface.json 2> $OUT/surface.err; python -c "
import json; d=json.load(open('$OUT/surface.json'))['surface']; d.pop('what'); print(json.dumps(d))"
    ;;
n)  # extraction kernels with less index arithmetic: k_describe (scalar keypoint data, 8-byte row pieces, moments by v_dot4, float pattern table,
    # eight keypoints per wavefront in batches), k_fast_select (staging without divisions, compass pass by rows with one append per wavefront, ring
    # masks by sign bits, occupied flags prefetched, two wavefronts per tile)
    timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_switches.py tests/test_gpu_surface.py -x -q > $OUT/pytest.log 2>&1; grep -n "passed\|failed" $OUT/pytest.log | tail -2
    python tools/stage_bench.py detect --batch 512 --reps 5 --probe k_fast_select,k_describe 2>&1 | tail -3
    benchline step python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras
    timeout 300 python bench.py --mode surface --no-cpu-baseline > $OUT/surface.json 2> $OUT/surface.err; python -c "
import json; d=json.load(open('$OUT/surface.json'))['surface']; d.pop('what'); print(json.dumps(d))"
    ;;
o)  # BA linearise / pose-only BA with one division per edge (reciprocal products): the whole GPU suite, stage times, step, surface, offline
    timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; grep -n "passed\|failed" $OUT/pytest.log | tail -2
    python tools/stage_bench.py ba --batch 512 --reps 5 --probe k_ba_points 2>&1 | tail -2
    benchline step python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras
    benchline off1024 $OFF --frames 1024
    benchline off128 $OFF --frames 128
    timeout 300 python bench.py --mode surface --no-cpu-baseline > $OUT/surface.json 2> $OUT/surface.err; python -c "
import json; d=json.load(open('$OUT/surface.json'))['surface']; d.pop('what'); print(json.dumps(d))"
    ;;
esac
