#!/bin/bash
# Run ON THE GPU BOX (through gpurun): everything profiles/<round>_* and bench.py's tables come from, on ONE set of kernel sources
# (ROUND=r06 by default; round 5 ran the same script as collect_r05_profiles.sh).
#   1. kernel tables of the default bench command in TWO parts, so that every `frac` of the driver line can be reproduced from one file:
#        step  : rocprofv3 --kernel-trace --stats of `bench.py --profile-part step`  (in-step launches only)   -> <round>_step_kernel_stats.md
#        alone : the same of `bench.py --profile-part alone` (every stage by itself)                          -> <round>_alone_kernel_stats.md
#   2. --pmc FETCH_SIZE / WRITE_SIZE (own passes) of the step part -> pmc tables, traffic.json
#   3. one SQ-counter pass over the stages -> sq counters, valu_counts.json
#   4. the bench lines: default (with offline / stream / surface blocks), offline at 1024 / 512 / 256 / 128 frames, gray, surface
#   5. the resident LM alone on the offline run's windows: phase timers + PMC traffic
#   6. the surface loop: kernel table, device timeline of two frames (tools/surface_timeline.sh)
# Outputs under gpurun_out/<round>p/; tools/collect_round_profiles.sh --copy (CPU, in the container) moves them into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
ROUND=${ROUND:-r06}
if [ "${1:-}" = "--copy" ]; then
    S=$R/gpurun_out/${ROUND}p; P=$R/profiles
    for f in step_kernel_stats.md alone_kernel_stats.md pmc_fetch.md pmc_write.md sq_counters_raw.md bench_default.json bench_offline_f1024.json bench_offline_f512.json \
             bench_offline_f256.json bench_offline_f128.json bench_offline_f1024_gray.json bench_surface.json surface_kernel_stats.md surface_timeline.txt lm_phases.md lm_pmc.md offline128_trace.txt; do
        [ -s $S/$f ] && cp $S/$f $P/${ROUND}_$f
    done
    [ -s $S/traffic.json ] && cp $S/traffic.json $P/traffic.json
    [ -s $S/valu_counts.json ] && cp $S/valu_counts.json $P/valu_counts.json
    ls -la $P | grep ${ROUND}_
    exit 0
fi
OUT=$R/gpurun_out/${ROUND}p
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
for part in step alone; do
    rm -rf $OUT/raw_$part
    timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/raw_$part -- $B --profile-part $part > $OUT/log_$part.txt 2>&1
    DB=$(find $OUT/raw_$part -name '*.db' | head -1)
    [ -n "$DB" ] && python $R/tools/rocprof_summary.py $DB $OUT/${part}_kernel_stats.md > /dev/null
    rm -rf $OUT/raw_$part
done
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $B --profile-part step > $OUT/log_fetch.txt 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $B --profile-part step > $OUT/log_write.txt 2>&1
python $R/tools/pmc_summary.py $OUT/fetch > $OUT/pmc_fetch.md
python $R/tools/pmc_summary.py $OUT/write > $OUT/pmc_write.md
rm -rf $OUT/fetch $OUT/write
python $R/tools/make_traffic.py $OUT/pmc_fetch.md $OUT/pmc_write.md $OUT/traffic.json 512
bash $R/tools/pmc_one_pass.sh > /dev/null 2>&1
cp $R/gpurun_out/sq_counters.md $OUT/sq_counters_raw.md
cd $R
# the two tables the default line reads, in place BEFORE the line is recorded (so that it carries counters of these very kernel sources)
NKP=$(timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['config']['keypoints_per_frame'])")
python tools/make_valu_counts.py $OUT/sq_counters_raw.md $OUT/valu_counts.json 256 $NKP
cp $OUT/traffic.json $OUT/valu_counts.json $R/profiles/
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
for f in 1024 512 256 128; do timeout 300 python bench.py --mode offline --frames $f --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_offline_f$f.json 2> $OUT/off_$f.err; done
timeout 300 python bench.py --mode offline --frames 1024 --steps 5 --warmup 2 --no-cpu-baseline --upload gray > $OUT/bench_offline_f1024_gray.json 2> $OUT/off_gray.err
YGZ_OFFLINE_TRACE=1 timeout 300 python bench.py --mode offline --frames 128 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 > /dev/null | grep "offline host" | tail -14 > $OUT/offline128_trace.txt
timeout 300 python bench.py --mode surface > $OUT/bench_surface.json 2> $OUT/surface.err
cd /tmp
rm -rf $OUT/raw_s
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/raw_s -- python $R/bench.py --mode surface --no-cpu-baseline > $OUT/log_surface.txt 2>&1
DB=$(find $OUT/raw_s -name '*.db' | head -1)
[ -n "$DB" ] && python $R/tools/rocprof_summary.py $DB $OUT/surface_kernel_stats.md > /dev/null
rm -rf $OUT/raw_s
bash $R/tools/surface_timeline.sh ${ROUND}_surface_timeline > /dev/null 2>&1; cp $R/gpurun_out/${ROUND}_surface_timeline.txt $OUT/surface_timeline.txt
cd /tmp
YGZ_LM_DEBUG=1 timeout 300 python $R/tools/lm_insitu.py --frames 256 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -10 > $OUT/lm_phases.md
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/lmf -- python $R/tools/lm_insitu.py --frames 256 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/lmw -- python $R/tools/lm_insitu.py --frames 256 > /dev/null 2>&1
{ echo "FETCH_SIZE (KiB per dispatch; x 2 x 1024 = bytes fetched, upper bound for this gather kernel)"; python $R/tools/pmc_summary.py $OUT/lmf | grep -E "kernel|---|k_ba_lm_team|k_win_project";
  echo; echo "WRITE_SIZE (KiB per dispatch)"; python $R/tools/pmc_summary.py $OUT/lmw | grep -E "kernel|---|k_ba_lm_team|k_win_project"; } > $OUT/lm_pmc.md
rm -rf $OUT/lmf $OUT/lmw
ls -la $OUT | head -40
python - $OUT/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.0f ms/step %.3f" % (d["value"], d["ms_per_step"]))
print("roofline", {k: d["roofline"].get(k) for k in ("achieved","frac","avg_launch_us","frac_alone","avg_launch_us_alone","traffic")})
print("offline", d["offline"].get("value"), d["offline"].get("ms_per_step"))
print("surface", {k: d["surface"].get(k) for k in ("frames_per_s","vs_cpu_1core")})
PY
