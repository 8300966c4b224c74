#!/bin/bash
# (CPU, after `gpurun -- bash tools/gpu_r04.sh z`) copy the closing batch's tables from gpurun_out/ into profiles/ under their round-4 names
cd "$(dirname "$0")/.." || exit 1
G=gpurun_out; P=profiles
cp $G/r04_v1/kernel_stats.md $P/r04_v1_kernel_stats.md; cp $G/r04_v1/pmc_fetch.md $P/r04_v1_pmc_fetch.md; cp $G/r04_v1/pmc_write.md $P/r04_v1_pmc_write.md
cp $G/r04_v1/traffic.json $P/traffic.json; cp $G/r04_v1/bench.json $P/r04_v1_bench.json
cp $G/sq_counters.md $P/r04_sq_counters_raw.md; cp $G/r04_step_timeline.txt $P/r04_step_timeline.txt
cp $G/r04z/bench_default.json $P/r04_bench_default.json
for f in 1024 512 256 128; do cp $G/r04z/off_f$f.json $P/r04_bench_offline_f$f.json; done
cp $G/r04z/off_f1024_gray.json $P/r04_bench_offline_f1024_gray.json
cp $G/r04_offline1024_kernel_stats.md $P/r04_offline1024_kernel_stats.md; cp $G/r04z/step_720p.json $P/r04_bench_step_720p.json
python tools/make_valu_counts.py $G/sq_counters.md $P/valu_counts.json 256 968.7 > /dev/null
python - <<'PY'
import json
d=json.load(open('profiles/r04_bench_default.json'))
print("default: %.0f frames/s, %.3f ms/step, step_ms %s, with transfers %.0f" % (d['value'], d['ms_per_step'], {k: round(v,3) for k,v in d['step_ms'].items() if k not in ('how',)}, d['value_with_transfers']))
r=d['roofline']; print("roofline: achieved %.0f GB/s frac %.3f, k_klt3 %.0f us in the step; traffic %s stale %s" % (r['achieved'], r['frac'], r['avg_launch_us'], r['traffic'], r['traffic_collected_on_other_kernel_sources']))
k=d['roofline_valu']['kernels']['k_klt3']; print("k_klt3 alone %.0f us, frac of issue ceiling %.3f (in step %.3f)" % (k['alone']['avg_launch_us'], k['alone']['frac'], k['frac']))
print("stages", {k: round(v,3) for k,v in d['stage_ms_per_batch'].items()})
print("cpu %.2f f/s 1 core, %s" % (d['cpu_baseline']['value'], d['cpu_baseline'].get('all_cores')))
o=d['offline']; print("offline %.0f f/s %.2f ms %s; gray %.0f" % (o['value'], o['ms_per_step'], {k: round(v,2) for k,v in o['phases_ms'].items()}, o['gray']['value']))
for f in (1024,512,256,128):
    x=json.load(open('profiles/r04_bench_offline_f%d.json'%f)); print(f, round(x['value']), round(x['ms_per_step'],2), {k: round(v,2) for k,v in x['phases_ms'].items()})
PY
