#!/bin/bash
# Run ON THE GPU BOX (through gpurun): tools/gpu_r04.sh <batch-name> -- the measurement batches of round 4, one function per batch.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
B=${1:-a}
OUT=gpurun_out/r04$B
mkdir -p $OUT
benchline() { tag=$1; shift; timeout 300 "$@" > $OUT/$tag.json 2> $OUT/$tag.err; python - $OUT/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-24s %9.1f frames/s  ms %.3f" % (sys.argv[2], d["value"], d["ms_per_step"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
STEP="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras"
case $B in
a)  # second form of k_sparse_align + framed copies written by the pyramid kernels: parity, bit-identity against the first forms, timing
    timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
    YGZ_SA_FORM=0 YGZ_PAD_FUSE=0 python tools/step_dump.py $OUT/d_old.npz > $OUT/dump.log 2>&1
    python tools/step_dump.py $OUT/d_new.npz >> $OUT/dump.log 2>&1
    YGZ_SA_HINLINE=0 python tools/step_dump.py $OUT/d_new_b.npz >> $OUT/dump.log 2>&1
    YGZ_SA_FORM=0 YGZ_PAD_FUSE=0 python tools/step_dump.py $OUT/d7_old.npz --size 720p --batch 8 >> $OUT/dump.log 2>&1
    python tools/step_dump.py $OUT/d7_new.npz --size 720p --batch 8 >> $OUT/dump.log 2>&1
    python tools/step_dump.py --compare $OUT/d_old.npz $OUT/d_new.npz
    python tools/step_dump.py --compare $OUT/d_old.npz $OUT/d_new_b.npz
    python tools/step_dump.py --compare $OUT/d7_old.npz $OUT/d7_new.npz
    tail -5 $OUT/dump.log
    for f in 0 1; do YGZ_SA_FORM=$f python tools/stage_bench.py sparse --batch 512 --reps 5; done
    YGZ_SA_HINLINE=0 python tools/stage_bench.py sparse --batch 512 --reps 5
    YGZ_SA_PLDS=0 python tools/stage_bench.py sparse --batch 512 --reps 5
    YGZ_SA_FORM=0 python tools/stage_bench.py sparse --batch 256 --reps 5
    python tools/stage_bench.py sparse --batch 256 --reps 5
    YGZ_PAD_FUSE=0 python tools/stage_bench.py klt --batch 512 --reps 5
    python tools/stage_bench.py klt --batch 512 --reps 5
    YGZ_PAD_FUSE=0 python tools/stage_bench.py pyramid --batch 512 --reps 5
    python tools/stage_bench.py pyramid --batch 512 --reps 5
    YGZ_SA_FORM=0 YGZ_PAD_FUSE=0 benchline old $STEP
    YGZ_SA_FORM=1 YGZ_PAD_FUSE=0 benchline sa2 $STEP
    YGZ_SA_FORM=0 YGZ_PAD_FUSE=1 benchline fuse $STEP
    benchline new $STEP
    YGZ_SA_FORM=0 YGZ_PAD_FUSE=0 benchline old_b $STEP
    benchline new_b $STEP
    ;;
b)  # why the step got slower with the faster sparse alignment / without k_klt_pad: footprint and schedule matrix, two timelines
    YGZ_SA_FORM=0 YGZ_PAD_FUSE=0 benchline base $STEP
    YGZ_SA_PLDS=0 YGZ_SA_HINLINE=0 YGZ_PAD_FUSE=0 benchline sa2_p0_h0 $STEP
    YGZ_SA_PLDS=0 YGZ_SA_HINLINE=1 YGZ_PAD_FUSE=0 benchline sa2_p0_h1 $STEP
    YGZ_SA_HINLINE=0 YGZ_PAD_FUSE=0 benchline sa2_p960_h0 $STEP
    YGZ_SA_LDS=512 YGZ_SA_PLDS=512 YGZ_SA_HINLINE=0 YGZ_PAD_FUSE=0 benchline sa2_l512_p512_h0 $STEP
    YGZ_SA_LDS=512 YGZ_SA_PLDS=0 YGZ_SA_HINLINE=0 YGZ_PAD_FUSE=0 benchline sa2_l512_p0_h0 $STEP
    YGZ_SA_FORM=0 YGZ_SA_LDS=512 YGZ_PAD_FUSE=0 benchline sa1_l512 $STEP
    YGZ_SA_FORM=0 YGZ_PAD_FUSE=1 YGZ_KLT_JOIN=6 benchline fuse_join6 $STEP
    YGZ_SA_FORM=0 YGZ_PAD_FUSE=1 YGZ_KLT_JOIN=7 benchline fuse_join7 $STEP
    YGZ_SA_FORM=0 YGZ_PAD_FUSE=1 YGZ_KLT_JOIN=2 benchline fuse_join2 $STEP
    YGZ_SA_FORM=0 YGZ_PAD_FUSE=1 YGZ_KLT_JOIN=4 benchline fuse_join4 $STEP
    YGZ_SA_FORM=0 YGZ_PAD_FUSE=0 YGZ_KLT_JOIN=6 benchline nofuse_join6 $STEP
    YGZ_SA_PLDS=0 YGZ_SA_HINLINE=0 YGZ_PAD_FUSE=1 YGZ_KLT_JOIN=6 benchline sa2_p0_h0_fuse_join6 $STEP
    YGZ_SA_FORM=0 YGZ_PAD_FUSE=0 benchline base_b $STEP
    YGZ_SA_FORM=0 YGZ_PAD_FUSE=1 bash tools/timeline.sh r04b_tl_fuse > /dev/null 2>&1
    YGZ_SA_FORM=1 YGZ_PAD_FUSE=0 bash tools/timeline.sh r04b_tl_sa2 > /dev/null 2>&1
    YGZ_SA_FORM=0 YGZ_PAD_FUSE=0 bash tools/timeline.sh r04b_tl_base > /dev/null 2>&1
    ;;
c)  # resident sparse-alignment workgroups (no second round to starve), flat frame kernel
    YGZ_SA_FORM=0 YGZ_PAD_FUSE=0 python tools/step_dump.py $OUT/d_old.npz > $OUT/dump.log 2>&1
    python tools/step_dump.py $OUT/d_new.npz >> $OUT/dump.log 2>&1
    YGZ_SA_FORM=0 YGZ_PAD_FUSE=0 python tools/step_dump.py $OUT/d_old5.npz --batch 520 >> $OUT/dump.log 2>&1
    python tools/step_dump.py $OUT/d_new5.npz --batch 520 >> $OUT/dump.log 2>&1
    python tools/step_dump.py --compare $OUT/d_old.npz $OUT/d_new.npz
    python tools/step_dump.py --compare $OUT/d_old5.npz $OUT/d_new5.npz
    python tools/stage_bench.py sparse --batch 512 --reps 5
    YGZ_SA_PERSIST=0 python tools/stage_bench.py sparse --batch 512 --reps 5
    python tools/stage_bench.py pyramid --batch 512 --reps 5
    YGZ_SA_FORM=0 YGZ_PAD_FUSE=0 benchline base $STEP
    YGZ_PAD_FUSE=0 benchline persist $STEP
    YGZ_PAD_FUSE=1 benchline persist_fuse $STEP
    YGZ_PAD_FUSE=1 YGZ_KLT_JOIN=6 benchline persist_fuse_join6 $STEP
    YGZ_PAD_FUSE=1 YGZ_KLT_JOIN=2 benchline persist_fuse_join2 $STEP
    YGZ_PAD_FUSE=1 YGZ_KLT_JOIN=4 benchline persist_fuse_join4 $STEP
    YGZ_PAD_FUSE=1 YGZ_KLT_JOIN=1 benchline persist_fuse_join1 $STEP
    YGZ_PAD_FUSE=1 YGZ_SA_HINLINE=1 benchline persist_fuse_h1 $STEP
    YGZ_PAD_FUSE=1 YGZ_SA_PLDS=0 benchline persist_fuse_p0 $STEP
    YGZ_PAD_FUSE=1 YGZ_BENCH_KLT_PREPARE=1 benchline persist_fuse_prep $STEP
    YGZ_PAD_FUSE=0 YGZ_SA_PERSIST=0 benchline nopersist $STEP
    YGZ_SA_FORM=0 YGZ_PAD_FUSE=0 benchline base_b $STEP
    YGZ_PAD_FUSE=1 bash tools/timeline.sh r04c_tl_persist_fuse > /dev/null 2>&1
    YGZ_PAD_FUSE=0 bash tools/timeline.sh r04c_tl_persist > /dev/null 2>&1
    ;;
d)  # resident grid size, register cap, LK working images beside the extractor
    benchline dflt $STEP
    YGZ_BENCH_KLT_PREPARE=1 benchline prep $STEP
    YGZ_SA_GRID=128 benchline grid128 $STEP
    YGZ_SA_GRID=192 benchline grid192 $STEP
    YGZ_SA_GRID=224 benchline grid224 $STEP
    YGZ_SA_GRID=128 YGZ_BENCH_KLT_PREPARE=1 benchline grid128_prep $STEP
    YGZ_SA_GRID=192 YGZ_BENCH_KLT_PREPARE=1 benchline grid192_prep $STEP
    YGZ_SA_REGS=288 benchline r288 $STEP
    YGZ_SA_REGS=288 YGZ_BENCH_KLT_PREPARE=1 benchline r288_prep $STEP
    YGZ_SA_REGS=288 YGZ_SA_PLDS=0 benchline r288_p0 $STEP
    YGZ_SA_LDS=512 YGZ_SA_PLDS=512 benchline l512_p512 $STEP
    YGZ_BENCH_BA_EARLY=1 benchline ba_early $STEP
    YGZ_BENCH_BA_EARLY=1 YGZ_BENCH_KLT_PREPARE=1 benchline ba_early_prep $STEP
    YGZ_SA_REGS=288 python tools/stage_bench.py sparse --batch 512 --reps 5
    benchline dflt_b $STEP
    ;;
e)  # parity suite on the new defaults + the three profile passes behind profiles/
    timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
    bash tools/collect_profiles.sh r04e_prof > $OUT/collect.log 2>&1
    cat gpurun_out/r04e_prof/kernel_stats.md
    cat gpurun_out/r04e_prof/traffic.json | head -80
    ;;
f)  # BA windows with direct-projection observations: offline tests, the offline bench line
    timeout 1200 python -m pytest tests/test_gpu_offline.py -x -q > $OUT/pytest.log 2>&1; tail -15 $OUT/pytest.log
    timeout 300 python bench.py --mode offline --frames 1024 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/off.json 2> $OUT/off.err
    python - $OUT/off.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("offline %.1f frames/s  %.2f ms" % (d["value"], d["ms_per_step"]), {k: round(v,2) for k,v in d["phases_ms"].items()})
print(json.dumps(d["result_check"], indent=1))
PY
    ;;
g)  # whole parity suite (orientation check, ba_last_path, framed copies from the producers), frame-fold A/B
    timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
    YGZ_FRAME_FUSE=0 python tools/stage_bench.py pyramid --batch 512 --reps 5
    python tools/stage_bench.py pyramid --batch 512 --reps 5
    YGZ_PAD_FUSE=0 python tools/stage_bench.py pyramid --batch 512 --reps 5
    YGZ_FRAME_FUSE=0 benchline frame_kernel $STEP
    benchline dflt $STEP
    YGZ_BENCH_KLT_PREPARE=1 benchline prep $STEP
    YGZ_FRAME_FUSE=0 benchline frame_kernel_b $STEP
    benchline dflt_b $STEP
    ;;
h)  # the rest of the parity suite after the test fix
    timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; tail -8 $OUT/pytest.log
    ;;
i)  # keyframe-free last chunk (LM of the last windows beside it), LM phases on the direct-mode graphs, offline timeline
    timeout 900 python -m pytest tests/test_gpu_offline.py tests/test_gpu_parity.py -m gpu -q -k "offline or resident_windows or framed" > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
    OFF="python bench.py --mode offline --frames 1024 --steps 3 --warmup 1 --no-cpu-baseline"
    offline() { tag=$1; shift; timeout 300 "$@" > $OUT/$tag.json 2> $OUT/$tag.err; python - $OUT/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-16s %9.1f frames/s  ms %.2f  %s" % (sys.argv[2], d["value"], d["ms_per_step"], {k: round(v,2) for k,v in d["phases_ms"].items()}))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
    }
    YGZ_OFF_KF_TAIL=0 offline tail0 $OFF
    offline tail1 $OFF
    YGZ_OFF_KF_TAIL=0 offline tail0_b $OFF
    offline tail1_b $OFF
    offline f128 python bench.py --mode offline --frames 128 --steps 3 --warmup 1 --no-cpu-baseline
    YGZ_OFF_KF_TAIL=0 offline f128_tail0 python bench.py --mode offline --frames 128 --steps 3 --warmup 1 --no-cpu-baseline
    YGZ_LM_DEBUG=1 timeout 200 python bench.py --mode offline --frames 256 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep "lm-debug" | tail -3
    bash tools/offline_timeline.sh r04i --mode offline --frames 1024 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
    python - <<'PY'
rows=[l.rstrip("\n").split("\t") for l in open("gpurun_out/r04i_timeline.tsv")]
# the last run: everything after the last big gap; print the kernels after the last upload
ks=[r for r in rows if r[0]=="K"]
cs=[r for r in rows if r[0]=="C" and float(r[3])-float(r[2])>300]
t_last=float(cs[-1][3])
print("last big copy ends", t_last)
for r in ks:
    if float(r[3])>t_last-1500: print("%-36s %9.1f %9.1f  %s" % (r[1], float(r[2])-t_last, float(r[3])-t_last, r[4]))
PY
    ;;
j)  # factorised H block of the sparse alignment, LM retry / degenerate-window paths: parity + timing
    timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log
    python tools/stage_bench.py sparse --batch 512 --reps 5
    python tools/stage_bench.py sparse --batch 256 --reps 5
    benchline dflt $STEP
    benchline dflt_b $STEP
    ;;
k)  # the resident LM on the run's own windows, alone
    python tools/lm_insitu.py --frames 256 2>&1 | grep -v amdgpu | tail -12
    YGZ_LM_DEBUG=1 python tools/lm_insitu.py --frames 128 2>&1 | grep "lm-debug" | tail -2
    ;;
r)  # the chunk that ends with the shard's last keyframe kept short (the last LM launch starts earlier); LM records / reset as one launch each
    timeout 900 python -m pytest tests/test_gpu_offline.py tests/test_gpu_parity.py -m gpu -q -k "offline or resident or ba_ or window" > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
    offline() { tag=$1; shift; timeout 300 "$@" > $OUT/$tag.json 2> $OUT/$tag.err; python - $OUT/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-16s %9.1f frames/s  ms %.2f  %s" % (sys.argv[2], d["value"], d["ms_per_step"], {k: round(v,2) for k,v in d["phases_ms"].items()}))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
    }
    for F in 1024 128; do
        OFF="python bench.py --mode offline --frames $F --steps 5 --warmup 2 --no-cpu-baseline"
        YGZ_OFF_KF_SMALL=0 offline f${F}_small0 $OFF
        offline f${F}_small8 $OFF
        YGZ_OFF_KF_SMALL=4 offline f${F}_small4 $OFF
        YGZ_OFF_KF_SMALL=0 offline f${F}_small0_b $OFF
        offline f${F}_small8_b $OFF
    done
    YGZ_OFFLINE_TRACE=1 timeout 300 python bench.py --mode offline --frames 1024 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep "offline trace" | head -16
    bash tools/offline_timeline.sh r04r --mode offline --frames 1024 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
    python tools/offline_timeline_summary.py gpurun_out/r04r_timeline.tsv | tail -75
    ;;
s)  # where the host is at the end of an offline run (per-call log of the last run), the LM alone with the new reset launch against the memsets
    python tools/lm_insitu.py --frames 256 2>&1 | grep -v amdgpu | tail -8
    YGZ_LM_RESET_MEMSETS=1 python tools/lm_insitu.py --frames 256 2>&1 | grep -v amdgpu | tail -6
    for SM in 0 8; do
        echo "== KF_SMALL=$SM"
        YGZ_OFF_KF_SMALL=$SM YGZ_OFFLINE_TRACE=1 timeout 300 python bench.py --mode offline --frames 1024 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep "offline host" | awk '$3 > 40' | head -80
    done
    echo "== f128 KF_SMALL=0"
    YGZ_OFF_KF_SMALL=0 YGZ_OFFLINE_TRACE=1 timeout 300 python bench.py --mode offline --frames 128 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep "offline host" | head -60
    ;;
t)  # the end of an offline run: hardware queues (streams that share one serialise) and lanes ahead (a lane is re-used only when its chunk is done)
    offline() { tag=$1; shift; timeout 300 "$@" > $OUT/$tag.json 2> $OUT/$tag.err; python - $OUT/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-22s %9.1f frames/s  ms %.2f  %s" % (sys.argv[2], d["value"], d["ms_per_step"], {k: round(v,2) for k,v in d["phases_ms"].items()}))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
    }
    for F in 1024 128; do
        OFF="python bench.py --mode offline --frames $F --steps 5 --warmup 2 --no-cpu-baseline"
        offline f${F}_base $OFF
        GPU_MAX_HW_QUEUES=8 offline f${F}_q8 $OFF
        GPU_MAX_HW_QUEUES=16 offline f${F}_q16 $OFF
        YGZ_OFF_AHEAD=2 offline f${F}_ahead2 $OFF
        GPU_MAX_HW_QUEUES=8 YGZ_OFF_AHEAD=2 offline f${F}_q8_ahead2 $OFF
        GPU_MAX_HW_QUEUES=8 YGZ_OFF_AHEAD=2 YGZ_OFF_KF_SMALL=0 offline f${F}_q8_ahead2_s0 $OFF
        GPU_MAX_HW_QUEUES=8 YGZ_OFF_KF_SMALL=0 offline f${F}_q8_s0 $OFF
        offline f${F}_base_b $OFF
    done
    GPU_MAX_HW_QUEUES=8 $STEP | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step q8', d['value'], d['ms_per_step'])"
    $STEP | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step base', d['value'], d['ms_per_step'])"
    ;;
u)  # the keyframe-free frames behind the last windows processed at the end, grouped into chunks of several ranges: the last LM launch beside them
    timeout 900 python -m pytest tests/test_gpu_offline.py -m gpu -q -x -k "offline or window" > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
    offline() { tag=$1; shift; timeout 300 "$@" > $OUT/$tag.json 2> $OUT/$tag.err; python - $OUT/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-22s %9.1f frames/s  ms %.2f  %s" % (sys.argv[2], d["value"], d["ms_per_step"], {k: round(v,2) for k,v in d["phases_ms"].items()}))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
    }
    OFF="python bench.py --mode offline --frames 1024 --steps 5 --warmup 2 --no-cpu-baseline"
    for D in 0 4 8 13 16 0 13; do YGZ_OFF_DEFER=$D offline f1024_defer$D $OFF; done
    YGZ_OFF_DEFER=13 YGZ_OFF_KF_SMALL=0 offline f1024_defer13_s0 $OFF
    for F in 128 256 512; do
        OFF="python bench.py --mode offline --frames $F --steps 5 --warmup 2 --no-cpu-baseline"
        for D in 0 16 0 16; do YGZ_OFF_DEFER=$D offline f${F}_defer$D $OFF; done
        YGZ_OFF_DEFER=0 YGZ_OFF_KF_SMALL=0 offline f${F}_defer0_s0 $OFF
        YGZ_OFF_DEFER=16 YGZ_OFF_KF_SMALL=0 offline f${F}_defer16_s0 $OFF
    done
    YGZ_OFF_DEFER=13 bash tools/offline_timeline.sh r04u --mode offline --frames 1024 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
    python tools/offline_timeline_summary.py gpurun_out/r04u_timeline.tsv | grep -v "rocclr\|k_pyr\|k_scharr\|k_track\|k_match\|k_compact\|k_detect" | tail -50
    ;;
v)  # tuning of the deferred plan: frames per deferred chunk, last chunk of the main pass, number of gaps; short shards on the defaults
    offline() { tag=$1; shift; timeout 300 "$@" > $OUT/$tag.json 2> $OUT/$tag.err; python - $OUT/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-22s %9.1f frames/s  ms %.2f  %s" % (sys.argv[2], d["value"], d["ms_per_step"], {k: round(v,2) for k,v in d["phases_ms"].items()}))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
    }
    OFF="python bench.py --mode offline --frames 1024 --steps 5 --warmup 2 --no-cpu-baseline"
    offline f1024_default $OFF
    YGZ_OFF_DEFER=0 offline f1024_defer0 $OFF
    YGZ_OFF_DEFER_GROUP=32 offline f1024_group32 $OFF
    YGZ_OFF_DEFER_GROUP=91 offline f1024_group91 $OFF
    YGZ_OFF_LAST_MAIN=0 offline f1024_lastmain32 $OFF
    YGZ_OFF_LAST_MAIN=8 offline f1024_lastmain8 $OFF
    YGZ_OFF_DEFER=10 offline f1024_defer10 $OFF
    YGZ_OFF_DEFER=16 offline f1024_defer16 $OFF
    offline f1024_default_b $OFF
    offline f512_default python bench.py --mode offline --frames 512 --steps 5 --warmup 2 --no-cpu-baseline
    YGZ_OFF_DEFER=8 offline f512_defer8 python bench.py --mode offline --frames 512 --steps 5 --warmup 2 --no-cpu-baseline
    offline f128_default python bench.py --mode offline --frames 128 --steps 5 --warmup 2 --no-cpu-baseline
    offline f1024_gray python bench.py --mode offline --frames 1024 --upload gray --steps 5 --warmup 2 --no-cpu-baseline
    YGZ_OFF_DEFER=0 offline f1024_gray_defer0 python bench.py --mode offline --frames 1024 --upload gray --steps 5 --warmup 2 --no-cpu-baseline
    bash tools/offline_timeline.sh r04v --mode offline --frames 1024 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
    python tools/offline_timeline_summary.py gpurun_out/r04v_timeline.tsv | grep -v "rocclr\|k_pyr\|k_scharr\|k_track\|k_match\|k_compact\|k_detect\|k_trel" | tail -60
    ;;
w)  # the host enqueues every chunk without waiting for a lane (per-chunk result rows): offline tests, plans with and without it
    timeout 900 python -m pytest tests/test_gpu_offline.py -m gpu -q -x -k "offline or window" > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
    offline() { tag=$1; shift; timeout 300 "$@" > $OUT/$tag.json 2> $OUT/$tag.err; python - $OUT/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-22s %9.1f frames/s  ms %.2f  %s" % (sys.argv[2], d["value"], d["ms_per_step"], {k: round(v,2) for k,v in d["phases_ms"].items()}))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
    }
    OFF="python bench.py --mode offline --frames 1024 --steps 5 --warmup 2 --no-cpu-baseline"
    offline f1024_default $OFF
    YGZ_OFF_LAST_MAIN=0 offline f1024_lastmain32 $OFF
    YGZ_OFF_RUN_AHEAD=0 YGZ_OFF_LAST_MAIN=0 offline f1024_lastmain32_ra0 $OFF
    YGZ_OFF_DEFER=0 offline f1024_defer0 $OFF
    YGZ_OFF_DEFER=0 YGZ_OFF_RUN_AHEAD=0 offline f1024_defer0_ra0 $OFF
    YGZ_OFF_DEFER=16 offline f1024_defer16 $OFF
    YGZ_OFF_DEFER_GROUP=32 offline f1024_group32 $OFF
    offline f1024_default_b $OFF
    for F in 128 256 512; do
        OFF="python bench.py --mode offline --frames $F --steps 5 --warmup 2 --no-cpu-baseline"
        offline f${F}_default $OFF
        YGZ_OFF_RUN_AHEAD=0 offline f${F}_ra0 $OFF
        YGZ_OFF_DEFER=16 offline f${F}_defer16 $OFF
    done
    offline f1024_gray python bench.py --mode offline --frames 1024 --upload gray --steps 5 --warmup 2 --no-cpu-baseline
    YGZ_OFF_RUN_AHEAD=0 YGZ_OFF_DEFER=0 offline f1024_gray_old python bench.py --mode offline --frames 1024 --upload gray --steps 5 --warmup 2 --no-cpu-baseline
    bash tools/offline_timeline.sh r04w --mode offline --frames 1024 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
    python tools/offline_timeline_summary.py gpurun_out/r04w_timeline.tsv | grep -v "rocclr\|k_pyr\|k_scharr\|k_track\|k_match\|k_compact\|k_detect\|k_trel" | tail -48
    ;;
x)  # the per-point phases of the resident LM as one-wavefront tasks per chunk of 64 points, spread over all members of a team
    timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_offline.py tests/test_gpu_surface.py -m gpu -q -x -k "ba_ or resident or offline or window or surface" > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log
    python tools/lm_insitu.py --frames 256 2>&1 | grep -v amdgpu | tail -8
    YGZ_LM_DEBUG=1 python tools/lm_insitu.py --frames 128 2>&1 | grep "lm-debug" | tail -2
    python tools/lm_probe.py 2>&1 | tail -4
    for F in 128 1024; do timeout 300 python bench.py --mode offline --frames $F --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f$F', round(d['value'],1), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phases_ms'].items()})"; done
    ;;
y)  # number of deferred gaps once more with the shorter LM (3.2 ms), all shard sizes on the defaults
    offline() { tag=$1; shift; timeout 300 "$@" > $OUT/$tag.json 2> $OUT/$tag.err; python - $OUT/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-22s %9.1f frames/s  ms %.2f  %s" % (sys.argv[2], d["value"], d["ms_per_step"], {k: round(v,2) for k,v in d["phases_ms"].items()}))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
    }
    OFF="python bench.py --mode offline --frames 1024 --steps 5 --warmup 2 --no-cpu-baseline"
    for D in 13 6 9 0 13 9; do YGZ_OFF_DEFER=$D offline f1024_defer$D $OFF; done
    for F in 512 256 128; do offline f${F} python bench.py --mode offline --frames $F --steps 5 --warmup 2 --no-cpu-baseline; done
    YGZ_OFF_DEFER=4 offline f512_defer4 python bench.py --mode offline --frames 512 --steps 5 --warmup 2 --no-cpu-baseline
    ;;
c2) # chunk size and lanes of a short shard (what one rank of an 8-rank run does)
    for CH in 32 16 24 48 64; do for LN in 3 4; do
        YGZ_OFF_CHUNK=$CH timeout 120 python bench.py --mode offline --frames 128 --lanes $LN --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f128 chunk $CH lanes $LN', round(d['value'],1), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phases_ms'].items()})"
    done; done
    ;;
ra) # limited run-ahead of the host: N chunks enqueued beyond one per lane before the oldest chunk's records are read
    for F in 1024 128; do for RA in 0 1 2 3 0 1; do
        YGZ_OFF_RUN_AHEAD=$RA timeout 120 python bench.py --mode offline --frames $F --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f$F ahead $RA', round(d['value'],1), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['phases_ms'].items()})"
    done; done
    ;;
z)  # round-4 closing batch: full GPU suite, the three rocprofv3 passes of the default command, the SQ pass, the step timeline, the default
    # bench line with its extra blocks, the offline lines per shard size, the kernel statistics and the device timeline of the offline mode
    timeout 900 python -m pytest tests -q -m gpu --no-header -rf 2>&1 | tail -4
    bash tools/collect_profiles.sh r04_v1 > $OUT/collect.log 2>&1; tail -3 $OUT/collect.log
    bash tools/pmc_one_pass.sh > $OUT/sq.log 2>&1; tail -2 $OUT/sq.log | cut -c1-200
    bash tools/timeline.sh r04_step_timeline > $OUT/timeline.log 2>&1; tail -2 $OUT/timeline.log | cut -c1-200
    timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 1200 $OUT/bench_default.json
    bash tools/stats_cmd.sh r04_offline1024 --mode offline --frames 1024 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -26
    for f in 1024 512 256 128; do
        timeout 300 python bench.py --mode offline --frames $f --steps 5 --warmup 2 --no-cpu-baseline > $OUT/off_f$f.json 2>/dev/null; cut -c1-300 $OUT/off_f$f.json
    done
    timeout 300 python bench.py --mode offline --frames 1024 --steps 5 --warmup 2 --no-cpu-baseline --upload gray > $OUT/off_f1024_gray.json 2>/dev/null; cut -c1-300 $OUT/off_f1024_gray.json
    bash tools/offline_timeline.sh r04_final --mode offline --frames 1024 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
    timeout 200 python bench.py --size 720p --batch 128 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/step_720p.json 2>/dev/null; cut -c1-300 $OUT/step_720p.json
    ;;
l)  # Pixel2Camera stored once: sparse-alignment parity + timing
    timeout 900 python -m pytest tests -m gpu -q -k "sparse or golden or offline or surface" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
    python tools/stage_bench.py sparse --batch 512 --reps 5
    benchline dflt $STEP
    ;;
m)  # unaligned 8-byte window loads (no v_alignbyte / address masking): parity + stage and step timing
    timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
    for st in klt detect direct sparse; do python tools/stage_bench.py $st --batch 512 --reps 5; done
    benchline dflt $STEP
    benchline dflt_b $STEP
    ;;
n)  # k_klt3: fourth window row from the neighbouring lane (LDS crossbar) instead of a fourth gather
    timeout 900 python -m pytest tests -m gpu -q -k "klt or surface or offline_sharded or golden" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
    YGZ_KLT_ROW_SHARE=0 python tools/step_dump.py $OUT/d_off.npz > $OUT/dump.log 2>&1
    python tools/step_dump.py $OUT/d_on.npz >> $OUT/dump.log 2>&1
    python tools/step_dump.py --compare $OUT/d_off.npz $OUT/d_on.npz
    YGZ_KLT_ROW_SHARE=0 python tools/stage_bench.py klt --batch 512 --reps 5
    python tools/stage_bench.py klt --batch 512 --reps 5
    YGZ_KLT_ROW_SHARE=0 benchline share0 $STEP
    benchline share1 $STEP
    YGZ_KLT_ROW_SHARE=0 benchline share0_b $STEP
    benchline share1_b $STEP
    ;;
o)  # the keyframe-free gaps behind the last keyframes processed at the end of the shard (the last LM launches run beside them)
    timeout 900 python -m pytest tests/test_gpu_offline.py tests/test_gpu_bench.py -m gpu -q > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
    offline() { tag=$1; shift; timeout 300 "$@" > $OUT/$tag.json 2> $OUT/$tag.err; python - $OUT/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-16s %9.1f frames/s  ms %.2f  %s" % (sys.argv[2], d["value"], d["ms_per_step"], {k: round(v,2) for k,v in d["phases_ms"].items()}))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
    }
    for d in 0 4 8 12 16; do YGZ_OFF_DEFER=$d offline f1024_d$d python bench.py --mode offline --frames 1024 --steps 3 --warmup 1 --no-cpu-baseline; done
    for d in 0 2; do YGZ_OFF_DEFER=$d offline f128_d$d python bench.py --mode offline --frames 128 --steps 3 --warmup 1 --no-cpu-baseline; done
    for d in 0 4; do YGZ_OFF_DEFER=$d offline f256_d$d python bench.py --mode offline --frames 256 --steps 3 --warmup 1 --no-cpu-baseline; done
    YGZ_OFF_DEFER=16 offline gray_d16 python bench.py --mode offline --frames 1024 --steps 3 --warmup 1 --no-cpu-baseline --upload gray
    YGZ_OFF_DEFER=0 offline gray_d0 python bench.py --mode offline --frames 1024 --steps 3 --warmup 1 --no-cpu-baseline --upload gray
    ;;
p)  # two resident batches per GPU (consecutive steps alternate): does the extractor head of one step hide behind the LK tail of the other now?
    benchline single $STEP
    benchline double $STEP --double-buffer
    YGZ_BENCH_KLT_PREPARE=1 benchline double_prep $STEP --double-buffer
    benchline single_b $STEP
    benchline double_b $STEP --double-buffer
    ;;
q)  # sparse alignment without the 16 384-cell limit (flags byte instead of per-lane bit masks)
    timeout 900 python -m pytest tests -m gpu -q -k "sparse or golden or offline_sharded or track_handover" > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
    python tools/stage_bench.py sparse --batch 512 --reps 5
    benchline dflt $STEP
    ;;
*)  echo "unknown batch $B"; exit 2 ;;
esac
