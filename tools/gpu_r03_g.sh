#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03g
export TMPDIR=/tmp
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    s=d.get("stream",{})
    print("%-22s step %.0f | stream bgr %s gray %s" % (sys.argv[2], d["value"], s.get("bgr",{}).get("value"), s.get("gray",{}).get("value")), s.get("error",""))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
B="python bench.py --no-cpu-baseline --offline-frames 64"
for bufs in 2 3 4; do for ov in 0 1; do
  YGZ_STREAM_BUFS=$bufs YGZ_STREAM_OVERLAP=$ov timeout 300 $B > gpurun_out/r03g/s_${bufs}_${ov}.json 2> gpurun_out/r03g/s_${bufs}_${ov}.err; show gpurun_out/r03g/s_${bufs}_${ov}.json "bufs$bufs ov$ov"
done; done
GPU_MAX_HW_QUEUES=16 YGZ_STREAM_BUFS=3 YGZ_STREAM_OVERLAP=1 timeout 300 $B > gpurun_out/r03g/s_3_1_q16.json 2> gpurun_out/r03g/s_3_1_q16.err; show gpurun_out/r03g/s_3_1_q16.json "bufs3 ov1 q16"
tail -n 3 gpurun_out/r03g/s_3_0.err | cut -c1-300
