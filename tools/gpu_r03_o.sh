#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03o
export TMPDIR=/tmp
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    s=d.get("stream",{})
    print("%-22s step %.0f | stream bgr %s gray %s" % (sys.argv[2], d["value"], round(s.get("bgr",{}).get("value",0)), round(s.get("gray",{}).get("value",0))), s.get("error",""))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
B="python bench.py --no-cpu-baseline --offline-frames 64"
for bufs in 2 3 4; do for fifo in 0 1; do
  YGZ_STREAM_BUFS=$bufs YGZ_STREAM_FIFO=$fifo timeout 300 $B > gpurun_out/r03o/s_${bufs}_${fifo}.json 2> gpurun_out/r03o/s_${bufs}_${fifo}.err; show gpurun_out/r03o/s_${bufs}_${fifo}.json "bufs$bufs fifo$fifo"
done; done
YGZ_STREAM_BUFS=3 YGZ_STREAM_FIFO=1 YGZ_STREAM_OVERLAP=1 timeout 300 $B > gpurun_out/r03o/s_3_1_ov.json 2>/dev/null; show gpurun_out/r03o/s_3_1_ov.json "bufs3 fifo1 ov1"
tail -n 3 gpurun_out/r03o/s_3_1.err | cut -c1-300
