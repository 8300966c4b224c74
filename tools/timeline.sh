#!/bin/bash
# Run ON THE GPU BOX: kernel timeline (start / end per dispatch, microseconds from the first dispatch of the step) of the LAST
# step of `bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras`, from one rocprofv3 --kernel-trace run.  -> gpurun_out/<tag>.txt (tools/timeline.sh <tag>)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-timeline}          # output: gpurun_out/<tag>.txt (experiment switches come from the environment)
OUT=$R/gpurun_out/tl_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace -d $OUT/raw -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/log.txt 2>&1
DB=$(find $OUT/raw -name '*.db' | head -1)
python - "$DB" > $R/gpurun_out/$TAG.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
qcol = [c for c in cols if "queue" in c or "stream" in c]
rows = list(cur.execute("select %s, start, end%s from kernels order by start" % (name, (", " + qcol[0]) if qcol else "")))
# the last build of the pyramid marks the start of the last timed step... the bench's stage-timing passes follow it; take the
# last k_bgr2gray16 that is followed by a k_klt3 AND preceded (3 steps earlier) by the same pattern: simply use the 4th from the end
starts = [i for i, r in enumerate(rows) if r[0].startswith("k_bgr2gray16")]
print("columns:", cols)
print("pyramid launches at rows", starts[-12:])
def dump(i0, i1):
    t0 = rows[i0][1]
    for r in rows[i0:i1]:
        print("%-34s %9.1f %9.1f  %s" % (r[0].split("(")[0][:34], (r[1] - t0) / 1e3, (r[2] - t0) / 1e3, r[3] if len(r) > 3 else ""))
# steps are consecutive pyramid launches roughly 3.4 ms apart; print the third timed step: find three consecutive gaps < 5 ms
for j in range(len(starts) - 1, 2, -1):
    g = [(rows[starts[j - k]][1] - rows[starts[j - k - 1]][1]) / 1e6 for k in range(3)]
    if all(2.0 < x < 14.0 for x in g[:2]) and 2.0 < g[2] < 60.0:       # (the gap behind the warm-up step holds the probe set-up: up to tens of ms)
        print("step starting at row", starts[j - 1], "gaps ms", g)
        dump(starts[j - 1], starts[j])
        break
PY
rm -rf $OUT/raw
tail -5 $OUT/log.txt | cut -c1-200
