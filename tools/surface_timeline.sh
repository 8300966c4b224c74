#!/bin/bash
# Run ON THE GPU BOX: start / end of every GPU operation (kernels and copies) of a few frames in the middle of `bench.py --mode surface`, microseconds
# from the frame's first operation -- what the device does, and for how long it waits, while the class surfaces of one frame run.  -> gpurun_out/<tag>.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-surface_timeline}
OUT=$R/gpurun_out/tl_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/raw -- python $R/bench.py --mode surface --no-cpu-baseline --frames 60 > $OUT/log.txt 2>&1
DB=$(find $OUT/raw -name '*.db' | head -1)
python - "$DB" > $R/gpurun_out/$TAG.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
def rows_of(t):
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % t)]
    name = "name" if "name" in cols else ([c for c in cols if "name" in c] or [None])[0]
    if not ("start" in cols and "end" in cols): return []
    return [(str(r[0] if r[0] is not None else t).split("(")[0][:40] if name else t, r[1], r[2]) for r in cur.execute("select %s, start, end from %s" % (name or "'%s'" % t, t))]
ops = rows_of("kernels")
for t in tabs:
    if "memory_cop" in t.lower() and t != "kernels":
        ops += [("copy:" + n, s, e) for n, s, e in rows_of(t)]
ops.sort(key=lambda r: r[1])
# frames: every k_bgr2gray16 starts one; print frames 20..22 of the first loop that has at least 40
starts = [i for i, r in enumerate(ops) if r[0].startswith("k_bgr2gray16")]
print("tables:", [t for t in tabs if "cop" in t.lower() or t == "kernels"])
for fr in (20, 21):
    i0, i1 = starts[fr], starts[fr + 1]
    t0 = ops[i0][1]; busy = 0.0; last_end = t0
    print("--- frame", fr, "(%d operations, %.1f us wall)" % (i1 - i0, (ops[i1][1] - t0) / 1e3))
    for n, s, e in ops[i0:i1]:
        print("%-42s start %8.1f  dur %7.1f  gap before %6.1f" % (n, (s - t0) / 1e3, (e - s) / 1e3, (s - last_end) / 1e3))
        busy += (e - s) / 1e3; last_end = max(last_end, e)
    print("busy %.1f us of %.1f" % (busy, (ops[i1][1] - t0) / 1e3))
PY
rm -rf $OUT/raw
head -90 $R/gpurun_out/$TAG.txt
