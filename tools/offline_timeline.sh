#!/bin/bash
# Run ON THE GPU BOX: device timeline (kernels + memory copies, with their streams) of `bench.py --mode offline` -> gpurun_out/<tag>_timeline.tsv
# usage: tools/offline_timeline.sh <tag> <bench.py arguments...>
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$R/gpurun_out/tl_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/raw -- python $R/bench.py "$@" > $OUT/log.txt 2>&1
DB=$(find $OUT/raw -name '*.db' | head -1)
python - "$DB" > $R/gpurun_out/${TAG}_timeline.tsv <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
def cols(t): return [r[1] for r in cur.execute("pragma table_info(%s)" % t)]
kc = cols("kernels")
name = "name" if "name" in kc else [c for c in kc if "name" in c][0]
q = [c for c in kc if c in ("stream_id", "queue_id", "stream", "queue")]
rows = [("K", r[0].split("(")[0][:40], r[1], r[2], r[3] if len(r) > 3 else "") for r in cur.execute("select %s, start, end%s from kernels" % (name, (", " + q[0]) if q else ""))]
mt = [t for t in tabs if "memory_cop" in t]
if mt:
    mc = cols(mt[0])
    nm = [c for c in mc if c in ("name", "direction", "kind")]
    sz = [c for c in mc if c in ("size", "bytes")]
    qs = [c for c in mc if c in ("stream_id", "queue_id", "stream", "queue")]
    sel = "select %s, start, end, %s, %s from %s" % (nm[0] if nm else "'copy'", sz[0] if sz else "0", qs[0] if qs else "''", mt[0])
    rows += [("C", str(r[0])[:28] + ":" + str(r[3]), r[1], r[2], r[4]) for r in cur.execute(sel)]
rows.sort(key=lambda r: r[2])
t0 = rows[0][2]
print("# tables:", tabs, file=sys.stderr)
for k, n, a, b, s in rows:
    print("%s\t%s\t%.1f\t%.1f\t%s" % (k, n, (a - t0) / 1e3, (b - t0) / 1e3, s))
PY
rm -rf $OUT/raw
grep -h "offline trace" $OUT/log.txt
grep -h '^{' $OUT/log.txt | tail -1 > $R/gpurun_out/${TAG}_bench.json
wc -l $R/gpurun_out/${TAG}_timeline.tsv
