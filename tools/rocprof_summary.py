#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (or kernel_stats csv) into a per-kernel table:
calls, total / average / min / max duration.  Usage: tools/rocprof_summary.py results.db [out.md]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = list(cur.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                        "from kernels group by %s order by 3 desc" % (name, name)))
tot = sum(r[2] for r in rows) or 1
lines = ["| kernel | calls | total us | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
for n, c, s, a, mn, mx in rows:
    lines.append("| %s | %d | %.1f | %.2f | %.2f | %.2f | %.1f |" % (n.split("(")[0][:60], c, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
