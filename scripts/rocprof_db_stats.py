#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 results database (rocpd sqlite), in the column layout of its `--stats` CSV.
    scripts/rocprof_db_stats.py <results.db> <out.csv>"""
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = list(c.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows) or 1
with open(out, "w") as f:
    f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"\n')
    for r in rows:
        f.write('"%s",%d,%d,%.6f,%.2f,%d,%d\n' % (r[0], r[1], r[2], r[3], 100.0 * r[2] / tot, r[4], r[5]))
