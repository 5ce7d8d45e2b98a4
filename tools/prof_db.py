"""Per-kernel totals of a rocprofv3 rocpd database (the .db `rocprofv3 --kernel-trace --stats` writes when no csv format is asked for).
usage: python tools/prof_db.py <results.db> <steps> [top]"""
import sqlite3
import sys

db, steps = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc").fetchall()
print("total kernel ms/step %.3f   launches/step %.0f" % (sum(r[2] for r in rows) / 1e6 / steps, sum(r[1] for r in rows) / steps))
for r in rows[:top]:
    print("%-118s %6d %8.3f ms/step %8.1f us" % (r[0][:118], r[1], r[2] / 1e6 / steps, r[3] / 1e3))
