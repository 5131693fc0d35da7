"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count / total / avg / min / max (us)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print("total kernel time %.1f us over %d dispatches" % (tot, sum(r[1] for r in rows)))
print("%-100s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print("%-100s %7d %12.1f %10.2f %10.2f %10.2f %6.1f" % (r[0][:100], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot))
