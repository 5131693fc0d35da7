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
# the dominant symbol by launch shape (one kernel serves the 48-tile x 16-K-range dW_hh products and several smaller weight gradients)
try:
    cols = [c[1] for c in cur.execute("pragma table_info(kernels)")]
    gcols = [c for c in ("grid_size_x", "grid_size", "grid_x") if c in cols]
    if gcols:
        g = gcols[0]
        print("\ngemm_tn_kernel by launch shape (%s = threads of the grid's x dimension; 196608 = 768 workgroups = a dW_hh-shaped product):" % g)
        q = "select %s, count(*), avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 from kernels where name like '%%gemm_tn_kernel%%' group by %s order by 2 desc" % (g, g)
        for r in cur.execute(q):
            print("   grid %8s  calls %5d  avg %9.2f us  min %9.2f  max %9.2f" % r)
    else:
        print("\n(kernels table has no grid column: %s)" % cols)
except Exception as e:
    print("\n(no per-shape breakdown: %s)" % e)
