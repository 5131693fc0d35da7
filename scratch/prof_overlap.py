import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("select start, end, name, stream_id from kernels order by start"))
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy = 0; cur_end = t0; 
for s, e, n, st in rows:
    if s > cur_end: busy += 0; cur_s = s
    if e > cur_end:
        busy += e - max(s, cur_end); cur_end = e
tot = sum(e - s for s, e, _, _ in rows)
print("span %.1f ms, union-busy %.1f ms, sum of durations %.1f ms, idle %.1f ms, streams: %s" % ((t1 - t0) / 1e6, busy / 1e6, tot / 1e6, (t1 - t0 - busy) / 1e6, sorted(set(r[3] for r in rows))))
