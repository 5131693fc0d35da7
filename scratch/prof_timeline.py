"""Timeline of the LAST training step in a rocprofv3 rocpd database: runs of equal kernels per stream, with start offsets."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("select start, end, name, stream_id from kernels order by start"))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:44]
starts = [r[0] for r in rows if "gru_fwd_pp_kernel<1>" in r[2] or "gru_fwd_x6pp_kernel<1>" in r[2] or "gru_fwd_persist_kernel<4, 1, 2" in r[2]]      # the encoder forward scan opens a step
t0 = starts[int(sys.argv[3]) if len(sys.argv) > 3 else -1]
t_end = starts[(int(sys.argv[3]) if len(sys.argv) > 3 else -1) + 1] if len(sys.argv) > 3 else 1 << 62
rows = [r for r in rows if t0 <= r[0] < t_end]
runs = []
for s, e, n, st in rows:
    n = short(n)
    if runs and runs[-1][2] == n and runs[-1][3] == st and s - runs[-1][1] < 50000:
        runs[-1][1] = e; runs[-1][4] += 1; runs[-1][5] += e - s
    else:
        runs.append([s, e, n, st, 1, e - s])
minus = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
for s, e, n, st, c, busy in runs:
    if (e - s) / 1e3 >= minus:
        print("%9.1f us  +%8.1f us  stream %d  x%-4d busy %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, st, c, busy / 1e3, n))
print("step span %.1f us" % ((max(r[1] for r in rows) - t0) / 1e3))
# idle analysis: time inside the step during which NO kernel runs (launch gaps of the graph), and the largest gaps
iv = sorted((s, e, short(n)) for s, e, n, st in rows)
cur_e, idle, gaps, last = iv[0][1], 0, [], iv[0][2]
for s, e, n in iv[1:]:
    if s > cur_e:
        idle += s - cur_e
        gaps.append((s - cur_e, (cur_e - t0) / 1e3, last, n))
    if e > cur_e:
        cur_e, last = e, n
print("no kernel running for %.1f us of the step (%d gaps); largest:" % (idle / 1e3, len(gaps)))
for g, at, a, b in sorted(gaps, reverse=True)[:14]:
    print("   %6.1f us at %9.1f us   after %-40s before %s" % (g / 1e3, at, a, b))
