"""Per-queue busy time and inter-kernel gaps from a rocprofv3 rocpd database (kernel trace)."""
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print(cols)
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = cur.execute(f"select start, end, {qcol or '0'}, name from kernels order by start").fetchall()
t0, t1 = rows[0][0], max(r[1] for r in rows)
# analyse the last 40% of the window (steady state)
cut = t0 + (t1 - t0) * 0.6
rows = [r for r in rows if r[0] >= cut]
byq = collections.defaultdict(list)
for s, e, q, n in rows:
    byq[q].append((s, e, n))
span = (rows[-1][1] - rows[0][0]) / 1e6
print(f"window {span:.2f} ms, {len(rows)} dispatches")
for q, lst in byq.items():
    busy = sum(e - s for s, e, _ in lst) / 1e6
    gaps = [lst[i + 1][0] - lst[i][1] for i in range(len(lst) - 1)]
    pos = [g for g in gaps if g > 0]
    small = [g for g in pos if g < 50e3]
    print(f"queue {q}: {len(lst)} kernels busy {busy:.2f} ms ({100*busy/span:.1f}%), gaps<50us: n={len(small)} sum={sum(small)/1e6:.2f} ms avg={sum(small)/max(1,len(small))/1e3:.2f} us; gaps>=50us: n={len(pos)-len(small)} sum={(sum(pos)-sum(small))/1e6:.2f} ms")
# union busy across queues
ev = sorted([(s, 1) for s, e, _, _ in rows] + [(e, -1) for s, e, _, _ in rows])
cur_n, last, union = 0, None, 0
for t, d in ev:
    if cur_n > 0: union += t - last
    cur_n += d; last = t
print(f"union busy {union/1e6:.2f} ms = {100*union/1e6/span:.1f}% of the window")
