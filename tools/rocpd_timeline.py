"""One steady-state step of a rocprofv3 kernel trace as a timeline: every gap >= 40 us on the busiest (compute) queue with
what ran on the other queues meanwhile, plus a coarse phase table.  Usage: rocpd_timeline.py trace.db"""
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
q = "queue_id" if "queue_id" in cols else "stream_id"
rows = cur.execute(f"select start, end, {q}, name from kernels order by start").fetchall()
def short(n):
    n = n.replace("void ", "").replace("vtxg::", "").replace("unsigned short", "bf16")
    return n[:95]
# steps are delimited by the optimizer kernel
ends = [e for s, e, qq, n in rows if "sgd_lookahead" in n]
if len(ends) < 3:
    sys.exit("fewer than 3 steps in the trace")
a, b = ends[-3], ends[-2]
step = [r for r in rows if a < r[0] <= b or (r[0] <= b and r[1] > a and r[0] > a)]
print(f"step window {(b-a)/1e6:.3f} ms, {len(step)} dispatches")
byq = collections.defaultdict(list)
for r in step: byq[r[2]].append(r)
main = max(byq, key=lambda k: sum(e - s for s, e, _, _ in byq[k]))
for k, lst in byq.items():
    print(f"queue {k}{' (compute)' if k == main else ''}: {len(lst)} kernels, busy {sum(e-s for s,e,_,_ in lst)/1e6:.3f} ms")
lst = byq[main]
print("\ngaps >= 40 us on the compute queue:")
prev_end = a
for i, (s, e, _, n) in enumerate(lst):
    gap = s - prev_end
    if gap >= 40e3:
        others = [(os_, oe, oq, on) for (os_, oe, oq, on) in step if oq != main and oe > prev_end and os_ < s]
        busy = sum(min(oe, s) - max(os_, prev_end) for os_, oe, _, _ in others)
        names = collections.Counter(short(on)[:60] for _, _, _, on in others).most_common(3)
        before = short(lst[i-1][3])[:50] if i else "(step start)"
        print(f"  t={(prev_end-a)/1e6:7.3f} ms  gap {gap/1e3:7.1f} us  after [{before}] before [{short(n)[:50]}]  other queues busy {busy/1e3:7.1f} us: {names}")
    prev_end = max(prev_end, e)
print(f"  tail: compute queue ends at t={(prev_end-a)/1e6:.3f} ms of {(b-a)/1e6:.3f}")
# optional: rocpd_timeline.py trace.db T0_MS T1_MS -> every dispatch of the step in that window, all queues
if len(sys.argv) >= 4:
    t0, t1 = float(sys.argv[2]) * 1e6 + a, float(sys.argv[3]) * 1e6 + a
    print(f"\ndispatches between t={sys.argv[2]} and t={sys.argv[3]} ms (q = queue; * = compute queue):")
    for s, e, qq, n in step:
        if e >= t0 and s <= t1:
            print(f"  {(s-a)/1e6:8.3f} +{(e-s)/1e3:7.1f} us  q{qq}{'*' if qq == main else ' '} {short(n)[:110]}")
