"""Print per-kernel PMC counter averages from a rocprofv3 rocpd database."""
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
filt = sys.argv[2] if len(sys.argv) > 2 else "contraction"
cols = [r[1] for r in cur.execute("pragma table_info(pmc_events)")]
rows = cur.execute("select * from pmc_events").fetchall()
print(cols)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
ci = {c: i for i, c in enumerate(cols)}
for r in rows:
    name = str(r[ci.get("name", ci.get("kernel_name", 0))]) if ("name" in ci or "kernel_name" in ci) else "?"
    agg[name][r[ci["counter_name"]] if "counter_name" in ci else r[ci["pmc_name"]]].append(r[ci["value"]] if "value" in ci else r[ci["counter_value"]])
for k, d in agg.items():
    if filt in k:
        cut = k.find(">(")                                              # template arguments in full, no parameter list
        print(k[:cut + 1] if cut > 0 else k[:200])
        for c, v in sorted(d.items()):
            print(f"   {c:32s} n={len(v):3d} avg={sum(v)/len(v):.4g}")
