"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / %.
Usage: python tools/rocpd_stats.py results.db [top_n] [--tail-ms X]  -> text summary on stdout.
--tail-ms X: only the dispatches that start in the last X ms of the trace (the steady state of a run whose beginning is set-up:
MIOpen's solver search in the stock-PyTorch baseline)."""
import re
import sqlite3
import sys


def short(name):
    name = name.replace("vtxg::", "").replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"unsigned short", "bf16", name)
    return name[:150]


def main():
    db = sys.argv[1]
    argv = list(sys.argv)
    tail_ms = None
    if "--tail-ms" in argv:
        i = argv.index("--tail-ms")
        tail_ms = float(argv[i + 1])
        del argv[i:i + 2]
    top = int(argv[2]) if len(argv) > 2 else 40
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {namecol}, start, end from kernels").fetchall()
    agg = {}
    t0 = min(r[1] for r in rows); t1 = max(r[2] for r in rows)
    if tail_ms is not None:
        rows = [r for r in rows if r[1] >= t1 - tail_ms * 1e6]
        t0 = min(r[1] for r in rows)
    for n, s, e in rows:
        a = agg.setdefault(short(n), [0, 0])
        a[0] += 1; a[1] += e - s
    total = sum(a[1] for a in agg.values())
    print(f"# kernels: {len(rows)} dispatches, {len(agg)} distinct; busy {total/1e6:.2f} ms over a {(t1-t0)/1e6:.2f} ms window")
    print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'pct':>6}  kernel")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{c:7d} {t/1e6:10.3f} {t/c/1e3:10.1f} {100*t/total:6.2f}  {n}")


if __name__ == "__main__":
    main()
