"""Per (kernel, grid) averages of one PMC counter from a rocprofv3 rocpd database: splits a kernel class into the SHAPES it
was launched on (the grid identifies the layer), which the per-name averages of tools/pmc_dump.py merge.
    python tools/pmc_by_grid.py results.db <substring of the kernel name> [counter]"""
import collections
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
filt = sys.argv[2]
kcols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
grid = {}
for d, gx, gy, wx in cur.execute("select dispatch_id, grid_x, grid_y, workgroup_x from kernels"):
    grid[d] = (gx // max(wx, 1), gy)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for name, d, counter, value, t in cur.execute("select name, dispatch_id, counter_name, counter_value, duration from pmc_events"):
    if filt in name:
        cut = name.find(">(")
        key = (name[:cut + 1] if cut > 0 else name[:160], grid.get(d))
        agg[key][counter].append(value)
        dur[key].append(t)
for key in sorted(agg, key=lambda k: (k[0], -(k[1] or (0, 0))[0])):
    d = agg[key]
    print(f"{key[0][:150]}  grid {key[1]}")
    for c, v in sorted(d.items()):
        print(f"   {c:16s} n={len(v):3d} avg={sum(v) / len(v):12.1f}   avg duration {sum(dur[key]) / len(dur[key]) / 1e3:8.1f} us")
