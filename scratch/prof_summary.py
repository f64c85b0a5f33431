#!/usr/bin/env python
"""Summarise a rocprofv3 results .db (kernel trace) per kernel name, optionally per grid size."""
import sqlite3, sys, re
db = sys.argv[1]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = c.execute("select name, start, end, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count from kernels" if 'grid_x' in cols else "select * from kernels limit 1").fetchall()
if 'grid_x' not in cols:
    print(cols); sys.exit()
agg = {}
for name, s, e, gx, wx, lds, vg, ag, sg in rows:
    short = re.sub(r"\(.*", "", name)
    short = re.sub(r"void ", "", short)
    k = (short, gx // max(wx,1), wx, lds, vg, ag)
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
print("%-92s %8s %6s %7s %5s %5s %6s %10s %9s %6s" % ("kernel", "blocks", "wg", "lds", "vgpr", "agpr", "calls", "total_us", "avg_us", "pct"))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 60]:
    print("%-92s %8d %6d %7d %5d %5d %6d %10.1f %9.1f %6.2f" % (k[0][:92], k[1], k[2], k[3], k[4], k[5], v[0], v[1], v[1] / v[0], 100 * v[1] / tot))
print("total kernel time us", tot)
