#!/usr/bin/env python
"""Kernel neighbours of a pattern in a rocprofv3 kernel-trace .db: what launches around <pattern>?"""
import sqlite3, sys, re, collections
db, pat = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = c.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
short = lambda n: re.sub(r"\(.*", "", n).replace("void ", "")[:60]
ctx = collections.Counter()
for i, r in enumerate(rows):
    if pat in r[0]:
        prev = short(rows[i - 1][0]) if i else "-"
        nxt = short(rows[i + 1][0]) if i + 1 < len(rows) else "-"
        ctx[(prev, r[3] // max(r[4], 1), nxt)] += 1
for (p, g, n), k in ctx.most_common(40):
    print("%4d  %-60s -> [%6d blocks] -> %s" % (k, p, g, n))
