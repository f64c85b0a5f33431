#!/usr/bin/env python
"""Per-train-step kernel table from a rocprofv3 kernel-trace .db of scratch/train_prof.py (NSTEP steps
incl. warm-up): kernels launched fewer than NSTEP times (plan-build autotuning) are dropped."""
import sqlite3, sys, re
db, nstep = sys.argv[1], int(sys.argv[2])
c = sqlite3.connect(db)
rows = c.execute("select name, start, end, grid_x, workgroup_x from kernels").fetchall()
agg = {}
for name, s, e, gx, wx in rows:
    short = re.sub(r"\(.*", "", name).replace("void ", "")
    k = (short, gx // max(wx, 1))
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e3
fam = {}
tot = 0.0
lines = []
for (n, blocks), (calls, us) in agg.items():
    if calls < nstep:
        continue
    per_step = calls // nstep
    ms = us / calls * per_step / 1e3
    tot += ms
    f = re.sub(r"<.*", "", n)
    fam[f] = fam.get(f, 0.0) + ms
    lines.append((ms, "%-64s %8d blocks %3d/step %9.1f us avg %8.3f ms/step" % (n[:64], blocks, per_step, us / calls, ms)))
for ms, l in sorted(lines, reverse=True)[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    print(l)
print("---- families (ms/step)")
for f, ms in sorted(fam.items(), key=lambda kv: -kv[1])[:30]:
    print("%-48s %8.3f" % (f, ms))
print("total kernel ms/step %.2f" % tot)
# ---- GPU busy fraction over the last ~60 % of the trace (steady state): union of kernel intervals / wall
iv = sorted((s, e) for _, s, e, _, _ in rows)
t0 = iv[0][0] + (iv[-1][1] - iv[0][0]) * 2 // 5
iv = [(max(s, t0), e) for s, e in iv if e > t0]
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = iv[-1][1] - t0
gaps = sorted(((iv[i + 1][0] - max(x[1] for x in iv[:i + 1][-8:])) for i in range(len(iv) - 1)), reverse=True)
print("steady-state wall %.1f ms, GPU busy (union) %.1f ms = %.1f %%" % (wall / 1e6, busy / 1e6, 100.0 * busy / wall))
