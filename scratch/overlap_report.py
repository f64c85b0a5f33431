#!/usr/bin/env python
"""How much of the main stream's kernel time runs beside a kernel of the weight-gradient stream (and vice versa):
per kernel family, time alone vs time overlapped, from a rocprofv3 --kernel-trace csv of scratch/train_prof.py."""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_lo = int(rows[0]["Start_Timestamp"]); t_hi = int(rows[-1]["End_Timestamp"])
t0 = t_lo + (t_hi - t_lo) * 55 // 100          # steady state: the last ~4 steps
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], re.sub(r"<.*|\(.*", "", r["Kernel_Name"]).replace("void ", "")) for r in rows if int(r["Start_Timestamp"]) >= t0]
queues = collections.Counter(q for _, _, q, _ in ks)
main_q = queues.most_common(1)[0][0]
side = sorted((s, e) for s, e, q, n in ks if q != main_q)
import bisect
starts = [s for s, _ in side]
def overlapped(s, e):
    tot = 0
    i = max(0, bisect.bisect_left(starts, s) - 2)
    while i < len(side) and side[i][0] < e:
        a, b = max(s, side[i][0]), min(e, side[i][1])
        if b > a: tot += b - a
        i += 1
    return min(tot, e - s)
fam = collections.defaultdict(lambda: [0, 0.0, 0.0])
for s, e, q, n in ks:
    if q != main_q: continue
    o = overlapped(s, e)
    f = fam[n]; f[0] += 1; f[1] += (e - s) / 1e6; f[2] += o / 1e6
print("queues:", dict(queues), "main =", main_q)
print("%-40s %6s %10s %12s %6s" % ("main-stream kernel", "calls", "total ms", "beside side", "%"))
T = O = 0
for n, (c, t, o) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:28]:
    print("%-40s %6d %10.2f %12.2f %5.0f%%" % (n[:40], c, t, o, 100 * o / t if t else 0)); T += t; O += o
print("main total %.1f ms, of which beside a side-stream kernel %.1f ms; side total %.1f ms" % (T, O, sum(e - s for s, e in side) / 1e6))
