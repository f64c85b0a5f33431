#!/usr/bin/env python
"""per train step: time with only the main queue busy, only the side queue busy, both, none (rocprofv3 kernel-trace csv)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
q = collections.Counter(r["Queue_Id"] for r in rows); main_q = q.most_common(1)[0][0]
# step boundaries: stem_f16_kernel launches
starts = [int(r["Start_Timestamp"]) for r in rows if "stem_f16_kernel" in r["Kernel_Name"]]
ev = []
for r in rows:
    m = r["Queue_Id"] == main_q
    ev.append((int(r["Start_Timestamp"]), 1, m)); ev.append((int(r["End_Timestamp"]), -1, m))
ev.sort()
for si in range(len(starts) - 4, len(starts) - 1):
    lo, hi = starts[si], starts[si + 1]
    nm = ns = 0; last = lo; acc = {"main": 0, "side": 0, "both": 0, "idle": 0}
    for t, d, m in ev:
        if t > hi: break
        if t >= lo:
            k = "both" if nm and ns else ("main" if nm else ("side" if ns else "idle"))
            acc[k] += t - last; last = t
        if m: nm += d
        else: ns += d
        if t < lo: last = lo
    print("step %d: %.2f ms  main-only %.2f  side-only %.2f  both %.2f  idle %.2f" % (si, (hi - lo) / 1e6, acc["main"] / 1e6, acc["side"] / 1e6, acc["both"] / 1e6, acc["idle"] / 1e6))
