#!/usr/bin/env python
"""the kernels of the last `ms` milliseconds of a train step on both queues, and of its first `ms` (rocprofv3 kernel-trace csv)"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ms = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
q = collections.Counter(r["Queue_Id"] for r in rows); main_q = q.most_common(1)[0][0]
starts = [int(r["Start_Timestamp"]) for r in rows if "stem_f16_kernel" in r["Kernel_Name"]]
lo, hi = starts[-3], starts[-2]
def short(n): return re.sub(r"\(.*", "", n).replace("void ", "").replace("mc::", "")[:44]
step = [r for r in rows if lo <= int(r["Start_Timestamp"]) < hi]
# the step's own end: the last kernel before the next stem that is not a torch / pack kernel of the next step
print("step %.2f ms, %d kernels" % ((hi - lo) / 1e6, len(step)))
bwd0 = next(int(r["Start_Timestamp"]) for r in step if "focal_grad" in r["Kernel_Name"] or "head_bwd" in r["Kernel_Name"])
print("backward starts at %.2f ms" % ((bwd0 - lo) / 1e6))
for title, a, b in (("first %.1f ms of the backward" % ms, bwd0, bwd0 + ms * 1e6), ("last %.1f ms" % ms, hi - ms * 1e6, hi)):
    print("----", title)
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if e < a or s > b: continue
        print("%s %8.3f .. %8.3f  (%7.1f us)  %s" % ("M" if r["Queue_Id"] == main_q else "    S", (s - lo) / 1e6, (e - lo) / 1e6, (e - s) / 1e3, short(r["Kernel_Name"])))
