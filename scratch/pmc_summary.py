#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc counter_collection.csv per (kernel, grid) -> mean per dispatch."""
import csv, sys, re, collections
path = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
dur = collections.defaultdict(float)
for r in csv.DictReader(open(path)):
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    k = (name, int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in cnt[k]:
        cnt[k].add(r["Dispatch_Id"])
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
names = sorted({c for v in agg.values() for c in v})
print("kernel,blocks,dispatches,avg_us," + ",".join(names))
for k, v in sorted(agg.items(), key=lambda kv: -dur[kv[0]]):
    n = len(cnt[k])
    if not k[0].startswith("mc::"): continue
    print("%s,%d,%d,%.1f," % (k[0], k[1], n, dur[k] / n) + ",".join("%.4g" % (v.get(c, 0) / n) for c in names))
