#!/usr/bin/env python
"""per-dispatch summary of a rocprofv3 --pmc csv of the harness: duration, clock (GRBM_GUI_ACTIVE / 8 XCDs / duration),
MFMA busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / cycles)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.OrderedDict()
for r in rows:
    k = r["Dispatch_Id"]
    e = d.setdefault(k, {"name": r["Kernel_Name"], "grid": r["Grid_Size"], "t": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
    e[r["Counter_Name"]] = float(r["Counter_Value"])
last = collections.OrderedDict()
for k, e in d.items():
    if "wgrad" not in e["name"]:
        continue
    last[(e["name"][:48], e["grid"], len([1 for kk in last if kk[0] == e["name"][:48] and kk[1] == e["grid"]]))] = e
seen = {}
for k, e in d.items():
    if "wgrad" not in e["name"]:
        continue
    seen.setdefault((e["name"][9:50], e["grid"]), []).append(e)
for (n, g), es in seen.items():
    e = es[-1]
    cyc = e.get("GRBM_GUI_ACTIVE", 0) / 8
    out = "%-42s grid %7s  %7.1f us" % (n, g, e["t"])
    if cyc:
        out += "  clock %.2f GHz" % (cyc / e["t"] / 1e3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e:
            out += "  mfma_busy %.2f" % (e["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc)
        for c in ("SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INST_CYCLES_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
            if c in e:
                out += "  %s %.2f" % (c[3:].lower(), e[c] / (cyc * (1024 if c != "SQ_BUSY_CYCLES" else 32)))
    print(out)
