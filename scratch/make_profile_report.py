#!/usr/bin/env python
"""Turn gpurun_out/<tag> (profiles/collect.sh output) into profiles/<tag>_*.txt summaries."""
import csv, sys, re, collections, os
tag = sys.argv[1]
src = os.path.join("gpurun_out", tag)
dst = "profiles"

def short(n): return re.sub(r"\(.*", "", n).replace("void ", "")

# ---- kernel trace stats (rocprofv3 --kernel-trace --stats)
rows = list(csv.DictReader(open(os.path.join(src, "trace", "t_kernel_stats.csv"))))
with open(os.path.join(dst, tag + "_kernel_stats.txt"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --forward-steps 2 --no-cpu-baseline --no-extra-modes (B=32, 384x1280; headline precision mode of bench.py)\n")
    f.write("# trace = 1 warm-up + 2 timed + 2 event-profiled train steps, then 3 warm-up + 2 timed + 3 event-profiled eval forwards\n")
    f.write("# (plus the one-off plan-build autotuning launches unless MONOCON_HIP_TUNE_CACHE pointed at a warm cache)\n")
    f.write("%-70s %7s %14s %12s %7s\n" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
    for r in rows[:40]:
        f.write("%-70s %7s %14s %12.0f %7s\n" % (short(r["Name"])[:70], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["Percentage"]))

# ---- per (kernel, grid) trace
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(os.path.join(src, "trace", "t_kernel_trace.csv"))):
    k = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)) if "Grid_Size_X" in r else (short(r["Kernel_Name"]), int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
    a = agg[k]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
with open(os.path.join(dst, tag + "_kernel_by_grid.txt"), "w") as f:
    tot = sum(v[1] for v in agg.values())
    f.write("%-60s %8s %6s %11s %9s %6s\n" % ("kernel", "blocks", "calls", "total_us", "avg_us", "pct"))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        f.write("%-60s %8d %6d %11.1f %9.1f %6.2f\n" % (k[0][:60], k[1], v[0], v[1], v[1] / v[0], 100 * v[1] / tot))

# ---- PMC
def pmc(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set); dur = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        k = (short(r["Kernel_Name"]), int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in cnt[k]:
            cnt[k].add(r["Dispatch_Id"]); dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return agg, cnt, dur
sq, cnt, dur = pmc(os.path.join(src, "pmc_sq", "p_counter_collection.csv"))
fe, cf, _ = pmc(os.path.join(src, "pmc_fetch", "p_counter_collection.csv"))
wr, cw, _ = pmc(os.path.join(src, "pmc_write", "p_counter_collection.csv"))
with open(os.path.join(dst, tag + "_pmc.txt"), "w") as f:
    f.write("# per dispatch averages.  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE/8 XCDs)\n")
    f.write("# fetch/write: FETCH_SIZE / WRITE_SIZE in KB as reported; hbm_rd_MB doubles FETCH_SIZE per the gfx950 note in\n")
    f.write("# MI355X_MICROARCH.md (FETCH_SIZE reports 1/2 of a wide coalesced read stream); WRITE_SIZE uncorrected.\n")
    f.write("%-46s %7s %5s %9s %9s %9s %10s %10s %9s\n" % ("kernel", "blocks", "n", "avg_us", "clk_GHz", "mfma_busy", "hbm_rd_MB", "hbm_wr_MB", "GB/s"))
    for k in sorted(sq, key=lambda k: -dur[k]):
        if not k[0].startswith("mc::"): continue
        n = len(cnt[k]); v = sq[k]
        gui = v["GRBM_GUI_ACTIVE"] / n / 8.0
        us = dur[k] / n
        busy = v["SQ_VALU_MFMA_BUSY_CYCLES"] / n / (1024.0 * gui) if gui else 0
        rd = 2.0 * fe[k]["FETCH_SIZE"] / max(len(cf[k]), 1) / 1024.0 if k in fe else float("nan")
        wrm = wr[k]["WRITE_SIZE"] / max(len(cw[k]), 1) / 1024.0 if k in wr else float("nan")
        f.write("%-46s %7d %5d %9.1f %9.2f %9.3f %10.1f %10.1f %9.0f\n" % (k[0][:46], k[1], n, us, gui / us / 1e3, busy, rd, wrm, (rd + wrm) / us * 1e3 if us else 0))
# ---- per-launch HBM traffic of the kernel families (bench.py reports it as roofline.traffic)
import json
fam = collections.defaultdict(lambda: [0.0, 0.0, 0, 0.0])
busy = collections.defaultdict(lambda: [0.0, 0.0, 0.0])     # family -> [sum MFMA-busy cycles, sum SIMD cycles available, sum clock*us]
for k in sq:
    f_ = re.sub(r"<.*", "", k[0])
    v = sq[k]
    gui = v["GRBM_GUI_ACTIVE"] / 8.0
    busy[f_][0] += v["SQ_VALU_MFMA_BUSY_CYCLES"]
    busy[f_][1] += 1024.0 * gui
    busy[f_][2] += dur[k]
    if k in fe and k in wr:
        n = min(len(cf[k]), len(cw[k]))
        fam[f_][0] += 2.0 * fe[k]["FETCH_SIZE"] * 1024.0 / max(len(cf[k]), 1) * n
        fam[f_][1] += wr[k]["WRITE_SIZE"] * 1024.0 / max(len(cw[k]), 1) * n
        fam[f_][2] += n
        fam[f_][3] += dur[k] / len(cnt[k]) * n
json.dump({"source": "profiles/%s_pmc.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE doubled per "
                     "MI355X_MICROARCH.md gfx950 note, WRITE_SIZE as reported)" % tag,
           "families": {f_: {"hbm_read_bytes_per_launch": v[0] / v[2], "hbm_write_bytes_per_launch": v[1] / v[2],
                             "launches_sampled": v[2], "avg_us_under_pmc": v[3] / v[2],
                             # time-weighted over all launches of the family in the SQ pass
                             "mfma_busy": busy[f_][0] / busy[f_][1] if busy[f_][1] else None,
                             "clock_ghz": busy[f_][1] / 1024.0 / busy[f_][2] / 1e3 if busy[f_][2] else None}
                        for f_, v in fam.items() if v[2] and f_.startswith("mc::")}},
          open(os.path.join(dst, "latest_traffic.json"), "w"), indent=1)
# ---- LDS counters (bank conflicts) of the kernels that use the LDS most
lp = os.path.join(src, "pmc_lds", "p_counter_collection.csv")
if os.path.exists(lp):
    lds, lcnt, _ = pmc(lp)
    with open(os.path.join(dst, tag + "_lds.txt"), "w") as f:
        f.write("# per dispatch averages of rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA (own pass).\n")
        f.write("# conflict_ratio = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE: share of the LDS array's cycles spent on bank-conflict replays\n")
        f.write("%-50s %7s %5s %13s %13s %9s %11s %11s\n" % ("kernel", "blocks", "n", "lds_conflict", "lds_active", "ratio", "insts_lds", "insts_mfma"))
        for k in sorted(lds, key=lambda k: -lds[k].get("SQ_LDS_IDX_ACTIVE", 0))[:24]:
            if not k[0].startswith("mc::"): continue
            v = lds[k]; n = len(lcnt[k])
            f.write("%-50s %7d %5d %13.4g %13.4g %9.3f %11.4g %11.4g\n" % (k[0][:50], k[1], n, v["SQ_LDS_BANK_CONFLICT"] / n, v["SQ_LDS_IDX_ACTIVE"] / n,
                    v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1), v["SQ_INSTS_LDS"] / n, v["SQ_INSTS_MFMA"] / n))
for extra in ("conv_shapes.txt", "mfma_peak.txt"):
    p = os.path.join(src, extra)
    if os.path.exists(p):
        open(os.path.join(dst, tag + "_" + extra), "w").write(open(p).read())
print(open(os.path.join(dst, tag + "_pmc.txt")).read())
