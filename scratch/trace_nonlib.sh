#!/bin/bash
# which kernels of a train step do NOT come from libmonocon_hip (torch fills / copies / arithmetic of the loss dict, optimizer glue)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/nonlib; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MONOCON_HIP_TUNE_CACHE=/tmp/monocon_tune_cache_f16x2.txt
PREC=f16x2 python $ROOT/scratch/train_prof.py > $OUT/plain.log 2>&1
PREC=f16x2 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python $ROOT/scratch/train_prof.py > $OUT/trace.log 2>&1
python - <<PY
import csv, glob, re, collections
f = glob.glob("$OUT/trace/**/t_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = 9.0
agg = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"]
    if "mc::" in name: continue
    short = re.sub(r"\(.*", "", name).replace("void ", "")[:90]
    k = (short, r["Grid_Size"], r["Workgroup_Size"])
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
out = open("$ROOT/gpurun_out/nonlib.txt", "w")
tot = 0
for k, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    out.write("%-90s grid %9s wg %5s  %6.1f /step  %8.1f us avg  %7.3f ms/step\n" % (k[0], k[1], k[2], c / n, us / c, us / n / 1e3))
    tot += us / n / 1e3
out.write("total non-library kernel time %.3f ms/step\n" % tot)
PY
cat $ROOT/gpurun_out/nonlib.txt | head -40
