#!/bin/bash
# usage: bash scratch/trace_fwd.sh <tag> [PREC]  -> gpurun_out/<tag>_fwd_kernels.txt : kernels of 9 eval forwards at B=32
TAG=${1:-tf}; PREC=${2:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PREC=$PREC python $ROOT/scratch/fwd_prof.py > $OUT/plain.log 2>&1
PREC=$PREC rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/scratch/fwd_prof.py > $OUT/trace.log 2>&1
python - <<PY
import csv, re, glob
f = glob.glob("$OUT/trace/**/t_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
n = 9.0
tot = 0
out = open("$ROOT/gpurun_out/${TAG}_fwd_kernels.txt", "w")
out.write(open("$OUT/plain.log").read())
out.write("%-80s %8s %10s %10s\n" % ("kernel", "calls/fw", "ms/fwd", "avg_us"))
for r in rows:
    name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")
    ms = float(r["TotalDurationNs"]) / 1e6 / n
    tot += ms
    out.write("%-80s %8.1f %10.3f %10.1f\n" % (name[:80], float(r["Calls"]) / n, ms, float(r["AverageNs"]) / 1e3))
out.write("total kernel ms/fwd %.2f\n" % tot)
PY
