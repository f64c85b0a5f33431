#!/bin/bash
# kernel trace only (one warm run for the tune cache, one traced run): gpurun_out/<tag>/by_grid.txt
set -u
TAG=${1:-qt}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export MONOCON_HIP_TUNE_CACHE=/tmp/monocon_tune_cache.txt
BENCH="timeout 600 python $ROOT/bench.py --steps 2 --warmup 1 --forward-steps 2 --no-cpu-baseline --no-extra-modes $*"
$BENCH > "$OUT/bench_plain.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- $BENCH > "$OUT/trace.log" 2>&1
python - "$OUT" <<'PY'
import csv, sys, re, collections, os
out = sys.argv[1]
def short(n): return re.sub(r"\(.*", "", n).replace("void ", "")
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(os.path.join(out, "trace", "t_kernel_trace.csv"))):
    k = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)) if "Grid_Size_X" in r else (short(r["Kernel_Name"]), int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
    a = agg[k]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
with open(os.path.join(out, "by_grid.txt"), "w") as f:
    f.write("%-64s %8s %6s %11s %9s %6s\n" % ("kernel", "blocks", "calls", "total_us", "avg_us", "pct"))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:140]:
        f.write("%-64s %8d %6d %11.1f %9.1f %6.2f\n" % (k[0][:64], k[1], v[0], v[1], v[1] / v[0], 100 * v[1] / tot))
PY
rm -rf "$OUT/trace"
head -40 "$OUT/by_grid.txt"
