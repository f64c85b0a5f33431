#!/bin/bash
# which memsets run inside the train step, and how long they take (kernel trace, one traced run)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/fill
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export MONOCON_HIP_TUNE_CACHE=/tmp/monocon_tune_cache.txt
BENCH="timeout 600 python $ROOT/bench.py --steps 2 --warmup 1 --forward-steps 0 --no-cpu-baseline --no-extra-modes"
$BENCH > "$OUT/plain.log" 2>&1
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o t -- $BENCH > "$OUT/trace.log" 2>&1
python - "$OUT" <<'PY'
import csv, sys, os, re
out = sys.argv[1]
rows = list(csv.DictReader(open(os.path.join(out, "trace", "t_kernel_trace.csv"))))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n): return re.sub(r"\(.*", "", n).replace("void ", "")[:50]
# the last 1/3 of the run is the timed steps; print every fill with its neighbours
n = len(rows)
with open(os.path.join(out, "fills.txt"), "w") as f:
    for i, r in enumerate(rows):
        if "fillBuffer" in r["Kernel_Name"] or "elementwise" in r["Kernel_Name"]:
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            prev = short(rows[i - 1]["Kernel_Name"]) if i else ""
            nxt = short(rows[i + 1]["Kernel_Name"]) if i + 1 < n else ""
            f.write("%6d q%s %8.1f us  grid %s  %s | after %s | before %s\n" % (i, r.get("Queue_Id", "?"), d, r.get("Grid_Size_X", r.get("Grid_Size", "?")), short(r["Kernel_Name"]), prev, nxt))
PY
rm -rf "$OUT/trace"
tail -150 "$OUT/fills.txt"
