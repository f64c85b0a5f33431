#!/bin/bash
# A/B of HBM read traffic (rocprofv3 --pmc FETCH_SIZE) of the conv / wgrad kernels for two builds of the library.
# usage (under gpurun, from the repo root): bash scratch/traffic_ab.sh <libA.so> <libB.so>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
L=$ROOT/monocon-pytorch_amd/hipmonocon/libmonocon_hip.so
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  cp "$ROOT/$v" "$L"
  export MONOCON_HIP_TUNE_CACHE=/tmp/tune_$(basename $v).txt
  timeout 300 python $ROOT/scratch/train_prof.py > /dev/null 2>&1 < /dev/null        # warm the tune cache
  rm -rf /tmp/pm_$$; timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pm_$$ -o p -- python $ROOT/scratch/train_prof.py > /dev/null 2>&1 < /dev/null
  python - "$v" $(find /tmp/pm_$$ -name "*counter_collection.csv" | head -1) <<'PY'
import csv, sys, re, collections
tag, path = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(path)):
    if r["Counter_Name"] != "FETCH_SIZE": continue
    fam = re.sub(r"<.*", "", re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", ""))
    a = agg[fam]; a[0] += 1; a[1] += float(r["Counter_Value"])
for fam in ("mc::conv_mfma_kernel", "mc::wgrad_mfma_kernel", "mc::conv_small_kernel"):
    n, kb = agg.get(fam, (0, 0.0))
    if n: print("%-24s %-26s launches %5d  hbm read %.1f MB / launch (FETCH_SIZE x2)" % (tag, fam, n, kb * 2 / n / 1e3))
PY
done
