#!/bin/bash
# kernel-trace stats of the B=32 f16x2 train loop with scratch/ab/lib_old.so and lib_new.so (conv kernels only)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
L=$ROOT/monocon-pytorch_amd/hipmonocon/libmonocon_hip.so
cp $L /tmp/lib_orig.so
O=$ROOT/gpurun_out/ab; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export MONOCON_HIP_TUNE_CACHE=/tmp/monocon_tune_cache_ab.txt
for v in old new; do
  cp $ROOT/scratch/ab/lib_$v.so $L
  PREC=f16x2 python $ROOT/scratch/train_prof.py > /dev/null 2>&1
  rm -rf $O/tr_$v
  PREC=f16x2 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_$v -o t -- python $ROOT/scratch/train_prof.py > $O/tr_$v.log 2>&1
done
cp /tmp/lib_orig.so $L
python - <<PY
import csv, glob, re
res = {}
for v in ("old", "new"):
    f = glob.glob("$O/tr_%s/**/t_kernel_stats.csv" % v, recursive=True)[0]
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")
        res.setdefault(n, {})[v] = (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6)
out = open("$O/trace_ab.txt", "w")
tot = {"old": 0, "new": 0}
for n, d in sorted(res.items(), key=lambda kv: -kv[1].get("old", (0, 0))[1]):
    if "old" in d and "new" in d and ("conv" in n):
        out.write("%-64s calls %5d  old %9.2f ms  new %9.2f ms  %+6.1f %%\n" % (n[:64], d["old"][0], d["old"][1], d["new"][1], 100 * (d["new"][1] / d["old"][1] - 1)))
    for v in ("old", "new"):
        if v in d: tot[v] += d[v][1]
out.write("all kernels: old %.1f ms new %.1f ms\n" % (tot["old"], tot["new"]))
PY
cat $O/trace_ab.txt
