#!/bin/bash
# step time with kernel families turned into no-ops (WRONG numerics): upper bounds on what removing them can gain
# needs a library built with the debug hooks: make -C monocon-pytorch_amd/csrc clean && make -C monocon-pytorch_amd/csrc EXTRA=-DMC_DEBUG_HOOKS
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export MONOCON_HIP_TUNE_CACHE=/tmp/tune_$1.txt
for skip in none fold,fin,bfin aact abwd cred aact,abwd,cred fold,fin,bfin,aact,abwd,cred; do
  for dual in 1 0; do
    r=$(MONOCON_HIP_DEBUG_SKIP=$skip MONOCON_HIP_DUAL_STREAM=$dual PREC=$1 python $ROOT/scratch/train_prof.py 2>/dev/null | grep "ms/step" | head -1)
    echo "prec=${1:-fp32} skip=$skip dual=$dual $r"
  done
done
