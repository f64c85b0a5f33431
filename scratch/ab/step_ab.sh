#!/bin/bash
# A/B of scratch/ab/lib_old.so vs lib_new.so on the B=32 f16x2 train step (scratch/time_step.py), one session
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
L=$ROOT/monocon-pytorch_amd/hipmonocon/libmonocon_hip.so
cp $L /tmp/lib_orig.so
O=$ROOT/gpurun_out/ab; mkdir -p $O; : > $O/step_ab.txt
for rep in 1 2; do for v in old new; do cp $ROOT/scratch/ab/lib_$v.so $L; TAG=$v timeout 300 python $ROOT/scratch/time_step.py f16x2 10 2>&1 | tail -1 >> $O/step_ab.txt; done; done
cp /tmp/lib_orig.so $L
cat $O/step_ab.txt
