#!/bin/bash
# lib_step_ab.sh at both widths (1280: the benchmark's; 1248: KITTI's after Pad(32)) + the eval forward
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
L=$ROOT/monocon-pytorch_amd/hipmonocon/libmonocon_hip.so
cp $L /tmp/lib_orig.so
for rep in 1 2; do for w in 1280 1248; do for v in old new; do
  cp $ROOT/scratch/ab/lib_$v.so $L
  WIDTH=$w TAG="lib_$v W$w" python $ROOT/scratch/time_step.py f16x2 10 2>/dev/null
done; done; done
for w in 1280 1248; do for v in old new; do
  cp $ROOT/scratch/ab/lib_$v.so $L
  echo "eval forward lib_$v W$w: $(WIDTH=$w PREC=f16x2 python $ROOT/scratch/fwd_ops.py 2>/dev/null | tail -1)"
done; done
cp /tmp/lib_orig.so $L
