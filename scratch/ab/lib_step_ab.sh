#!/bin/bash
# one-session A/B of two builds of libmonocon_hip.so (scratch/ab/lib_old.so, lib_new.so) on scratch/time_step.py (step + bucket profile)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
L=$ROOT/monocon-pytorch_amd/hipmonocon/libmonocon_hip.so
cp $L /tmp/lib_orig.so
for rep in 1 2 3; do for v in old new; do
  cp $ROOT/scratch/ab/lib_$v.so $L
  TAG=lib_$v python $ROOT/scratch/time_step.py f16x2 10 2>/dev/null
done; done
cp /tmp/lib_orig.so $L
