#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
L=$ROOT/monocon-pytorch_amd/hipmonocon/libmonocon_hip.so
cp $L /tmp/lib_orig.so
for rep in 1 2; do for v in old new; do cp $ROOT/scratch/ab/lib_$v.so $L; for p in "" bf16x3; do echo "$v prec=$p $(NSTEP=30 PREC=$p python $ROOT/scratch/train_loop.py 2>&1 | grep '^ms/step')"; done; done; done
cp /tmp/lib_orig.so $L
