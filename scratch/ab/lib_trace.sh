#!/bin/bash
# kernel trace of the headline bench with two builds of the library: gpurun_out/lt_old/by_grid.txt, gpurun_out/lt_new/by_grid.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
L=$ROOT/monocon-pytorch_amd/hipmonocon/libmonocon_hip.so
cp $L /tmp/lib_orig.so
for v in old new; do
  cp $ROOT/scratch/ab/lib_$v.so $L
  rm -f /tmp/monocon_tune_cache.txt
  bash $ROOT/scratch/quick_trace.sh lt_$v > /dev/null 2>&1
done
cp /tmp/lib_orig.so $L
