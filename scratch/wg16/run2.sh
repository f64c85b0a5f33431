#!/bin/bash
cd $GRAFT_REPO_ROOT/scratch/wg16
O=$GRAFT_REPO_ROOT/gpurun_out/wg16
mkdir -p $O
: > $O/delay.txt
for d in 0 2 4 8 16; do
  echo "---- WG_MODE=1 WG_DELAY=$d (x512 cycles)" >> $O/delay.txt
  WG_MODE=1 WG_DELAY=$d timeout 100 ./bench_wg_delay 10 >> $O/delay.txt 2>&1
done
cat $O/delay.txt
