#!/bin/bash
cd $GRAFT_REPO_ROOT/scratch/wg16
O=$GRAFT_REPO_ROOT/gpurun_out/wg16
mkdir -p $O
: > $O/t3.txt
for v in "" _nostore _nostore_nofetch _fetchonly; do
  for b in 512 256; do
    echo "---- bench_wg$v WG_BLOCKS=$b" >> $O/t3.txt
    WG_BLOCKS=$b timeout 200 ./bench_wg$v 10 >> $O/t3.txt 2>&1
  done
done
cat $O/t3.txt
