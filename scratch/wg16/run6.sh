#!/bin/bash
cd $GRAFT_REPO_ROOT/scratch/wg16
O=$GRAFT_REPO_ROOT/gpurun_out/wg16; mkdir -p $O; : > $O/t6.txt
echo "---- old PD=1" >> $O/t6.txt; timeout 100 ./bench_wg 10 >> $O/t6.txt 2>&1
echo "---- old PD=2" >> $O/t6.txt; timeout 100 ./bench_wg_pd2 10 >> $O/t6.txt 2>&1
cat $O/t6.txt
