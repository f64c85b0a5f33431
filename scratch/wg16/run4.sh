#!/bin/bash
cd $GRAFT_REPO_ROOT/scratch/wg16
O=$GRAFT_REPO_ROOT/gpurun_out/wg16
mkdir -p $O
: > $O/t4.txt
echo "---- old" >> $O/t4.txt; timeout 100 ./bench_wg 10 >> $O/t4.txt 2>&1
echo "---- pipe" >> $O/t4.txt; WG_PIPE=1 timeout 100 ./bench_wg 10 >> $O/t4.txt 2>&1
echo "---- pipe 512 blocks" >> $O/t4.txt; MONOCON_HIP_WGRAD_PIPE_BLOCKS=512 WG_PIPE=1 timeout 100 ./bench_wg 10 >> $O/t4.txt 2>&1
cat $O/t4.txt
