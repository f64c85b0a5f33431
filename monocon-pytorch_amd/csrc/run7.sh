#!/bin/bash
cd $GRAFT_REPO_ROOT/scratch/wg16
O=$GRAFT_REPO_ROOT/gpurun_out/wg16; mkdir -p $O; : > $O/t7.txt
echo "---- 8-wave tiles (MONOCON_HIP_WGRAD_PIPE=2)" >> $O/t7.txt; MONOCON_HIP_WGRAD_PIPE=2 WG_PIPE=1 timeout 100 ./bench_wg 10 >> $O/t7.txt 2>&1
echo "---- 4-wave shared tile (MONOCON_HIP_WGRAD_PIPE=1)" >> $O/t7.txt; MONOCON_HIP_WGRAD_PIPE=1 WG_PIPE=1 timeout 100 ./bench_wg 10 >> $O/t7.txt 2>&1
echo "---- 4-wave shared tile, 512 blocks" >> $O/t7.txt; MONOCON_HIP_WGRAD_PIPE_BLOCKS=512 MONOCON_HIP_WGRAD_PIPE=1 WG_PIPE=1 timeout 100 ./bench_wg 10 >> $O/t7.txt 2>&1
cat $O/t7.txt
