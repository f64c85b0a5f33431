#!/bin/bash
# in-situ A/B: B=32 train step with the pipelined weight-gradient kernel on / off (one session)
cd $GRAFT_REPO_ROOT
O=gpurun_out/wg16; mkdir -p $O; : > $O/ab_step.txt
for r in 1 2; do
  for p in 0 1; do
    TAG=pipe$p MONOCON_HIP_WGRAD_PIPE=$p timeout 300 python scratch/time_step.py f16x2 10 2>&1 | tail -1 >> $O/ab_step.txt
  done
done
cat $O/ab_step.txt
