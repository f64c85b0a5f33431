#!/bin/bash
# usage: env_step_ab.sh "ENV1=a" "ENV1=b" ...  -- B=32 f16x2 step time per environment setting, two rounds, one session
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/ab; mkdir -p $O; : > $O/env_step_ab.txt
for rep in 1 2; do for e in "$@"; do env $e TAG="$e" timeout 300 python $ROOT/scratch/time_step.py f16x2 10 2>&1 | tail -1 >> $O/env_step_ab.txt; done; done
cat $O/env_step_ab.txt
