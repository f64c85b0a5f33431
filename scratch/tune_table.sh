#!/bin/bash
# which shape ids the autotuner picks for the conv_wres-eligible layers, tuned on zeros (default) and on noise
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/tune
for noise in 0 1; do
  rm -f /tmp/tc_$noise.txt
  MONOCON_HIP_TUNE_NOISE=$noise MONOCON_HIP_TUNE_CACHE=/tmp/tc_$noise.txt python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra-modes 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('noise $noise', d['value'], d['ms_per_step'], 'fwd', d['forward_only']['ms'])"
  cp /tmp/tc_$noise.txt $ROOT/gpurun_out/tune/tc_$noise.txt
done
