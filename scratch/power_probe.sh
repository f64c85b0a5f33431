#!/bin/bash
# sample clock / power while the train step runs (is the step power-limited?)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for p in "" bf16x3 bf16; do
  ( TB=32 PREC=$p python $ROOT/scratch/train_loop.py > /tmp/loop_$p.log 2>&1 ) &
  PID=$!
  sleep 25
  for i in 1 2 3 4 5 6; do
    rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "sclk|mclk|Socket Power|Average Graphics|junction|fclk" | tr -s ' ' | tr '\n' ';'
    echo
    sleep 0.5
  done
  wait $PID
  echo "prec=$p $(grep ms/step /tmp/loop_$p.log)"
done
rocm-smi --showmaxpower 2>/dev/null | grep -i power | head -3
