#!/bin/bash
# one-session A/B of two builds of libmonocon_hip.so (scratch/ab/lib_old.so, lib_new.so) on the headline bench
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
L=$ROOT/monocon-pytorch_amd/hipmonocon/libmonocon_hip.so
cp $L /tmp/lib_orig.so
for rep in 1 2; do for v in old new; do
  cp $ROOT/scratch/ab/lib_$v.so $L
  python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-modes 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('lib_$v', d['value'], d['ms_per_step'], 'conv', r['conv_ms'], 'wgrad', r['wgrad_ms'], 'other', r['other_ms'], 'fwd', d['forward_only']['ms'], 'fwd_conv', d['forward_only']['conv_ms'])"
done; done
cp /tmp/lib_orig.so $L
