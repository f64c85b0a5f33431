#!/bin/bash
# eval-forward and train-step numbers with library variants under scratch/exp/
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
L=$ROOT/monocon-pytorch_amd/hipmonocon/libmonocon_hip.so
cp $L /tmp/lib_orig.so
for v in "$@"; do
  cp $ROOT/scratch/exp/lib_$v.so $L
  python $ROOT/bench.py --steps 6 --warmup 2 --forward-steps 8 --realistic-steps 0 --no-cpu-baseline --no-extra-modes 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; f=d['forward_only']; print('$v', 'step', d['ms_per_step'], 'conv', r['conv_ms_per_step'], 'wgrad', r['wgrad']['ms_per_step'], 'other', r['other_ms_per_step'], 'fwd_ms', f.get('ms_per_batch', f.get('ms_per_step')), 'fwd img/s', f['images_per_sec'])"
done
cp /tmp/lib_orig.so $L
