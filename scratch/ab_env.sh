#!/bin/bash
# one-session A/B of environment settings on the headline bench: bash scratch/ab_env.sh "A=1" "A=0" ...  (each run twice, interleaved)
for rep in 1 2; do
for v in "$@"; do
  env $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-modes 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-28s' % '$v', d['value'], d['ms_per_step'], 'conv', r['conv_ms'], 'wgrad', r['wgrad_ms'], 'other', r['other_ms'], 'fwd', d['forward_only']['ms'], 'fwd_conv', d['forward_only']['conv_ms'])"
done; done
