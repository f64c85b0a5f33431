#!/bin/bash
# A/B of the gradient-map pool in one session: step time and workspace
for v in "0 0" "1 0" "1 1" "1 2" "0 0" "1 0" "1 1" "1 2"; do
set -- $v
MONOCON_HIP_GRAD_POOL=$1 MONOCON_HIP_GRAD_POOL_COOL=$2 python bench.py --steps 6 --warmup 2 --forward-steps 0 --realistic-steps 0 --no-cpu-baseline --no-extra-modes 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('pool', '$v', 'step', d['ms_per_step'], 'ws_gb', d.get('workspace_gb'), 'conv', r['conv_ms_per_step'], 'wgrad', r['wgrad']['ms_per_step'], 'other', r['other_ms_per_step'])"
done
