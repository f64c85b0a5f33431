#!/bin/bash
# A/B of an environment switch in one session: usage env_ab.sh VAR val1 val2 ...
VAR=$1; shift
for v in "$@" "$@"; do
env $VAR=$v python bench.py --steps 6 --warmup 2 --forward-steps 0 --realistic-steps 0 --no-cpu-baseline --no-extra-modes 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$VAR=$v', 'step', d['ms_per_step'], 'conv', r['conv_ms_per_step'], 'wgrad', r['wgrad']['ms_per_step'], 'other', r['other_ms_per_step'])"
done
