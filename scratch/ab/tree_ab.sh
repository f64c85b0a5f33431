#!/bin/bash
# one-session A/B of two TREES (their own Python + library): scratch/ab/r4tree (not committed: a checkout of the round-4
# final commit, built) against the working tree, headline bench, alternating
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2 3; do for t in r4 r5; do
  if [ $t = r4 ]; then B=$ROOT/scratch/ab/r4tree/bench.py; else B=$ROOT/bench.py; fi
  python $B --steps 10 --warmup 3 --no-cpu-baseline --no-extra-modes --full-json /tmp/tree_ab_$t.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$t', d['value'], d['ms_per_step'], 'conv', r['conv_ms'], 'wgrad', r['wgrad_ms'], 'other', r['other_ms'], 'fwd', d['forward_only']['ms'], 'fwd_conv', d['forward_only']['conv_ms'])"
done; done
