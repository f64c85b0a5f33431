#!/bin/bash
# train / eval-forward numbers of bench.py at other per-GPU batch sizes
for b in 1 8 16 64; do
  python bench.py --batch $b --steps 5 --warmup 2 --forward-steps 10 --realistic-steps 0 --no-cpu-baseline --no-extra-modes > /tmp/sweep_$b.json 2> /tmp/sweep_$b.err
  python - $b <<'PY'
import json, sys
b = sys.argv[1]
try:
    d = json.loads(open('/tmp/sweep_%s.json' % b).read().strip().splitlines()[-1])
    f = d["forward_only"]
    print("B", b, "train ms", d["ms_per_step"], "img/s", d["value"], "ws GB", d.get("workspace_gb"), "fwd ms", f["ms_per_step"], "fwd img/s", f["images_per_sec"])
except Exception as e:
    print("B", b, "failed", e, open('/tmp/sweep_%s.err' % b).read()[-400:])
PY
done
