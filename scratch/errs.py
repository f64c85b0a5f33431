import sys, os
sys.path.insert(0, 'monocon-pytorch_amd'); sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from conftest import load_golden, rel_err
from hipmonocon import synth
from hipmonocon.engine import Engine
stats = load_golden("bn_calib_seed7.npz")
sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
e = Engine(); st = {k: v.cuda() for k, v in sd.items()}; e.bind_state(st)
g = load_golden("fwd_full_eval.npz")
img = synth.make_batch(9, 2, 384, 1280, with_labels=False)["img"].cuda()
p = e.forward_infer(img)
for k, v in p.items():
    s = v.cpu().reshape(-1)[::97]
    print("%-26s vs f64 %.2e  vs f32 %.2e   ref32-vs-ref64 %.2e" % (k, rel_err(s, g[k+".f64sample"]), rel_err(s, g[k+".sample"]), rel_err(g[k+".sample"], g[k+".f64sample"])))
