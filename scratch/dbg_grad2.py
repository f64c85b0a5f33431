import sys, os
sys.path.insert(0, 'monocon-pytorch_amd'); sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from conftest import load_golden, rel_err
from hipmonocon import synth
from model import MonoConDetector
stats = load_golden("bn_calib_seed7.npz")
sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
m = MonoConDetector(34, pretrained_backbone=False); m.load_state_dict(sd); m = m.cuda().train()
batch = synth.make_batch(11, 2, 192, 384)
cb = dict(batch); cb["img"] = batch["img"].cuda(); cb["label"] = {k: v.cuda() for k, v in batch["label"].items()}
pred, loss = m(cb); sum(loss.values()).backward(); torch.cuda.synchronize()
g = load_golden("train_step.npz")
rows = []
for n, p in m.named_parameters():
    if p.grad is None: continue
    rn = float(g["gnorm." + n]); gn = float(p.grad.double().norm())
    e = rel_err(p.grad.cpu().reshape(-1)[::101], g["gsample." + n])
    rows.append((n, gn / rn if rn else 0, e))
bad = [r for r in rows if r[2] > 5e-3]
print("params", len(rows), "bad", len(bad))
for r in rows[::-1]:
    print("%-60s ratio %.4f err %.3e" % r)
for n in ["head.kpt_heatmap_offset_head.3.bias", "head.wh_head.3.bias", "head.depth_head.3.bias"]:
    p = dict(m.named_parameters())[n]
    print(n, p.grad.cpu().numpy(), "golden sample", g["gsample." + n], "gnorm", float(g["gnorm." + n]))
