import sys, os
sys.path.insert(0, 'monocon-pytorch_amd'); sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from conftest import load_golden, rel_err
from hipmonocon import synth, netspec
from oracle import monocon_oracle as O
from model import MonoConDetector
stats = load_golden("bn_calib_seed7.npz")
sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
m = MonoConDetector(34, pretrained_backbone=False); m.load_state_dict(sd); m = m.cuda().train()
batch = synth.make_batch(11, 2, 192, 384)
cb = dict(batch); cb["img"] = batch["img"].cuda(); cb["label"] = {k: v.cuda() for k, v in batch["label"].items()}
pred, loss = m(cb); sum(loss.values()).backward(); torch.cuda.synchronize()
roles = netspec.state_shapes()
sd64 = {k: (v.detach().clone().double() if v.dtype == torch.float32 else v.clone()) for k, v in sd.items()}
for k, v in sd64.items():
    if roles[k][2] == 'param': v.requires_grad_(True)
b64 = synth.make_batch(11, 2, 192, 384); b64["img"] = b64["img"].double()
_, _, L, _ = O.train_forward(sd64, b64); sum(L.values()).backward()
for n in ["backbone.level4.tree1.tree1.conv1.weight", "backbone.level4.tree1.tree1.bn1.weight", "backbone.level3.tree1.tree1.conv1.weight", "backbone.level4.tree1.tree1.conv2.weight"]:
    a = dict(m.named_parameters())[n].grad.cpu().double(); r = sd64[n].grad
    d = (a - r).abs(); mx = float(r.abs().max())
    print(n, "max err/max", float(d.max()/mx), "frac>1%", float((d > 0.01*mx).float().mean()), "frac>5%", float((d > 0.05*mx).float().mean()))
    if a.dim() == 4:
        idx = (d > 0.05*mx).nonzero()
        print("  bad idx sample", idx[:12].tolist(), "taps hist", torch.bincount((idx[:,2]*3+idx[:,3]), minlength=9).tolist() if len(idx) else None)
