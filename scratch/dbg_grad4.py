import sys, os
sys.path.insert(0, 'monocon-pytorch_amd'); sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from conftest import load_golden
from hipmonocon import synth, netspec
from oracle import monocon_oracle as O
from model import MonoConDetector
stats = load_golden("bn_calib_seed7.npz")
sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
roles = netspec.state_shapes()
B, H, W, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
m = MonoConDetector(34, pretrained_backbone=False); m.load_state_dict(sd); m = m.cuda().train()
batch = synth.make_batch(seed, B, H, W)
cb = dict(batch); cb["img"] = batch["img"].cuda(); cb["label"] = {k: v.cuda() for k, v in batch["label"].items()}
pred, loss = m(cb); sum(loss.values()).backward(); torch.cuda.synchronize()
def oracle(dtype):
    s = {k: (v.detach().clone().to(dtype) if v.dtype == torch.float32 else v.clone()) for k, v in sd.items()}
    for k, v in s.items():
        if roles[k][2] == 'param': v.requires_grad_(True)
    b = synth.make_batch(seed, B, H, W); b["img"] = b["img"].to(dtype)
    _, _, L, _ = O.train_forward(s, b); sum(L.values()).backward()
    return {k: v.grad for k, v in s.items() if roles[k][2] == 'param' and v.grad is not None}
g64 = oracle(torch.float64); g32 = oracle(torch.float32)
rows = []
for n, p in m.named_parameters():
    if p.grad is None: continue
    r = g64[n]; sc = float(r.norm()) + 1e-30
    rows.append((n.split(".")[0], n, float((p.grad.cpu().double() - r).norm() / sc), float((g32[n].double() - r).norm() / sc)))
for sec in ("head", "neck", "backbone"):
    sel = [x for x in rows if x[0] == sec]
    print("%-9s hip med %.2e max %.2e | cpu32 med %.2e max %.2e" % (sec, np.median([x[2] for x in sel]), max(x[2] for x in sel), np.median([x[3] for x in sel]), max(x[3] for x in sel)))
rows.sort(key=lambda x: -x[2] / (x[3] + 1e-9))
print("largest hip/cpu32 ratios:"); [print("  %-55s hip %.2e cpu32 %.2e" % x[1:]) for x in rows[:8]]
