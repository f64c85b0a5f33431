import sys, os
sys.path.insert(0, 'monocon-pytorch_amd'); sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import test_hip_train_pieces as tp
from hipmonocon.engine import Engine
from oracle import monocon_oracle as O
eng = Engine()
batch, preds, Tref, _ = tp._train_case(16)
k = "kpt_heatmap_pred"
v = preds[k]
raw = torch.logit(v).clone().requires_grad_(True)
act = torch.clamp(torch.sigmoid(raw), 1e-4, 1 - 1e-4)
L = O.gaussian_focal(act, Tref["kpt_heatmap_target"]) * 0.7
L.backward()
label = {kk: vv.cuda() for kk, vv in batch["label"].items()}
T = eng.make_targets(label, (192, 384), (48, 96))
w = torch.zeros(10); w[5] = 0.7
pd = {kk: vv.detach().cuda().contiguous() for kk, vv in preds.items()}
pd[k] = act.detach().cuda().contiguous()
d = eng.losses_backward(pd, T, w.cuda())[k].cpu()
ref = raw.grad
diff = (d - ref).abs()
i = int(diff.argmax()); print("max diff", float(diff.max()), "ref max", float(ref.abs().max()))
print("at", np.unravel_index(i, d.shape), "mine", float(d.flatten()[i]), "ref", float(ref.flatten()[i]), "p", float(act.flatten()[i]), "t_ref", float(Tref["kpt_heatmap_target"].flatten()[i]), "t_gpu", float(T["kpt_heatmap_target"].cpu().flatten()[i]))
print("npos ref", float((Tref["kpt_heatmap_target"]==1).sum()), "gpu", float((T["kpt_heatmap_target"]==1).sum()))
print("num clamped lo", int((v <= 1e-4).sum()), "hi", int((v >= 1-1e-4).sum()))
