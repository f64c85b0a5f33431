"""fused stride-2 thin data gradient (MONOCON_HIP_DGRAD_S2_THIN) on vs off: gradients of the layers below level1"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "monocon-pytorch_amd")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from hipmonocon import synth
from model import MonoConDetector
stats = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "bn_calib_seed7.npz"))
sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
b = synth.make_batch(5, 3, 96, 224)
gb = {"img": b["img"].cuda(), "label": {k: v.cuda() for k, v in b["label"].items()}, "img_metas": b["img_metas"]}
res = []
for flag in ("0", "1"):
    os.environ["MONOCON_HIP_DGRAD_S2_THIN"] = flag
    m = MonoConDetector(34, pretrained_backbone=False); m.load_state_dict(sd); m = m.cuda().train().set_precision("f16x2")
    _, loss = m(gb); sum(loss.values()).backward(); torch.cuda.synchronize()
    res.append({n: p.grad.detach().double().cpu() for n, p in m.named_parameters() if p.grad is not None})
worst = 0
for n in res[0]:
    a, c = res[0][n], res[1][n]
    e = float((a - c).norm() / (a.norm() + 1e-30))
    if e > 0: print("%-50s rel %.3e" % (n, e))
    worst = max(worst, e)
print("worst", worst)
