import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "monocon-pytorch_amd")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from hipmonocon import synth
from model import MonoConDetector
stats = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "bn_calib_seed7.npz"))
sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
batch = synth.make_batch(10, 2, 96, 160)
batch = {"img": batch["img"].cuda(), "label": {k: v.cuda() for k, v in batch["label"].items()}, "img_metas": batch["img_metas"]}
def run(sdx, mode="fp32"):
    m = MonoConDetector(34, pretrained_backbone=False); m.load_state_dict(sdx, strict=True)
    m = m.cuda().train().set_precision(mode)
    _, loss = m(batch)
    sum(v for k, v in loss.items() if k != "loss_depth").backward()
    return {n: p.grad.detach().double().flatten().clone() for n, p in m.named_parameters() if p.grad is not None}, {k: float(v.detach()) for k, v in loss.items()}
g0, l0 = run(sd)
gen = torch.Generator().manual_seed(3)
for eps in (1e-5, 1e-4, 2e-3):
    sdp = {k: (v * (1 + eps * torch.randn(v.shape, generator=gen)) if (v.dtype == torch.float32 and v.dim() == 4) else v) for k, v in sd.items()}
    g1, l1 = run(sdp)
    for n in ("head.kpt_heatmap_head.0.weight", "head.heatmap_head.0.weight", "neck.ida_2.node_3.conv.weight", "backbone.level3.tree1.tree1.conv1.weight"):
        a, b = g0[n], g1[n]
        print("eps %.0e %-44s cos %.4f ratio %.3f" % (eps, n, float(torch.dot(a, b) / (a.norm() * b.norm())), float(b.norm() / a.norm())))
    print("   loss_kpt_heatmap", l0["loss_kpt_heatmap"], l1["loss_kpt_heatmap"])
