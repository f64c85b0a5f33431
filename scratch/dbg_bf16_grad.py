import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "monocon-pytorch_amd")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from hipmonocon import synth
from model import MonoConDetector
stats = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "bn_calib_seed7.npz"))
sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
batch = synth.make_batch(10, 2, 96, 160)
batch = {"img": batch["img"].cuda(), "label": {k: v.cuda() for k, v in batch["label"].items()}, "img_metas": batch["img_metas"]}
G = {}
for mode in ("fp32", os.environ.get("DBG_MODE", "bf16")):
    m = MonoConDetector(34, pretrained_backbone=False); m.load_state_dict(sd, strict=True)
    m = m.cuda().train().set_precision(mode)
    _, loss = m(batch)
    sum(v for k, v in loss.items() if k != "loss_depth").backward()
    G[mode] = {n: p.grad.detach().double().flatten().clone() for n, p in m.named_parameters() if p.grad is not None}
def cos(a, b): return float(torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-300))
rows = []
for n in G["fp32"]:
    if n.endswith("weight") and G["fp32"][n].numel() > 500:
        rows.append((n, cos(G[os.environ.get("DBG_MODE", "bf16")][n], G["fp32"][n]), float(G[os.environ.get("DBG_MODE", "bf16")][n].norm() / G["fp32"][n].norm().clamp_min(1e-300))))
for n, c, r in rows:
    if ".0.weight" in n and n.startswith("head.") or any(t in n for t in ("head.heatmap_head", "head.wh_head", "neck.ida_2.node_3", "neck.ida_2.proj_1", "neck.ida_0", "level5.tree2.conv2", "level5.tree1.conv1", "level4.tree1.tree1.conv1", "level3.tree1.tree1.conv1", "level2.tree1.conv1", "level2.root", "level1", "level0", "base_layer")):
        print("%-52s cos %.4f  norm ratio %.3f" % (n, c, r))
