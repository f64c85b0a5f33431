"""print the train plan's backward-statistics twins (MONOCON_HIP_PLAN_DEBUG=1) at the headline shape, B = 2"""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "monocon-pytorch_amd"), REPO):
    sys.path.insert(0, p)
os.environ["MONOCON_HIP_PLAN_DEBUG"] = "1"
import numpy as np, torch
from hipmonocon import synth
from model import MonoConDetector
stats = np.load(os.path.join(REPO, "tests", "golden", "bn_calib_seed7.npz"))
sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
b = synth.make_batch(5, 2, 384, 1280)
gb = {"img": b["img"].cuda(), "label": {k: v.cuda() for k, v in b["label"].items()}, "img_metas": b["img_metas"]}
m = MonoConDetector(34, pretrained_backbone=False); m.load_state_dict(sd); m = m.cuda().train().set_precision("f16x2")
_, loss = m(gb); sum(loss.values()).backward(); torch.cuda.synchronize()
