"""per-launch profile dump (mc_profile_train with MONOCON_HIP_PROFILE_DUMP) of the B=32 train step in a given mode"""
import os, sys
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(REPO, "monocon-pytorch_amd"), REPO):
    sys.path.insert(0, p)
from hipmonocon import synth
from model import MonoConDetector
mode = sys.argv[1]
stats = np.load(os.path.join(REPO, "tests", "golden", "bn_calib_seed7.npz"))
sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
m = MonoConDetector(34, pretrained_backbone=False); m.load_state_dict(sd, strict=True)
m = m.cuda().train().set_precision(mode)
b = synth.make_batch(5, 8, 384, 1280)
bt = {"img": b["img"].repeat(4, 1, 1, 1).cuda().contiguous(),
      "label": {k: v.repeat(4, *([1] * (v.dim() - 1))).cuda().contiguous() for k, v in b["label"].items()},
      "img_metas": {"pad_shape": [(384, 1280)] * 32}}
for _ in range(2):
    _, loss = m(bt); sum(loss.values()).backward()
torch.cuda.synchronize()
os.environ["MONOCON_HIP_PROFILE_DUMP"] = "1"
m._rt.engine.profile_train(iters=1)
