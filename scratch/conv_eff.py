import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "monocon-pytorch_amd")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
os.environ["MONOCON_HIP_PROFILE_DUMP"] = "1"
import numpy as np, torch
from hipmonocon import synth
from model import MonoConDetector
B, H, W = 32, 384, 1280
stats = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "bn_calib_seed7.npz"))
sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
m = MonoConDetector(34, pretrained_backbone=False); m.load_state_dict(sd); m = m.cuda().train()
small = synth.make_batch(500, 8, H, W); rep = 4
batch = {"img": small["img"].repeat(rep, 1, 1, 1)[:B].cuda().contiguous(),
         "label": {k: v.repeat(rep, *([1] * (v.dim() - 1)))[:B].cuda().contiguous() for k, v in small["label"].items()},
         "img_metas": {"pad_shape": [(H, W)] * B}}
for _ in range(2):
    m.zero_grad(); _, loss = m(batch); sum(loss.values()).backward()
torch.cuda.synchronize()
m._rt.engine.profile_train(iters=1)
