"""B=32 train step time of one precision mode (one process = one plan build + autotune): scratch A/B helper"""
import os, sys, time
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "monocon-pytorch_amd"), REPO):
    sys.path.insert(0, p)
from hipmonocon import synth
from model import MonoConDetector
from solver import AdamW
mode = sys.argv[1] if len(sys.argv) > 1 else "f16x2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
stats = np.load(os.path.join(REPO, "tests", "golden", "bn_calib_seed7.npz"))
sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
m = MonoConDetector(34, pretrained_backbone=False); m.load_state_dict(sd, strict=True)
m = m.cuda().train().set_precision(mode)
opt = AdamW(m.parameters(), lr=2.25e-4, weight_decay=1e-5, betas=(0.95, 0.99), max_grad_norm=35.0)
W = int(os.environ.get("WIDTH", "1280"))
b = synth.make_batch(500, 8, 384, W)
bt = {"img": b["img"].repeat(4, 1, 1, 1).cuda().contiguous(),
      "label": {k: v.repeat(4, *([1] * (v.dim() - 1))).cuda().contiguous() for k, v in b["label"].items()},
      "img_metas": {"pad_shape": [(384, W)] * 32}}
def step():
    opt.zero_grad(); _, loss = m(bt); sum(loss.values()).backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
prof = m._rt.engine.profile_train(iters=2)
print("%-8s %s %.2f ms/step  %.1f img/s   conv %.2f  wgrad %.2f  other %.2f ms" % (mode, os.environ.get("TAG", ""), ms, 32 / ms * 1e3, prof["conv"]["ms"], prof["wgrad"]["ms"], prof["other"]["ms"]))
