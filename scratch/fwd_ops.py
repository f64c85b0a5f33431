"""per-op durations of the B = 32 eval forward (mc_profile_forward with MONOCON_HIP_PROFILE_DUMP=1)"""
import sys, os
os.environ["MONOCON_HIP_PROFILE_DUMP"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "monocon-pytorch_amd")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from hipmonocon import synth
from model import MonoConDetector
B = int(os.environ.get("TB", "32")); H, W = 384, int(os.environ.get("WIDTH", "1280"))
stats = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "bn_calib_seed7.npz"))
sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
m = MonoConDetector(34, pretrained_backbone=False); m.load_state_dict(sd); m = m.cuda().eval().set_precision(os.environ.get("PREC", "f16x2"))
img = torch.randn(B, 3, H, W, device="cuda")
batch = {"img": img, "img_metas": {"pad_shape": [(H, W)] * B}}
with torch.no_grad():
    for _ in range(3): m(batch, return_loss=False)
    torch.cuda.synchronize()
    print(m._rt.engine.profile_forward(iters=2))
