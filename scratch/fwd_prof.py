import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "monocon-pytorch_amd")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from hipmonocon import synth
from model import MonoConDetector
B = int(os.environ.get("TB", "32")); H, W = 384, 1280
stats = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "bn_calib_seed7.npz"))
sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
m = MonoConDetector(34, pretrained_backbone=False); m.load_state_dict(sd); m = m.cuda().eval()
if os.environ.get('PREC'): m.set_precision(os.environ['PREC'])
img = torch.randn(B, 3, H, W, device="cuda")
batch = {"img": img, "img_metas": {"pad_shape": [(H, W)] * B}}
with torch.no_grad():
    for _ in range(3): m(batch, return_loss=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(6): m(batch, return_loss=False)
    torch.cuda.synchronize(); print("fwd ms", (time.perf_counter() - t0) / 6 * 1e3)
