import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "monocon-pytorch_amd")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from hipmonocon import synth
from model import MonoConDetector
from solver import AdamW, CyclicScheduler
B, H, W = 32, 384, 1280
stats = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "bn_calib_seed7.npz"))
sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
m = MonoConDetector(34, pretrained_backbone=False); m.load_state_dict(sd); m = m.cuda().train()
if os.environ.get("PREC"): m.set_precision(os.environ["PREC"])
opt = AdamW(m.parameters(), lr=2.25e-4, weight_decay=1e-5, betas=(0.95, 0.99), max_grad_norm=35.0)
sch = CyclicScheduler(opt, total_steps=1000)
small = synth.make_batch(500, 8, H, W)
batch = {"img": small["img"].repeat(4, 1, 1, 1)[:B].cuda().contiguous(),
         "label": {k: v.repeat(4, *([1] * (v.dim() - 1)))[:B].cuda().contiguous() for k, v in small["label"].items()},
         "img_metas": {"pad_shape": [(H, W)] * B}}
def step():
    opt.zero_grad(); _, loss = m(batch); t = sum(loss.values()); t.backward(); opt.step(); sch.step(); return t
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    step(); torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
