import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "monocon-pytorch_amd")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from hipmonocon import synth
from model import MonoConDetector
from solver import AdamW, CyclicScheduler
B = int(os.environ.get("TB", "32")); H, W = 384, 1280
stats = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "bn_calib_seed7.npz"))
sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
m = MonoConDetector(34, pretrained_backbone=False); m.load_state_dict(sd); m = m.cuda().train()
if os.environ.get('PREC'): m.set_precision(os.environ['PREC'])
opt = AdamW(m.parameters(), lr=2.25e-4, weight_decay=1e-5, betas=(0.95, 0.99), max_grad_norm=35.0)
sch = CyclicScheduler(opt, total_steps=1000)
small = synth.make_batch(500, 8, H, W); rep = (B + 7) // 8
batch = {"img": small["img"].repeat(rep, 1, 1, 1)[:B].cuda().contiguous(),
         "label": {k: v.repeat(rep, *([1] * (v.dim() - 1)))[:B].cuda().contiguous() for k, v in small["label"].items()},
         "img_metas": {"pad_shape": [(H, W)] * B}}
def step():
    opt.zero_grad(); _, loss = m(batch); t = sum(loss.values()); t.backward(); opt.step(); sch.step(); return t
step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(int(os.environ.get("NSTEP","150"))): step()
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) / int(os.environ.get("NSTEP","150")) * 1e3, "ws GB", m._rt.engine.workspace_bytes() / 1e9)
t0 = time.perf_counter()
for _ in range(5): step()
t_issue = (time.perf_counter() - t0) / 5
torch.cuda.synchronize()
print("cpu issue ms/step", t_issue * 1e3)
