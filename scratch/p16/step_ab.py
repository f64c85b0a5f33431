"""mode 3 (f16x2) vs mode 4 (f16x2p) on the same train step: losses, gradient agreement, then step time at B=32."""
import os, sys, time
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(REPO, "monocon-pytorch_amd"), REPO):
    sys.path.insert(0, p)
from hipmonocon import synth
from model import MonoConDetector
from solver import AdamW, CyclicScheduler

stats = np.load(os.path.join(REPO, "tests", "golden", "bn_calib_seed7.npz"))
sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})

def build(mode):
    m = MonoConDetector(34, pretrained_backbone=False)
    m.load_state_dict(sd, strict=True)
    return m.cuda().train().set_precision(mode)

def batch(B, H, W, seed=5):
    b = synth.make_batch(seed, min(B, 8), H, W)
    rep = (B + 7) // 8
    return {"img": b["img"].repeat(rep, 1, 1, 1)[:B].cuda().contiguous(),
            "label": {k: v.repeat(rep, *([1] * (v.dim() - 1)))[:B].cuda().contiguous() for k, v in b["label"].items()},
            "img_metas": {"pad_shape": [(H, W)] * B}}

small = len(sys.argv) > 1 and sys.argv[1] == "small"
res = {}
for mode in ("f16x2", "f16x2p"):
    m = build(mode)
    bt = batch(2, 96, 320) if small else batch(2, 384, 1280)
    _, loss = m(bt)
    sum(loss.values()).backward()
    torch.cuda.synchronize()
    res[mode] = ({k: float(v.detach()) for k, v in loss.items()},
                 {n: p.grad.detach().double().clone() for n, p in m.named_parameters() if p.grad is not None})
    del m
la, ga = res["f16x2"]; lb, gb = res["f16x2p"]
print("losses:", {k: (round(la[k], 6), round(lb[k], 6)) for k in la})
worst = max(abs(la[k] - lb[k]) / (abs(la[k]) + 1e-9) for k in la)
errs = sorted(((float((ga[n] - gb[n]).norm() / (ga[n].norm() + 1e-30)), n) for n in ga), reverse=True)
print("max rel loss diff %.3e; gradient rel-L2 mode4 vs mode3: worst %s median %.3e" % (worst, errs[:3], errs[len(errs) // 2][0]))
assert all(np.isfinite(v) for v in lb.values())
if small:
    sys.exit(0)
for mode in ("f16x2", "f16x2p", "f16x2", "f16x2p"):
    m = build(mode)
    opt = AdamW(m.parameters(), lr=2.25e-4, weight_decay=1e-5, betas=(0.95, 0.99), max_grad_norm=35.0)
    bt = batch(32, 384, 1280)
    def step():
        opt.zero_grad(); _, loss = m(bt); sum(loss.values()).backward(); opt.step()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 8 * 1e3
    prof = m._rt.engine.profile_train(iters=2)
    print("%-7s %.2f ms/step  %.1f img/s   conv %.2f  wgrad %.2f  other %.2f ms" % (mode, ms, 32 / ms * 1e3, prof["conv"]["ms"], prof["wgrad"]["ms"], prof["other"]["ms"]))
    del m, opt
