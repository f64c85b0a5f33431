"""GPU probe: actual parity numbers behind the tolerances in tests/ (run on the MI355X box)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "monocon-pytorch_amd")); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
from conftest import load_golden, grad_rel_l2, rel_err, GOLDEN_SEED
from hipmonocon import synth, netspec
from model import MonoConDetector

def to_cuda(b):
    d = dict(b); d["img"] = b["img"].cuda(); d["label"] = {k: v.cuda() for k, v in b["label"].items()}; return d
stats = load_golden("bn_calib_seed7.npz")
cond = synth.make_conditioned_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
gold = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
def model(sd, prec):
    m = MonoConDetector(34, pretrained_backbone=False); m.load_state_dict(sd, strict=True); m = m.cuda().train(); m.set_precision(prec); return m
for prec in ("fp32", "bf16x3"):
    for case in range(4):
        g = load_golden("train_cond_%d.npz" % case)
        B, H, W = (int(x) for x in g["shape"])
        m = model(cond, prec)
        _, loss = m(to_cuda(synth.make_conditioned_batch(int(g["seed"]), B, H, W)))
        sum(loss.values()).backward(); torch.cuda.synchronize()
        le = max(abs(float(v) - float(g["f64." + k])) / abs(float(g["f64." + k])) for k, v in loss.items())
        errs = {n: grad_rel_l2(p.grad, g["g64." + n], g["gnorm64." + n], p.numel()) for n, p in m.named_parameters() if p.grad is not None}
        e = np.array(list(errs.values()))
        worst = max(errs, key=errs.get)
        be = max(rel_err(v.cpu(), g["buf64." + k]) for k, v in m.state_dict().items() if k.endswith(("running_mean", "running_var")))
        print("cond %s case %d  loss %.1e  grad max %.1e (%s) med %.1e n>1e-3 %d  buf %.1e  ref32 %.1e" % (prec, case, le, e.max(), worst, np.median(e), (e > 1e-3).sum(), be, float(g["ref32_max_err"])), flush=True)
    g = load_golden("train_step.npz")
    m = model(gold, prec)
    _, loss = m(to_cuda(synth.make_batch(GOLDEN_SEED + 4, 2, 192, 384)))
    for k, v in loss.items():
        print("  old fixture %s %-28s vs f64 %.1e  vs ref32 %.1e  (ref32 vs f64 %.1e)" % (prec, k, abs(float(v) - float(g["f64." + k])) / abs(float(g["f64." + k])), abs(float(v) - float(g[k])) / abs(float(g[k])), abs(float(g[k]) - float(g["f64." + k])) / abs(float(g["f64." + k]))))
    from oracle import monocon_oracle as O
    for shape in [(3, 64, 128), (2, 128, 512), (5, 96, 160), (2, 96, 1248)]:
        B, H, W = shape
        batch = synth.make_batch(2000 + B + H + W, B, H, W)
        live64 = {k: (v.double().clone() if v.dtype == torch.float32 else v.clone()) for k, v in gold.items()}
        b64 = dict(batch); b64["img"] = batch["img"].double()
        with torch.no_grad():
            _, _, L64, _ = O.train_forward(live64, b64)
            _, _, L32, _ = O.train_forward({k: v.clone() for k, v in gold.items()}, batch)
        m = model(gold, prec)
        _, loss = m(to_cuda(batch))
        print("  sweep %s %s: hip vs oracle64 %.1e | hip vs oracle32 %.1e | oracle32 vs 64 %.1e" % (prec, shape,
              max(abs(float(v) - float(L64[k])) / abs(float(L64[k])) for k, v in loss.items()),
              max(abs(float(v) - float(L32[k])) / abs(float(L32[k])) for k, v in loss.items()),
              max(abs(float(L32[k]) - float(L64[k])) / abs(float(L64[k])) for k in L64)), flush=True)
