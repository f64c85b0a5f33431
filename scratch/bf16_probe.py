"""GPU probe: how the bf16-operand mode tracks the fp64 reference on the conditioned fixtures."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "monocon-pytorch_amd")); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
from conftest import load_golden, grad_rel_l2, gsample
from hipmonocon import synth, netspec
from model import MonoConDetector
def to_cuda(b):
    d = dict(b); d["img"] = b["img"].cuda(); d["label"] = {k: v.cuda() for k, v in b["label"].items()}; return d
stats = load_golden("bn_calib_seed7.npz")
cond = synth.make_conditioned_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
def run(prec, batch):
    m = MonoConDetector(34, pretrained_backbone=False); m.load_state_dict(cond, strict=True); m = m.cuda().train(); m.set_precision(prec)
    pred, loss = m(batch); sum(loss.values()).backward(); torch.cuda.synchronize()
    return m, {k: v.detach().double().cpu() for k, v in pred.items()}, {k: float(v.detach()) for k, v in loss.items()}
for case in range(4):
    g = load_golden("train_cond_%d.npz" % case)
    B, H, W = (int(x) for x in g["shape"])
    batch = to_cuda(synth.make_conditioned_batch(int(g["seed"]), B, H, W))
    m32, p32, l32 = run("fp32", batch)
    for prec in ("bf16",):
        m, p, l = run(prec, batch)
        le = max(abs(l[k] - float(g["f64." + k])) / abs(float(g["f64." + k])) for k in l)
        pe = max(float((p[k] - p32[k]).norm() / p32[k].norm()) for k in p)
        num = den1 = den2 = 0.0
        errs = []
        for n, q in m.named_parameters():
            if q.grad is None: continue
            a = gsample(q.grad).double().cpu().reshape(-1); b = torch.from_numpy(g["g64." + n]).double().reshape(-1)
            num += float((a * b).sum()); den1 += float((a * a).sum()); den2 += float((b * b).sum())
            errs.append(grad_rel_l2(q.grad, g["g64." + n], g["gnorm64." + n], q.numel()))
        e = np.array(errs)
        print("case %d %s: loss rel %.2e | pred rel-L2 vs fp32 path %.2e | grad cosine vs fp64 ref %.5f | per-tensor rel-L2 med %.2e max %.2e" % (case, prec, le, pe, num / (den1 * den2) ** 0.5, np.median(e), e.max()), flush=True)
