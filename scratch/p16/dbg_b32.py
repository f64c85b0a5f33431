"""mode 4: per-parameter gradient of the B=2 pair vs the pair repeated 16 times (should agree to ~1e-6)"""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(REPO, "monocon-pytorch_amd"), REPO, os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from hipmonocon import synth
from model import MonoConDetector
mode = sys.argv[1] if len(sys.argv) > 1 else "f16x2p"
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 16
g = np.load(os.path.join(REPO, "tests", "golden", "train_full.npz"))
sd = synth.make_conditioned_state_dict(int(g["sd_seed"])) if "sd_seed" in g.files else None
if sd is None:
    import conftest
    stats = np.load(os.path.join(REPO, "tests", "golden", "bn_calib_seed7.npz"))
    sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
b2 = synth.make_conditioned_batch(int(g["seed"]), 2, 384, 1280)
def run(batch):
    m = MonoConDetector(34, pretrained_backbone=False); m.load_state_dict(sd, strict=True)
    m = m.cuda().train().set_precision(mode)
    _, loss = m(batch); sum(loss.values()).backward(); torch.cuda.synchronize()
    return {n: p.grad.detach().double().cpu() for n, p in m.named_parameters() if p.grad is not None}
ga = run({"img": b2["img"].cuda(), "label": {k: v.cuda() for k, v in b2["label"].items()}, "img_metas": b2["img_metas"]})
gb = run({"img": b2["img"].repeat(rep, 1, 1, 1).cuda(), "label": {k: v.repeat(rep, *([1] * (v.dim() - 1))).cuda() for k, v in b2["label"].items()},
          "img_metas": {"pad_shape": [(384, 1280)] * (2 * rep)}})
rows = sorted(((float((ga[n] - gb[n]).norm() / (ga[n].norm() + 1e-30)), n) for n in ga), reverse=True)
for e, n in rows[:25]: print("%.3e %s" % (e, n))
print("median %.3e" % rows[len(rows) // 2][0])
gc = run({"img": b2["img"].repeat(rep, 1, 1, 1).cuda(), "label": {k: v.repeat(rep, *([1] * (v.dim() - 1))).cuda() for k, v in b2["label"].items()},
          "img_metas": {"pad_shape": [(384, 1280)] * (2 * rep)}})
nd = [n for n in gb if not torch.equal(gb[n], gc[n])]
print("B=%d run-to-run: %d of %d tensors differ" % (2 * rep, len(nd), len(gb)), nd[:8])
gd = run({"img": b2["img"].cuda(), "label": {k: v.cuda() for k, v in b2["label"].items()}, "img_metas": b2["img_metas"]})
nd = [n for n in ga if not torch.equal(ga[n], gd[n])]
print("B=2 run-to-run: %d of %d tensors differ" % (len(nd), len(ga)), nd[:8])
