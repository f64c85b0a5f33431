"""run-to-run reproducibility of the PARAMETER gradients of a B=NB train step in a given mode"""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(REPO, "monocon-pytorch_amd"), REPO):
    sys.path.insert(0, p)
import numpy as np, torch
from hipmonocon import synth
from model import MonoConDetector
mode = sys.argv[1]; NB = int(sys.argv[2])
stats = np.load(os.path.join(REPO, "tests", "golden", "bn_calib_seed7.npz"))
sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
batch = synth.make_conditioned_batch(11, NB, 384, 1280)
batch = {"img": batch["img"].cuda(), "label": {k: v.cuda() for k, v in batch["label"].items()}, "img_metas": batch["img_metas"]}
m = MonoConDetector(34, pretrained_backbone=False); m.load_state_dict(sd, strict=True)
m = m.cuda().train().set_precision(mode)
runs = []
for it in range(4):
    m.zero_grad(set_to_none=True)
    m.load_state_dict(sd, strict=True)
    _, loss = m(batch); sum(loss.values()).backward(); torch.cuda.synchronize()
    runs.append({n: p.grad.detach().clone().cpu() for n, p in m.named_parameters() if p.grad is not None})
for r in range(1, 4):
    bad = [(n, float((runs[0][n] - runs[r][n]).abs().max()), float(runs[0][n].abs().max())) for n in runs[0] if not torch.equal(runs[0][n], runs[r][n])]
    print("%s dual=%s run 0 vs %d: %d of %d parameter gradients differ" % (mode, os.environ.get("MONOCON_HIP_DUAL_STREAM", "1"), r, len(bad), len(runs[0])))
    names = list(runs[0])
    print("   equal: " + " ".join(n for n in names if n not in {b[0] for b in bad})[:3000])
    for n, d, mx in bad[-14:]:
        print("   %-50s rel diff %.3e" % (n, d / (mx + 1e-30)))
