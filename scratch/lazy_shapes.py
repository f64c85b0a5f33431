"""lazy activations on / off over batch sizes and image shapes (incl. the real KITTI pad shape 384x1248 and B = 64): losses and
the flat gradient must agree to operand-split round-off; prints the workspace of both"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "monocon-pytorch_amd"), REPO):
    sys.path.insert(0, p)
from hipmonocon import synth
from model import MonoConDetector
stats = np.load(os.path.join(REPO, "tests", "golden", "bn_calib_seed7.npz"))
sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
for B, H, W in ((8, 384, 1248), (16, 352, 1216), (3, 96, 160), (5, 160, 416), (64, 384, 1280)):
    b = synth.make_batch(900 + B, min(B, 8), H, W)
    rep = (B + 7) // 8
    bt = {"img": b["img"].repeat(rep, 1, 1, 1)[:B].cuda().contiguous(),
          "label": {k: v.repeat(rep, *([1] * (v.dim() - 1)))[:B].cuda().contiguous() for k, v in b["label"].items()},
          "img_metas": {"pad_shape": [(H, W)] * B}}
    out = {}
    for tag, env in (("stored", "0"), ("lazy", "3")):
        os.environ["MONOCON_HIP_LAZY_Z"] = env
        m = MonoConDetector(34, pretrained_backbone=False); m.load_state_dict(sd, strict=True)
        m = m.cuda().train().set_precision("f16x2")
        _, loss = m(bt); sum(loss.values()).backward(); torch.cuda.synchronize()
        g = torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None]).double()
        out[tag] = ({k: float(v) for k, v in loss.items()}, g, m._engine().workspace_bytes() / 1e9)
        del m; torch.cuda.empty_cache()
    dl = max(abs(out["lazy"][0][k] - v) / (abs(v) + 1e-12) for k, v in out["stored"][0].items())
    dg = float((out["lazy"][1] - out["stored"][1]).norm() / out["stored"][1].norm())
    ok = dl < 5e-5 and dg < 5e-3 and bool(torch.isfinite(out["lazy"][1]).all())
    print("B=%d %dx%d  loss rel diff %.2e  flat grad rel diff %.2e  workspace %.2f -> %.2f GB  %s" % (B, H, W, dl, dg, out["stored"][2], out["lazy"][2], "OK" if ok else "FAIL"), flush=True)
    assert ok
