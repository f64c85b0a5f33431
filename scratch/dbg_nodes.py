import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "monocon-pytorch_amd")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from hipmonocon import synth
from model import MonoConDetector
stats = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "bn_calib_seed7.npz"))
sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
batch = synth.make_batch(10, 2, 96, 160)
batch = {"img": batch["img"].cuda(), "label": {k: v.cuda() for k, v in batch["label"].items()}, "img_metas": batch["img_metas"]}
def node(eng, i, which):
    dims = (C.c_int * 4)()
    rc = eng.lib.mc_train_debug_node(eng.h, i, which, None, dims, None)
    if rc: return None
    t = torch.empty(tuple(dims), dtype=torch.float32, device="cuda")
    rc = eng.lib.mc_train_debug_node(eng.h, i, which, C.c_void_p(t.data_ptr()), dims, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return t if rc == 0 else None
R = {}
for mode in ("bf16x3", "bf16"):
    m = MonoConDetector(34, pretrained_backbone=False); m.load_state_dict(sd, strict=True)
    m = m.cuda().train().set_precision(mode)
    _, loss = m(batch)
    sum(v for k, v in loss.items() if k != "loss_depth").backward()
    torch.cuda.synchronize()
    eng = m._rt.engine
    out = []
    i = 0
    while True:
        a = node(eng, i, 0)
        if a is None: break
        g = node(eng, i, 1)
        out.append((tuple(a.shape), a.double().cpu(), None if g is None else g.double().cpu()))
        i += 1
    R[mode] = out
def cos(a, b): return float((a * b).sum() / (a.norm() * b.norm()).clamp_min(1e-300))
for i, (x, y) in enumerate(zip(R["bf16x3"], R["bf16"])):
    ga, gb = x[2], y[2]
    print("node %2d %-18s act relL2 %.3e   grad: %s" % (i, x[0], float((x[1] - y[1]).norm() / x[1].norm().clamp_min(1e-30)),
          "none" if ga is None or gb is None else "cos %.4f ratio %.3f" % (cos(ga, gb), float(gb.norm() / ga.norm().clamp_min(1e-300)))))
