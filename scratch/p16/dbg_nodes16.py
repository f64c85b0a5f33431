"""node-by-node activations of the train forward, mode 3 vs mode 4, at a given size (forward only: gradients pooled)"""
import sys, os, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(REPO, "monocon-pytorch_amd"), REPO):
    sys.path.insert(0, p)
import numpy as np, torch
from hipmonocon import synth
from model import MonoConDetector
H, W = int(sys.argv[1]), int(sys.argv[2])
cond = len(sys.argv) > 3
stats = np.load(os.path.join(REPO, "tests", "golden", "bn_calib_seed7.npz"))
sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
NB = int(os.environ.get("NB", "2"))
batch = synth.make_conditioned_batch(11, NB, H, W) if cond else synth.make_batch(10, NB, H, W)
batch = {"img": batch["img"].cuda(), "label": {k: v.cuda() for k, v in batch["label"].items()}, "img_metas": batch["img_metas"]}
def node(eng, i):
    dims = (C.c_int * 4)()
    if eng.lib.mc_train_debug_node(eng.h, i, 0, None, dims, None): return None
    t = torch.empty(tuple(dims), dtype=torch.float32, device="cuda")
    rc = eng.lib.mc_train_debug_node(eng.h, i, 0, C.c_void_p(t.data_ptr()), dims, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return t if rc == 0 else None
R = {}
for mode in ("f16x2", "f16x2p"):
    m = MonoConDetector(34, pretrained_backbone=False); m.load_state_dict(sd, strict=True)
    m = m.cuda().train().set_precision(mode)
    if mode == 'f16x2p' and os.environ.get('FORCE_CFG'):
        m._engine().set_conv_cfg(int(os.environ['FORCE_CFG']))
    pred, loss = m(batch)
    torch.cuda.synchronize()
    eng = m._rt.engine
    out, i = [], 0
    while True:
        a = node(eng, i)
        if a is None: break
        out.append(a.double().cpu()); i += 1
    R[mode] = (out, {k: v.detach().double().cpu() for k, v in pred.items()})
for i, (x, y) in enumerate(zip(R["f16x2"][0], R["f16x2p"][0])):
    d = (x - y).abs()
    print("node %2d %-22s max|x| %.3e  maxerr/max %.3e  relL2 %.3e  nan %d" % (i, tuple(x.shape), float(x.abs().max()), float(d.max() / x.abs().max().clamp_min(1e-30)),
          float((x - y).norm() / x.norm().clamp_min(1e-30)), int(torch.isnan(y).sum())))
for k in R["f16x2"][1]:
    x, y = R["f16x2"][1][k], R["f16x2p"][1][k]
    print("pred %-28s maxerr/max %.3e" % (k, float((x - y).abs().max() / x.abs().max())))
