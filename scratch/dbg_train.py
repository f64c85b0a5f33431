import sys, os
sys.path.insert(0, 'monocon-pytorch_amd'); sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from conftest import load_golden, rel_err
from hipmonocon import synth, netspec
from oracle import monocon_oracle as O
from model import MonoConDetector
stats = load_golden("bn_calib_seed7.npz")
sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
m = MonoConDetector(34, pretrained_backbone=False); m.load_state_dict(sd); m = m.cuda().train()
batch = synth.make_batch(11, 2, 192, 384)
cb = dict(batch); cb["img"] = batch["img"].cuda(); cb["label"] = {k: v.cuda() for k, v in batch["label"].items()}
pred, loss = m(cb)
with torch.no_grad():
    rp, T, L, nb = O.train_forward(sd, batch)
for k in pred: print("pred %-26s err %.3e" % (k, rel_err(pred[k], rp[k])))
for k in loss: print("loss %-26s hip %.5f ref %.5f" % (k, float(loss[k]), float(L[k])))
msd = m.state_dict()
bad = [(k, rel_err(msd[k], v)) for k, v in nb.items() if not k.endswith("tracked")]
bad.sort(key=lambda t: -t[1])
print("worst buffers:", bad[:8])
first = [(k, e) for k, e in ((k, rel_err(msd[k], nb[k])) for k in msd if k in nb and not k.endswith("tracked")) if e > 1e-3][:6]
print("first bad buffers in order:", first)
import ctypes as C
eng = m._rt.engine
def node(i, which=0):
    dims = (C.c_int*4)()
    eng.lib.mc_train_debug_node(eng.h, i, which, None, dims, None)
    t = torch.empty(tuple(dims), device='cuda')
    rc = eng.lib.mc_train_debug_node(eng.h, i, which, C.c_void_p(t.data_ptr()), dims, None)
    torch.cuda.synchronize(); return t
cx = O._Ctx(sd, True)
with torch.no_grad():
    x = cx.cbr(batch["img"], "backbone.base_layer.0", "backbone.base_layer.1", 1, 3)
    l0 = cx.cbr(x, "backbone.level0.0", "backbone.level0.1", 1)
    l1 = cx.cbr(l0, "backbone.level1.0", "backbone.level1.1", 2)
for i, ref in enumerate([x, l0, l1]):
    t = node(i); print("node", i, tuple(t.shape), "err", rel_err(t, ref), "absmax", float(t.abs().max()), float(ref.abs().max()))
