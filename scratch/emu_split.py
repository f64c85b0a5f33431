"""CPU emulation: is a 3-way bf16 split of both conv operands (6 partial products, fp32 accumulation) as close to
the fp64 reference as plain fp32?  Uses the oracle forward with F.conv2d monkeypatched."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "monocon-pytorch_amd")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch, torch.nn.functional as F
from hipmonocon import synth
from oracle import monocon_oracle as O
torch.set_num_threads(16)
stats = np.load("tests/golden/bn_calib_seed7.npz")
sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
img = synth.make_batch(8, 2, 128, 256, with_labels=False)["img"]
real_conv = F.conv2d
def split3(t):
    h = t.bfloat16().float(); r = t - h
    m = r.bfloat16().float(); r2 = r - m
    l = r2.bfloat16().float()
    return h, m, l
MODE = {"terms": None}
def emu_conv(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    if MODE["terms"] is None or x.dtype != torch.float32 or groups != 1 or w.shape[1] < 32:
        return real_conv(x, w, b, stride, padding, dilation, groups)
    xs, ws = split3(x), split3(w)
    out = None
    for (i, j) in MODE["terms"]:
        y = real_conv(xs[i], ws[j], None, stride, padding, dilation, groups)
        out = y if out is None else out + y
    if b is not None: out = out + b.view(1, -1, 1, 1)
    return out
O.F.conv2d = emu_conv
def rel(a, b): return float((a.double() - b).abs().max() / b.abs().max())
with torch.no_grad():
    ref64, _, _ = O.forward({k: (v.double() if v.dtype == torch.float32 else v) for k, v in sd.items()}, img.double())
    res = {}
    for name, terms in [("fp32", None), ("bf16x1", [(0, 0)]), ("bf16x3", [(0, 0), (0, 1), (1, 0)]),
                        ("bf16x6", [(0, 2), (2, 0), (1, 1), (0, 1), (1, 0), (0, 0)])]:
        MODE["terms"] = terms
        p, _, _ = O.forward(sd, img)
        res[name] = {k: rel(p[k], ref64[k]) for k in p}
for name, r in res.items():
    print("%-7s worst %.3e  median %.3e   %s" % (name, max(r.values()), float(np.median(list(r.values()))), {k[:12]: "%.1e" % v for k, v in r.items()}))
