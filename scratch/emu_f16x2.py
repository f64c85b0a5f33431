"""CPU emulation of the fp16x2 operand split (h = fp16(s*x), l = fp16(s*x - h); products hh + hl + lh, fp32 accumulation;
per-tensor power-of-two scale s so that amax lands in [2^14, 2^15)) against the 3-way bf16 split and plain fp32, through
the oracle forward (F.conv2d monkeypatched), judged vs the fp64 run."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "monocon-pytorch_amd")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch, torch.nn.functional as F
from hipmonocon import synth
from oracle import monocon_oracle as O
torch.set_num_threads(8)
stats = np.load("tests/golden/bn_calib_seed7.npz")
sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
img = synth.make_batch(8, 2, 128, 256, with_labels=False)["img"]
real_conv = F.conv2d
def split3(t):
    h = t.bfloat16().float(); r = t - h
    m = r.bfloat16().float(); r2 = r - m
    return h, m, r2.bfloat16().float()
def pow2scale(t):
    am = float(t.abs().max())
    if am == 0: return 1.0
    return 2.0 ** (14 - int(np.floor(np.log2(am))))
def split_f16(t):
    s = pow2scale(t)
    x = t * s
    h = x.half().float()
    l = (x - h).half().float()
    return (h, l), s
MODE = {"m": None}
def emu_conv(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    if MODE["m"] is None or x.dtype != torch.float32 or groups != 1 or w.shape[1] < 32:
        return real_conv(x, w, b, stride, padding, dilation, groups)
    if MODE["m"] == "bf16x3":
        xs, ws = split3(x), split3(w)
        out = None
        for (i, j) in [(0, 2), (2, 0), (1, 1), (0, 1), (1, 0), (0, 0)]:
            y = real_conv(xs[i], ws[j], None, stride, padding, dilation, groups)
            out = y if out is None else out + y
    else:
        (xh, xl), sx = split_f16(x)
        (wh, wl), sw = split_f16(w)
        minor = real_conv(xl, wh, None, stride, padding, dilation, groups) + real_conv(xh, wl, None, stride, padding, dilation, groups)
        out = (real_conv(xh, wh, None, stride, padding, dilation, groups) + minor) * (1.0 / (sx * sw))
    if b is not None: out = out + b.view(1, -1, 1, 1)
    return out
O.F.conv2d = emu_conv
def rel(a, b): return float((a.double() - b).abs().max() / b.abs().max())
with torch.no_grad():
    ref64, _, _ = O.forward({k: (v.double() if v.dtype == torch.float32 else v) for k, v in sd.items()}, img.double())
    for name in (None, "bf16x3", "f16x2"):
        MODE["m"] = name
        p, _, _ = O.forward(sd, img)
        r = {k: rel(p[k], ref64[k]) for k in p}
        print("%-7s worst %.3e  median %.3e" % (name or "fp32", max(r.values()), float(np.median(list(r.values())))))
