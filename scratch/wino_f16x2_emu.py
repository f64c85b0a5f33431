"""VERDICT r4 item 6, the kill rule's accuracy half, on the CPU: Winograd F(2x2, 3x3) in the f16x2 arithmetic (input transform in
fp32, weights transformed in fp64 at pack time, both transformed operands scaled by a power of two and split into two fp16
pieces, three partial products accumulated in fp32, output transform in fp32) against the DIRECT f16x2 convolution, both vs
fp64, on the two named layers.  Rule: more than 2x the direct kernel's error (or > 5e-6, the op-level gate of
tests/test_hip_bf16.py::test_conv_split_emulation_is_fp32_accurate) kills it."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "monocon-pytorch_amd")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch, torch.nn.functional as F
from hipmonocon import synth
torch.set_num_threads(16)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)

def scale_exp(t):            # max |x| * 2^e in [2^14, 2^15)
    m = float(t.abs().max())
    return 0 if m == 0 else 14 - int(np.floor(np.log2(m)))
def split2(t32):             # fp32 -> (hi, lo) fp16 pieces, returned as fp32 tensors
    hi = t32.half().float()
    lo = (t32 - hi).half().float()
    return hi, lo
def mm3(a32, b32):           # three partial products of the split operands, fp32 accumulation: minor ones first
    ah, al = split2(a32); bh, bl = split2(b32)
    return (al @ bh + ah @ bl) + ah @ bh

def direct_f16x2(x, w):
    B, C, H, W = x.shape; O = w.shape[0]
    ea, ew = scale_exp(x), scale_exp(w)
    cols = F.unfold(x * 2.0 ** ea, 3, padding=1)                     # B, C*9, H*W  (k order: c, tap -- order only affects rounding)
    wm = (w * 2.0 ** ew).reshape(O, C * 9)
    out = torch.stack([mm3(wm, cols[b]) for b in range(B)]) * (2.0 ** -ea * 2.0 ** -ew)
    return out.reshape(B, O, H, W)

def wino_f16x2(x, w, transform64=False):
    B, C, H, W = x.shape; O = w.shape[0]
    xp = F.pad(x, (1, 1, 1, 1))
    t = xp.unfold(2, 4, 2).unfold(3, 4, 2)                           # B, C, th, tw, 4, 4
    bt = BT if transform64 else BT.float()
    tt = t.double() if transform64 else t
    V = torch.einsum("ij,bcthjk,lk->ilbcth", bt, tt, bt).float()     # input transform: adds / subtracts only
    U = torch.einsum("ij,ocjk,lk->iloc", G, w.double(), G)           # weights in fp64
    eu = scale_exp(U); ev = scale_exp(V)
    U32 = (U * 2.0 ** eu).float(); V32 = V * 2.0 ** ev
    th, tw = V.shape[4], V.shape[5]
    M = torch.empty(4, 4, B, O, th * tw)
    for i in range(4):
        for l in range(4):
            for b in range(B):
                M[i, l, b] = mm3(U32[i, l], V32[i, l, b].reshape(C, th * tw))
    M = M.reshape(4, 4, B, O, th, tw) * (2.0 ** -eu * 2.0 ** -ev)
    Y = torch.einsum("ij,jkbothw,lk->bothiwl".replace("w", "x"), AT.float(), M, AT.float()) if False else \
        torch.einsum("ij,jkbopq,lk->bopiql", AT.float(), M, AT.float())     # B, O, th, 2, tw, 2
    return Y.reshape(B, O, H, W)

def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())

for name, C, H, W in (("128->128 @48x160", 128, 48, 160), ("256->256 @24x80", 256, 24, 80)):
    for data in ("normal", "relu"):
        x = torch.from_numpy(synth.normalish(5, "x" + name, (1, C, H, W), 0.0, 1.0).astype(np.float32))
        if data == "relu":
            x = F.relu(x)
        w = torch.from_numpy(synth.normalish(6, "w" + name, (C, C, 3, 3), 0.0, (2.0 / (9 * C)) ** 0.5).astype(np.float32))
        ref = F.conv2d(x.double(), w.double(), None, 1, 1)
        e_fp32 = rel(F.conv2d(x, w, None, 1, 1), ref)
        e_dir = rel(direct_f16x2(x, w), ref)
        e_win = rel(wino_f16x2(x, w), ref)
        e_win64 = rel(wino_f16x2(x, w, transform64=True), ref)
        print("%-18s %-6s  fp32 (MKLDNN) %.2e   direct f16x2 %.2e   Winograd f16x2 %.2e (x%.1f)   with fp64 input transform %.2e"
              % (name, data, e_fp32, e_dir, e_win, e_win / e_dir, e_win64), flush=True)
