import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "monocon-pytorch_amd")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch, torch.nn.functional as F
from hipmonocon import synth
from hipmonocon.engine import Engine
def rnd(seed, name, shape, scale=1.0): return torch.from_numpy((synth.normalish(seed, name, shape) * scale).astype(np.float32))
def nhwc(x): return x.permute(0, 2, 3, 1).contiguous()
eng = Engine()
for (B, H, W, cin, cout, k, s) in [(2, 24, 40, 64, 64, 3, 1), (2, 24, 40, 576, 64, 3, 1), (1, 16, 32, 128, 128, 3, 1), (1, 16, 32, 64, 128, 3, 2), (2, 16, 16, 256, 128, 1, 1)]:
    x = rnd(3, "x", (B, cin, H, W)); w = rnd(3, "w", (cout, cin, k, k), (2.0 / (k * k * cin)) ** 0.5)
    ref = F.conv2d(x.double(), w.double(), None, s, k // 2)
    for mode in (1, 2):
        eng.set_precision(mode)
        row = []
        for cfg in (1, 4, 5, 6, 7, 8):
            bnt = {1: 128, 4: 64, 5: 64, 6: 32, 7: 128, 8: 64}[cfg]
            coutp = (cout + (128 if cout >= 128 else 64 if cout > 32 else 32) - 1) // (128 if cout >= 128 else 64 if cout > 32 else 32) * (128 if cout >= 128 else 64 if cout > 32 else 32)
            if coutp % bnt: row.append("   -   "); continue
            eng.set_conv_cfg(cfg)
            try:
                got = eng.op_conv([nhwc(x).cuda()], w.cuda(), s).cpu().permute(0, 3, 1, 2)
                row.append("%.1e" % float((got.double() - ref).abs().max() / ref.abs().max()))
            except Exception as e:
                row.append("ERR")
        eng.set_conv_cfg(0)
        print("cin %3d cout %3d k%d s%d mode %d:" % (cin, cout, k, s, mode), " ".join(row))
