import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "monocon-pytorch_amd")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch, torch.nn.functional as F
from hipmonocon import synth
from hipmonocon.engine import Engine
def rnd(seed, name, shape, scale=1.0): return torch.from_numpy((synth.normalish(seed, name, shape) * scale).astype(np.float32))
def nhwc(x): return x.permute(0, 2, 3, 1).contiguous()
B, H, W, cin, cout, k = 2, 24, 40, 64, 576, 3
w = rnd(1, "w", (cout, cin, k, k), (2.0 / (k * k * cin)) ** 0.5); dy = rnd(1, "dy", (B, cout, H, W))
x = torch.zeros(B, cin, H, W, dtype=torch.float64, requires_grad=True)
F.conv2d(x, w.double(), None, 1, 1).backward(dy.double()); ref = x.grad
eng = Engine()
for mode in (0, 1, 2):
    eng.set_precision(mode)
    got = eng.op_conv_dgrad(nhwc(dy).cuda(), w.cuda(), (H, W), 0, cin, 1).cpu().permute(0, 3, 1, 2)
    print("mode", mode, "head dgrad rel err", float((got.double() - ref).abs().max() / ref.abs().max()))
