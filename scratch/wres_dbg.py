import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "monocon-pytorch_amd"), REPO, os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch, torch.nn.functional as F
from hipmonocon import synth
from hipmonocon.engine import Engine
import test_hip_wres as T
e = Engine(); e.set_precision(3)
for case in T.CASES:
    name, B, H, W, cout, use_res, relu, affine = case
    seed = 900 + T.CASES.index(case)
    x = T.rnd(seed, "x", (B, 64, H, W)); w = T.rnd(seed, "w", (cout, 64, 3, 3), (2.0 / (9 * 64)) ** 0.5)
    scale = (1.0 + 0.1 * T.rnd(seed, "sc", (cout,))) if affine else None
    bias = 0.1 * T.rnd(seed, "bi", (cout,)) if affine else None
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    res = T.rnd(seed, "res", tuple(ref.shape)) if use_res else None
    if affine: ref = ref * scale.double()[None, :, None, None] + bias.double()[None, :, None, None]
    if use_res: ref = ref + res.double()
    if relu: ref = F.relu(ref)
    dev = e.device
    args = ([T.nhwc(x).to(dev)], w.to(dev), 1, scale.to(dev) if affine else None, bias.to(dev) if affine else None, T.nhwc(res).to(dev) if use_res else None, relu)
    outs = {}
    for cfg in (6, 4, 70, 68):
        e.set_conv_cfg(cfg)
        for rep in range(2):
            o = e.op_conv(*args).cpu().permute(0, 3, 1, 2).double()
            outs[(cfg, rep)] = o
    e.set_conv_cfg(0)
    def err(o): return float((o - ref).abs().max() / ref.abs().max())
    print(name, " ".join("cfg%d/%d err %.2e" % (k[0], k[1], err(v)) for k, v in outs.items()))
    d = (outs[(70, 0)] - outs[(6, 0)]).abs()
    if float(d.max()) > 0:
        nz = d.nonzero()
        print("   mismatches:", len(nz), "first", nz[:6].tolist(), "max", float(d.max()), "cols(channels) bad:", sorted(set(nz[:, 1].tolist()))[:20], "rows:", sorted(set(nz[:, 2].tolist())), "xs:", sorted(set(nz[:, 3].tolist()))[:40])
