"""Precision mode 4 ("f16x2p"): the f16x2 arithmetic on activations STORED as P16 (csrc/p16.h) and staged into LDS by DMA
(csrc/conv_p16.hip).  Op level, through the C-ABI, against plain torch fp64 -- same tolerance as every fp32-grade mode."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from hipmonocon import synth

pytestmark = pytest.mark.gpu
TOL = 5e-6


@pytest.fixture(scope="module")
def eng4():
    from hipmonocon.engine import Engine
    e = Engine()
    e.set_precision(4)
    yield e
    e.set_precision(0)


def rnd(seed, name, shape, std=1.0):
    return torch.from_numpy(synth.normalish(seed, name, shape, 0.0, std).astype(np.float32))


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


CASES = [
    # (name, B, H, W, [Cin...], [source scales], Cout, k, residual, relu)
    ("s1_64_64", 2, 16, 24, [64], [1.0], 64, 3, True, True),
    ("s1_128_128", 1, 12, 40, [128], [1.0], 128, 3, True, True),
    ("s1_cat_64_64", 2, 8, 16, [64, 64], [1.0, 1.0], 64, 3, False, True),
    ("s1_cat_scales", 2, 8, 16, [64, 64], [300.0, 0.004], 64, 3, False, False),      # per-source exponents differ by 2^16
    ("s1_odd_edges", 1, 6, 10, [64], [1.0], 64, 3, False, False),
    ("s1_one_chunk", 1, 9, 21, [16], [1.0], 64, 3, False, False),                      # a single 16-channel chunk
    ("s1_three_chunks", 1, 9, 21, [48], [1.0], 128, 3, False, True),                   # odd chunk count
    ("s1_tiny_2x4", 2, 2, 4, [256], [1.0], 512, 3, True, True),
    ("k1_root4", 1, 12, 16, [128, 128, 64, 128], [1.0, 2.0, 0.5, 8.0], 128, 1, False, True),
    ("k1_project", 2, 8, 8, [64], [1.0], 128, 1, False, False),
    ("k1_root3_512", 1, 4, 8, [512, 512, 256], [1.0, 1.0, 1.0], 512, 1, False, True),
    ("head_576", 1, 8, 16, [64], [1e-3], 576, 3, False, False),
    ("big_scale", 1, 8, 16, [64], [3e5], 64, 3, False, False),
    ("tiny_scale", 1, 8, 16, [64], [1e-20], 64, 3, False, False),
]
SHAPES = {4: "128x64", 5: "128x64m", 7: "64x128", 8: "64x64"}


@pytest.mark.parametrize("cfg", sorted(SHAPES), ids=[SHAPES[k] for k in sorted(SHAPES)])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_p16(eng4, case, cfg):
    name, B, H, W, cins, scales, cout, k, use_res, relu = case
    if cout % {4: 64, 5: 64, 7: 128, 8: 64}[cfg]:
        pytest.skip("tile does not divide the column count")
    seed = 700 + CASES.index(case)
    xs = [rnd(seed, "x%d" % i, (B, c, H, W)) * s for i, (c, s) in enumerate(zip(cins, scales))]
    w = rnd(seed, "w", (cout, sum(cins), k, k), (2.0 / (k * k * sum(cins))) ** 0.5)
    scale = 1.0 + 0.1 * rnd(seed, "sc", (cout,))
    bias = 0.1 * rnd(seed, "bi", (cout,)) * float(max(scales))
    ref = F.conv2d(torch.cat(xs, 1).double(), w.double(), None, 1, k // 2)
    ref = ref * scale.double()[None, :, None, None] + bias.double()[None, :, None, None]
    res = rnd(seed, "res", tuple(ref.shape)) * float(max(scales)) if use_res else None
    if use_res:
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    dev = eng4.device
    eng4.set_conv_cfg(cfg)
    try:
        out = eng4.op_conv([nhwc(x).to(dev) for x in xs], w.to(dev), 1, scale.to(dev), bias.to(dev),
                           nhwc(res).to(dev) if use_res else None, relu)
    finally:
        eng4.set_conv_cfg(0)
    got = out.cpu().permute(0, 3, 1, 2)
    assert got.shape == ref.shape
    assert rel_err(got, ref) < TOL, rel_err(got, ref)
