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


DGRAD_CASES = [
    # (name, B, H, W, CinTotal, c_off, Cs, Cout, k, stride)
    ("dg_s1_64_64", 2, 16, 24, 64, 0, 64, 64, 3, 1),
    ("dg_s1_cat_slice", 1, 12, 16, 192, 64, 128, 128, 3, 1),
    ("dg_k1_root", 1, 12, 16, 448, 256, 64, 128, 1, 1),
    ("dg_s2_64_128", 2, 16, 32, 64, 0, 64, 128, 3, 2),          # the four output-parity windows 1x1 / 1x2 / 2x1 / 2x2
    ("dg_s2_256_512_tiny", 1, 8, 8, 256, 0, 256, 512, 3, 2),
    ("dg_head_576_64", 1, 8, 16, 64, 0, 64, 576, 3, 1),
]


@pytest.mark.parametrize("case", DGRAD_CASES, ids=[c[0] for c in DGRAD_CASES])
def test_conv_dgrad_p16(eng4, case):
    """data gradients in mode 4: dY stored as P16, the transposed / flipped panels as before; stride-2 layers exercise
    the 1x2 / 2x1 / 2x2 window builds of the DMA-staged kernel (few taps per chunk: the deeper tile ring)."""
    name, B, H, W, cin_total, c_off, cs, cout, k, stride = case
    seed = 800 + DGRAD_CASES.index(case)
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    w = rnd(seed, "w", (cout, cin_total, k, k), (2.0 / (k * k * cin_total)) ** 0.5)
    dy = rnd(seed, "dy", (B, cout, Ho, Wo))
    x = torch.zeros(B, cin_total, H, W, dtype=torch.float64, requires_grad=True)
    F.conv2d(x, w.double(), None, stride, k // 2).backward(dy.double())
    ref = x.grad[:, c_off:c_off + cs]
    dev = eng4.device
    got = eng4.op_conv_dgrad(nhwc(dy).to(dev), w.to(dev), (H, W), c_off, cs, stride)
    assert rel_err(got.cpu().permute(0, 3, 1, 2), ref) < TOL
    base = rnd(seed, "acc", (B, cs, H, W))
    acc = nhwc(base).to(dev).clone()
    eng4.op_conv_dgrad(nhwc(dy).to(dev), w.to(dev), (H, W), c_off, cs, stride, accumulate_into=acc)
    assert rel_err(acc.cpu().permute(0, 3, 1, 2), ref + base.double()) < TOL


WGRAD_CASES = [
    # (name, B, H, W, [Cin...], Cout, k, stride)
    ("wg_s1_64_64", 2, 16, 24, [64], 64, 3, 1),
    ("wg_s1_128_128", 1, 12, 40, [128], 128, 3, 1),
    ("wg_s1_cat_64_64", 2, 8, 16, [64, 64], 64, 3, 1),
    ("wg_s1_odd", 1, 6, 10, [64], 64, 3, 1),
    ("wg_s2_64_128", 2, 24, 48, [64], 128, 3, 2),
    ("wg_s2_256_512", 1, 8, 8, [256], 512, 3, 2),
    ("wg_k1_root4", 1, 12, 16, [128, 128, 64, 128], 128, 1, 1),
    ("wg_head_64_576", 1, 8, 16, [64], 576, 3, 1),
]


@pytest.mark.parametrize("operands", ["xd", "x", "d"])
@pytest.mark.parametrize("case", WGRAD_CASES, ids=[c[0] for c in WGRAD_CASES])
def test_conv_wgrad_p16(eng4, case, operands, monkeypatch):
    """weight gradients with X and / or dY stored as P16 (the train plan mixes them: level-1 activations and the head's
    hidden-map gradient stay fp32): the staging only transposes the stored pieces"""
    name, B, H, W, cins, cout, k, stride = case
    monkeypatch.setenv("MONOCON_HIP_P16_OPERANDS", operands)
    seed = 900 + WGRAD_CASES.index(case)
    xs = [rnd(seed, "x%d" % i, (B, c, H, W)) * (3.0 ** i) for i, c in enumerate(cins)]
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    dy = rnd(seed, "dy", (B, cout, Ho, Wo)) * 1e-3
    w = torch.zeros(cout, sum(cins), k, k, dtype=torch.float64, requires_grad=True)
    F.conv2d(torch.cat(xs, 1).double(), w, None, stride, k // 2).backward(dy.double())
    dev = eng4.device
    got = eng4.op_conv_wgrad([nhwc(x).to(dev) for x in xs], nhwc(dy).to(dev), k, stride).cpu()
    assert got.shape == w.grad.shape
    assert rel_err(got, w.grad) < TOL, rel_err(got, w.grad)
