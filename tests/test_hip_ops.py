"""Op-level parity of the HIP kernels (through the C-ABI) against plain torch CPU ops in fp64.
GPU-only: run with ``-m gpu`` on the MI355X box."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from hipmonocon import synth

pytestmark = pytest.mark.gpu
TOL = 2e-6   # fp32 MFMA FMA chain vs fp64, norm-wise


@pytest.fixture(scope="module")
def eng():
    from hipmonocon.engine import Engine
    return Engine()


def rnd(seed, name, shape, std=1.0):
    return torch.from_numpy(synth.normalish(seed, name, shape, 0.0, std).astype(np.float32))


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


CONV_CASES = [
    # (name, B, H, W, [Cin...], Cout, k, stride, residual, relu, affine)
    ("s1_64_64", 2, 16, 24, [64], 64, 3, 1, True, True, True),
    ("s1_128_128_t128", 1, 12, 40, [128], 128, 3, 1, True, True, True),
    ("s1_16_16_t32", 1, 20, 24, [16], 16, 3, 1, False, True, True),
    ("s1_cat_64_64_to64", 2, 8, 16, [64, 64], 64, 3, 1, False, True, True),
    ("s1_odd_edges", 1, 6, 10, [32], 64, 3, 1, False, False, False),
    ("s1_tiny_2x4", 2, 2, 4, [256], 512, 3, 1, True, True, True),
    ("s2_32_64", 2, 16, 32, [32], 64, 3, 2, False, True, True),
    ("s2_16_32", 1, 24, 16, [16], 32, 3, 2, False, True, True),
    ("s2_256_512", 1, 8, 8, [256], 512, 3, 2, False, True, True),
    ("k1_root4", 1, 12, 16, [128, 128, 64, 128], 128, 1, 1, False, True, True),
    ("k1_project", 2, 8, 8, [32], 64, 1, 1, False, False, True),
    ("k1_root3_512", 1, 4, 8, [512, 512, 256], 512, 1, 1, False, True, True),
    ("head_576", 1, 8, 16, [64], 576, 3, 1, False, False, True),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv(eng, case):
    name, B, H, W, cins, cout, k, stride, use_res, relu, affine = case
    seed = 100 + CONV_CASES.index(case)
    xs = [rnd(seed, "x%d" % i, (B, c, H, W)) for i, c in enumerate(cins)]
    w = rnd(seed, "w", (cout, sum(cins), k, k), (2.0 / (k * k * sum(cins))) ** 0.5)
    scale = (1.0 + 0.1 * rnd(seed, "sc", (cout,))) if affine else None
    bias = 0.1 * rnd(seed, "bi", (cout,)) if affine else None
    ref = F.conv2d(torch.cat(xs, 1).double(), w.double(), None, stride, k // 2)
    res = rnd(seed, "res", tuple(ref.shape)) if use_res else None
    if affine:
        ref = ref * scale.double()[None, :, None, None] + bias.double()[None, :, None, None]
    if use_res:
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    dev = eng.device
    out = eng.op_conv([nhwc(x).to(dev) for x in xs], w.to(dev), stride,
                      scale.to(dev) if affine else None, bias.to(dev) if affine else None,
                      nhwc(res).to(dev) if use_res else None, relu)
    got = out.cpu().permute(0, 3, 1, 2)
    assert got.shape == ref.shape
    assert rel_err(got, ref) < TOL


WS_SHAPES = {17: 128, 20: 64, 21: 64, 22: 32, 23: 128, 24: 64}   # id (16 + shape) -> channel tile


@pytest.mark.parametrize("cfg", sorted(WS_SHAPES))
def test_conv_wave_specialised_is_bit_identical(eng, cfg):
    """the producer/consumer kernel keeps the accumulation order: bit-identical to the plain one."""
    dev = eng.device
    cases = [c for c in CONV_CASES if c[7] == 1]
    try:
        for case in cases:
            name, B, H, W, cins, cout, k, stride, use_res, relu, affine = case
            tile = 128 if cout >= 128 else (64 if cout > 32 else 32)
            if ((cout + tile - 1) // tile * tile) % WS_SHAPES[cfg]:
                continue
            seed = 300 + CONV_CASES.index(case)
            xs = [nhwc(rnd(seed, "x%d" % i, (B, c, H, W))).to(dev) for i, c in enumerate(cins)]
            w = rnd(seed, "w", (cout, sum(cins), k, k), (2.0 / (k * k * sum(cins))) ** 0.5).to(dev)
            bias = (0.1 * rnd(seed, "bi", (cout,))).to(dev)
            eng.set_conv_cfg(cfg & 15)
            base = eng.op_conv(xs, w, stride, None, bias, None, relu)
            eng.set_conv_cfg(cfg)
            got = eng.op_conv(xs, w, stride, None, bias, None, relu)
            assert torch.equal(base, got), name
    finally:
        eng.set_conv_cfg(0)


SMALL_CASES = [
    # (name, B, H, W, Cin, Cout, stride, residual, relu)
    ("small_16_16_s1", 2, 6, 32, 16, 16, 1, True, True),
    ("small_16_16_s1_wide", 1, 5, 80, 16, 16, 1, False, False),
    ("small_16_32_s2", 2, 8, 64, 16, 32, 2, False, True),
    ("small_16_32_s1", 1, 4, 48, 16, 32, 1, True, True),
    ("small_32_16_s1", 1, 7, 48, 32, 16, 1, True, False),
    ("small_32_16_s2", 1, 6, 96, 32, 16, 2, False, True),
]


@pytest.mark.parametrize("case", SMALL_CASES, ids=[c[0] for c in SMALL_CASES])
def test_conv_small_channel_kernel(eng, case):
    """the LDS-free 16x16x4 kernel for the thin full-resolution layers (cfg 32) vs fp64 and vs the generic kernel."""
    name, B, H, W, cin, cout, stride, use_res, relu = case
    seed = 500 + SMALL_CASES.index(case)
    x = rnd(seed, "x", (B, cin, H, W))
    w = rnd(seed, "w", (cout, cin, 3, 3), (2.0 / (9 * cin)) ** 0.5)
    scale = 1.0 + 0.1 * rnd(seed, "sc", (cout,))
    bias = 0.1 * rnd(seed, "bi", (cout,))
    ref = F.conv2d(x.double(), w.double(), None, stride, 1) * scale.double()[None, :, None, None] + bias.double()[None, :, None, None]
    res = rnd(seed, "res", tuple(ref.shape)) if use_res else None
    if use_res:
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    dev = eng.device
    args = ([nhwc(x).to(dev)], w.to(dev), stride, scale.to(dev), bias.to(dev), nhwc(res).to(dev) if use_res else None, relu)
    try:
        eng.set_conv_cfg(32)
        got = eng.op_conv(*args)
    finally:
        eng.set_conv_cfg(0)
    base = eng.op_conv(*args)
    assert rel_err(got.cpu().permute(0, 3, 1, 2), ref) < TOL
    assert rel_err(got.cpu(), base.cpu()) < 1e-6


def test_conv_is_transpose_sensitive(eng):
    """asymmetric one-hot weight: output channel n must read input channel (n*7)%C at tap (0,2)."""
    B, H, W, Cc = 1, 8, 8, 64
    x = rnd(1, "x", (B, Cc, H, W))
    w = torch.zeros(Cc, Cc, 3, 3)
    for n in range(Cc):
        w[n, (n * 7) % Cc, 0, 2] = 1.0
    ref = F.conv2d(x, w, None, 1, 1)
    got = eng.op_conv([nhwc(x).to(eng.device)], w.to(eng.device)).cpu().permute(0, 3, 1, 2)
    assert torch.equal(got, ref)


def test_stem(eng):
    x = rnd(3, "img", (2, 3, 40, 72))
    w = rnd(3, "w", (16, 3, 7, 7), 0.1)
    sc, bi = 1.0 + 0.1 * rnd(3, "s", (16,)), 0.1 * rnd(3, "b", (16,))
    ref = F.relu(F.conv2d(x.double(), w.double(), None, 1, 3) * sc.double()[None, :, None, None]
                 + bi.double()[None, :, None, None])
    d = eng.device
    got = eng.op_stem(x.to(d), w.to(d), sc.to(d), bi.to(d)).cpu().permute(0, 3, 1, 2)
    assert rel_err(got, ref) < TOL


def test_maxpool(eng):
    x = rnd(4, "x", (2, 32, 12, 20))
    got = eng.op_maxpool2(nhwc(x).to(eng.device)).cpu().permute(0, 3, 1, 2)
    assert torch.equal(got, F.max_pool2d(x, 2, 2))


def test_deconv(eng):
    x = rnd(5, "x", (2, 64, 6, 10))
    w = rnd(5, "w", (64, 1, 4, 4), 0.5)
    ref = F.conv_transpose2d(x.double(), w.double(), None, stride=2, padding=1, groups=64)
    got = eng.op_deconv4x4(nhwc(x).to(eng.device), w.to(eng.device)).cpu().permute(0, 3, 1, 2)
    assert rel_err(got, ref) < TOL


def test_layout_roundtrip(eng):
    x = rnd(6, "x", (2, 19, 7, 33)).to(eng.device)
    y = eng.to_nhwc(x)
    assert torch.equal(y.cpu(), x.cpu().permute(0, 2, 3, 1))
    assert torch.equal(eng.to_nchw(y).cpu(), x.cpu())


WGRAD_CASES = [
    # (name, B, H, W, [Cin...], Cout, k, stride)
    ("wg_s1_64_64", 2, 16, 24, [64], 64, 3, 1),
    ("wg_s1_128_128", 2, 12, 40, [128], 128, 3, 1),
    ("wg_s1_cat_64_64_to64", 2, 8, 16, [64, 64], 64, 3, 1),
    ("wg_s1_16_16", 1, 20, 24, [16], 16, 3, 1),
    ("wg_s1_odd", 1, 6, 10, [32], 64, 3, 1),
    ("wg_s2_32_64", 2, 16, 32, [32], 64, 3, 2),
    ("wg_s2_16_32", 1, 24, 16, [16], 32, 3, 2),
    ("wg_s2_256_512", 1, 8, 8, [256], 512, 3, 2),
    ("wg_k1_root4", 1, 12, 16, [128, 128, 64, 128], 128, 1, 1),
    ("wg_k1_32_64", 2, 8, 8, [32], 64, 1, 1),
    ("wg_head_64_576", 1, 8, 16, [64], 576, 3, 1),
    ("wg_s2_128_256_24x48", 2, 24, 48, [128], 256, 3, 2),
    ("wg_s1_256_256_12x24", 2, 12, 24, [256], 256, 3, 1),
    ("wg_s2_64_128_48x96", 2, 48, 96, [64], 128, 3, 2),
]


@pytest.mark.parametrize("case", WGRAD_CASES, ids=[c[0] for c in WGRAD_CASES])
def test_conv_wgrad(eng, case):
    name, B, H, W, cins, cout, k, stride = case
    seed = 300 + WGRAD_CASES.index(case)
    xs = [rnd(seed, "x%d" % i, (B, c, H, W)) for i, c in enumerate(cins)]
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    dy = rnd(seed, "dy", (B, cout, Ho, Wo))
    w = torch.zeros(cout, sum(cins), k, k, dtype=torch.float64, requires_grad=True)
    out = F.conv2d(torch.cat(xs, 1).double(), w, None, stride, k // 2)
    out.backward(dy.double())
    dev = eng.device
    got = eng.op_conv_wgrad([nhwc(x).to(dev) for x in xs], nhwc(dy).to(dev), k, stride).cpu()
    assert got.shape == w.grad.shape
    assert rel_err(got, w.grad) < 5e-6


DGRAD_CASES = [
    # (name, B, Hin, Win, CinTotal, c_off, Cs, Cout, k, stride)
    ("dg_s1_64_64", 2, 16, 24, 64, 0, 64, 64, 3, 1),
    ("dg_s1_128_128", 1, 12, 40, 128, 0, 128, 128, 3, 1),
    ("dg_s1_source_slice", 2, 8, 16, 192, 64, 64, 64, 3, 1),      # second source of a 64+64+64 concat
    ("dg_s1_odd_edges", 1, 6, 10, 32, 0, 32, 64, 3, 1),
    ("dg_k1_root_slice", 1, 12, 16, 448, 256, 64, 128, 1, 1),
    ("dg_s2_32_64", 2, 16, 32, 32, 0, 32, 64, 3, 2),              # four output-parity window convs
    ("dg_s2_64_128", 1, 24, 48, 64, 0, 64, 128, 3, 2),
    ("dg_s2_16_32", 1, 24, 32, 16, 0, 16, 32, 3, 2),
    ("dg_s2_256_512_tiny", 1, 8, 8, 256, 0, 256, 512, 3, 2),
    ("dg_head_576_64", 1, 8, 16, 64, 0, 64, 576, 3, 1),
]


@pytest.mark.parametrize("case", DGRAD_CASES, ids=[c[0] for c in DGRAD_CASES])
def test_conv_dgrad(eng, case):
    """data gradient through exactly the launches the train plan emits (transposed / flipped panel for stride 1,
    output-parity classes scattering into dx for stride 2) vs autograd in fp64; then the accumulate form."""
    name, B, H, W, cin_total, c_off, cs, cout, k, stride = case
    seed = 600 + DGRAD_CASES.index(case)
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    w = rnd(seed, "w", (cout, cin_total, k, k), (2.0 / (k * k * cin_total)) ** 0.5)
    dy = rnd(seed, "dy", (B, cout, Ho, Wo))
    x = torch.zeros(B, cin_total, H, W, dtype=torch.float64, requires_grad=True)
    F.conv2d(x, w.double(), None, stride, k // 2).backward(dy.double())
    ref = x.grad[:, c_off:c_off + cs]
    dev = eng.device
    tol = TOL if cout * k * k <= 4608 else 5e-6        # one fp32 FMA chain over K = Cout*k*k terms
    got = eng.op_conv_dgrad(nhwc(dy).to(dev), w.to(dev), (H, W), c_off, cs, stride)
    assert rel_err(got.cpu().permute(0, 3, 1, 2), ref) < tol
    base = rnd(seed, "acc", (B, cs, H, W))
    acc = nhwc(base).to(dev).clone()
    eng.op_conv_dgrad(nhwc(dy).to(dev), w.to(dev), (H, W), c_off, cs, stride, accumulate_into=acc)
    assert rel_err(acc.cpu().permute(0, 3, 1, 2), ref + base.double()) < tol

