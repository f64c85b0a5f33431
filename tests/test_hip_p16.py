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
SHAPES = {4: "128x64", 5: "128x64m", 6: "128x32", 7: "64x128", 8: "64x64"}


@pytest.mark.parametrize("cfg", sorted(SHAPES), ids=[SHAPES[k] for k in sorted(SHAPES)])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_p16(eng4, case, cfg):
    name, B, H, W, cins, scales, cout, k, use_res, relu = case
    if cout % {4: 64, 5: 64, 6: 32, 7: 128, 8: 64}[cfg]:
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


S2_CASES = [("s2_64_128", 2, 16, 32, 64, 128), ("s2_128_256", 1, 24, 16, 128, 256), ("s2_256_512", 1, 8, 8, 256, 512)]


@pytest.mark.parametrize("case", S2_CASES, ids=[c[0] for c in S2_CASES])
def test_conv_stride2_from_p16(eng4, case):
    """the three stride-2 3x3 layers with >= 64 input channels: the register-staged kernel copying stored pieces"""
    name, B, H, W, cin, cout = case
    seed = 950 + S2_CASES.index(case)
    x = rnd(seed, "x", (B, cin, H, W)) * 7.0
    w = rnd(seed, "w", (cout, cin, 3, 3), (2.0 / (9 * cin)) ** 0.5)
    ref = F.relu(F.conv2d(x.double(), w.double(), None, 2, 1))
    dev = eng4.device
    out = eng4.op_conv([nhwc(x).to(dev)], w.to(dev), 2, None, None, None, True)
    assert rel_err(out.cpu().permute(0, 3, 1, 2), ref) < TOL


@pytest.mark.parametrize("cfg", sorted(SHAPES), ids=[SHAPES[k] for k in sorted(SHAPES)])
def test_conv_p16_under_load(eng4, cfg):
    """a launch that fills the chip several times over (B = 8 at 96x320: 7680 patches): the DMA pipeline's waits are
    counted, not drained -- a wrong count shows as rare wrong tiles only when the memory system is busy"""
    B, H, W, cin, cout = 8, 96, 320, 64, 64 if cfg != 7 else 128
    x = rnd(990, "x", (B, cin, H, W))
    w = rnd(990, "w", (cout, cin, 3, 3), (2.0 / (9 * cin)) ** 0.5)
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    dev = eng4.device
    eng4.set_conv_cfg(cfg)
    try:
        outs = [eng4.op_conv([nhwc(x).to(dev)], w.to(dev), 1, None, None, None, False).cpu() for _ in range(3)]
    finally:
        eng4.set_conv_cfg(0)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert rel_err(outs[0].permute(0, 3, 1, 2), ref) < TOL


def test_mode4_train_step_tracks_mode3():
    """EXPERIMENTAL mode 4 inside the train plan (P16 activations / BatchNorm-input gradients of >= 64 channels, the
    DMA-staged convolution for every stride-1 layer that reads them): one small train step against mode 3 on the same
    batch -- losses to 1e-4 (the fp32 budget), every gradient finite, flat gradient within 2 % (two fp32-grade paths on a
    B=2 train-mode-BN fixture).  Mode 4 is NOT the headline and not part of the full-size parity suite: with
    the weight gradients on their second stream its backward is not bit-reproducible run to run."""
    import os
    from conftest import GOLDEN_SEED
    from model import MonoConDetector
    stats = np.load(os.path.join(os.path.dirname(__file__), "golden", "bn_calib_seed7.npz"))
    sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
    b = synth.make_batch(GOLDEN_SEED + 9, 2, 96, 320)
    batch = {"img": b["img"].cuda(), "label": {k: v.cuda() for k, v in b["label"].items()}, "img_metas": b["img_metas"]}
    out = {}
    for mode in ("f16x2", "f16x2p"):
        m = MonoConDetector(34, pretrained_backbone=False)
        m.load_state_dict(sd, strict=True)
        m = m.cuda().train().set_precision(mode)
        _, loss = m(batch)
        sum(loss.values()).backward()
        torch.cuda.synchronize()
        g = torch.cat([p.grad.flatten().double() for p in m.parameters() if p.grad is not None])
        assert bool(torch.isfinite(g).all())
        out[mode] = ({k: float(v.detach()) for k, v in loss.items()}, g)
    for k, v in out["f16x2"][0].items():
        assert abs(out["f16x2p"][0][k] - v) <= 1e-4 * abs(v) + 1e-6, (k, v, out["f16x2p"][0][k])
    e = float((out["f16x2"][1] - out["f16x2p"][1]).norm() / out["f16x2"][1].norm())
    assert e < 2e-2, e
