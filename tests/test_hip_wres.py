"""The weight-resident persistent convolution (csrc/conv_wres.hip, shape id flag 64) against the tiled kernel it replaces for
the 64-channel 3x3 layers (conv_bf16_kernel, mode f16x2) and against fp64: same arithmetic in the same order, so the two
kernels must agree BIT FOR BIT -- forward (BasicBlock conv with residual, head-conv column groups, train-mode statistics
through the model) and data gradient.  Reference layers: model/backbone/dla.py:12-51 (level2 BasicBlocks),
model/backbone/dla_neck.py:94-106 (ida_2 nodes' per-source data gradients), model/dense_heads/monocon_heads.py:114-131."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from hipmonocon import synth

pytestmark = pytest.mark.gpu
BASE = 6             # 128 px x 32 ch: a tiling every layer of the network accepts
WRES = 64 | BASE     # flag + the tiling every ineligible launch falls back to


@pytest.fixture(scope="module")
def eng():
    from hipmonocon.engine import Engine
    e = Engine()
    e.set_precision(3)
    yield e
    e.set_conv_cfg(0)


def rnd(seed, name, shape, std=1.0):
    return torch.from_numpy(synth.normalish(seed, name, shape, 0.0, std).astype(np.float32))


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


CASES = [
    # (name, B, H, W, Cout, residual, relu, affine)
    ("one_tile", 1, 4, 16, 64, False, False, False),
    ("two_tiles_row", 1, 4, 32, 64, True, True, True),
    ("odd_rows", 2, 6, 48, 64, True, True, True),          # last patch row half outside the image: not eligible, falls back
    ("two_rows_res", 2, 8, 48, 64, True, True, True),
    ("many_tiles", 3, 24, 80, 64, False, True, True),       # 90 tiles: several per workgroup, image seams inside a range
    ("head_576", 2, 8, 32, 576, False, False, True),        # nine column groups
    ("cout_96_padded", 1, 8, 16, 96, False, True, True),    # CoutP = 128: the second group is half padding
    # widths of an odd number of 8-column patches: the last tile of a row has its second patch outside the image
    ("ragged_one_and_a_half", 1, 8, 24, 64, True, True, True),
    ("ragged_half_tile", 2, 4, 8, 64, False, True, True),
    ("ragged_kitti_row", 1, 8, 312, 64, True, True, True),  # the quarter-resolution width of a 1248-wide KITTI frame
    ("ragged_head_576", 2, 8, 40, 576, False, False, True),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_wres_forward_is_bit_identical_and_fp32_accurate(eng, case):
    name, B, H, W, cout, use_res, relu, affine = case
    seed = 900 + CASES.index(case)
    x = rnd(seed, "x", (B, 64, H, W))
    w = rnd(seed, "w", (cout, 64, 3, 3), (2.0 / (9 * 64)) ** 0.5)
    scale = (1.0 + 0.1 * rnd(seed, "sc", (cout,))) if affine else None
    bias = 0.1 * rnd(seed, "bi", (cout,)) if affine else None
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    res = rnd(seed, "res", tuple(ref.shape)) if use_res else None
    if affine:
        ref = ref * scale.double()[None, :, None, None] + bias.double()[None, :, None, None]
    if use_res:
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    dev = eng.device
    args = ([nhwc(x).to(dev)], w.to(dev), 1, scale.to(dev) if affine else None, bias.to(dev) if affine else None,
            nhwc(res).to(dev) if use_res else None, relu)
    try:
        eng.set_conv_cfg(BASE)
        base = eng.op_conv(*args)
        eng.set_conv_cfg(WRES)
        got = eng.op_conv(*args)
    finally:
        eng.set_conv_cfg(0)
    assert torch.equal(base, got), name
    assert rel_err(got.cpu().permute(0, 3, 1, 2), ref) < 5e-6      # the f16x2 op-level bound (test_hip_bf16.py)


def test_wres_data_gradient_is_bit_identical(eng):
    """the per-source 64 -> 64 data gradient (transposed, flipped panel) incl. accumulation into an existing gradient"""
    dev = eng.device
    B, H, W = 2, 12, 48
    dy = nhwc(rnd(950, "dy", (B, 64, H, W))).to(dev)
    w = rnd(950, "w", (64, 128, 3, 3), (2.0 / (9 * 128)) ** 0.5).to(dev)       # a node conv: two 64-channel sources
    acc0 = nhwc(rnd(950, "g", (B, 64, H, W))).to(dev)
    outs = {}
    try:
        for cfg in (BASE, WRES):
            eng.set_conv_cfg(cfg)
            o = [eng.op_conv_dgrad(dy, w, (H, W), c_off=64 * s, cs=64, stride=1) for s in (0, 1)]
            a = acc0.clone()
            eng.op_conv_dgrad(dy, w, (H, W), c_off=64, cs=64, stride=1, accumulate_into=a)
            outs[cfg] = o + [a]
    finally:
        eng.set_conv_cfg(0)
    for b, g in zip(outs[BASE], outs[WRES]):
        assert torch.equal(b, g)
    ref = F.conv_transpose2d(dy.cpu().permute(0, 3, 1, 2).double(), w.cpu().double()[:, 64:], None, 1, 1)
    assert rel_err(outs[WRES][1].cpu().permute(0, 3, 1, 2), ref) < 5e-6


def test_wres_ineligible_launches_fall_back(eng):
    """the flag is a no-op for launches the kernel does not take (two sources, 128 channels, widths that are no multiple of 8)"""
    dev = eng.device
    try:
        for cins, cout, W in (([64, 64], 64, 32), ([128], 128, 32), ([64], 64, 20)):
            xs = [nhwc(rnd(960, "x%d" % i, (1, c, 8, W))).to(dev) for i, c in enumerate(cins)]
            w = rnd(960, "w", (cout, sum(cins), 3, 3), 0.05).to(dev)
            eng.set_conv_cfg(BASE)
            base = eng.op_conv(xs, w, 1, None, None, None, True)
            eng.set_conv_cfg(WRES)
            got = eng.op_conv(xs, w, 1, None, None, None, True)
            assert torch.equal(base, got)
    finally:
        eng.set_conv_cfg(0)


@pytest.mark.parametrize("width", (256, 224))
def test_wres_train_step_is_bit_identical_to_the_tiled_kernels(width):
    """whole train step (train-mode statistics partials, backward-statistics epilogue, all gradients) with the flag forced
    on every eligible layer vs forced off; width 224: 56-wide maps, rows of three and a half tiles"""
    from model import MonoConDetector
    import os
    stats = np.load(os.path.join(os.path.dirname(__file__), "golden", "bn_calib_seed7.npz"))
    sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
    batch = synth.make_conditioned_batch(21, 2, 96, width)
    gb = {"img": batch["img"].cuda(), "label": {k: v.cuda() for k, v in batch["label"].items()}, "img_metas": batch["img_metas"]}
    runs = []
    for cfg in (BASE, WRES):
        m = MonoConDetector(34, pretrained_backbone=False)
        m.load_state_dict(sd, strict=True)
        m = m.cuda().train().set_precision("f16x2")
        e = m._engine()
        e.set_conv_cfg(cfg)
        try:
            _, loss = m(gb)
            sum(loss.values()).backward()
            torch.cuda.synchronize()
            runs.append(({k: v.detach().cpu() for k, v in loss.items()},
                         {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters() if p.grad is not None},
                         {n: b.detach().cpu().clone() for n, b in m.named_buffers()}))
        finally:
            e.set_conv_cfg(0)
    (l0, g0, b0), (l1, g1, b1) = runs
    for k in l0:
        assert torch.equal(l0[k], l1[k]), k
    for n in g0:
        assert torch.equal(g0[n], g1[n]), n
    for n in b0:
        assert torch.equal(b0[n], b1[n]), n


def test_default_train_plan_keeps_the_epilogue_twins_on_the_tiled_kernels(monkeypatch, capfd):
    """the defaults of the two plan policies (mc_train_plan.hip): every BatchNorm whose input gradient is completed by a data
    gradient gets that launch's backward-statistics epilogue (MONOCON_HIP_BM_EPILOGUE default 1), and no backward launch
    takes the weight-resident kernel (MONOCON_HIP_WRES_BWD default 0: one-session A/B)"""
    import os
    from model import MonoConDetector
    monkeypatch.setenv("MONOCON_HIP_PLAN_DEBUG", "1")
    monkeypatch.delenv("MONOCON_HIP_BM_EPILOGUE", raising=False)
    monkeypatch.delenv("MONOCON_HIP_WRES_BWD", raising=False)
    batch = synth.make_batch(5, 2, 384, 1280)
    gb = {"img": batch["img"].cuda(), "label": {k: v.cuda() for k, v in batch["label"].items()}, "img_metas": batch["img_metas"]}
    m = MonoConDetector(34, pretrained_backbone=False).cuda().train().set_precision("f16x2")
    _, loss = m(gb)
    sum(loss.values()).backward()
    torch.cuda.synchronize()
    err = capfd.readouterr().err
    twins = [l for l in err.splitlines() if l.startswith("[plan]   twin of")]
    assert len(twins) >= 25, err[-2000:]          # 33 at this shape (rocprofv3: 33 launches of the <..., true> kernels per step)
    assert all(l.rstrip().endswith("wres 0") for l in twins), [l for l in twins if not l.rstrip().endswith("wres 0")]
