"""Mixed-precision mode (BASELINE config 3): bf16 MFMA operands, fp32 accumulation / activations / weights.
NOT the parity path -- there is no reference counterpart (the reference forces fp32); the
tolerance here is the bf16 operand rounding (2^-9 relative per operand) carried through the network,
stated per test, against the fp64 reference / the fp32 HIP path on identical inputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err, GOLDEN_SEED
from hipmonocon import synth

pytestmark = pytest.mark.gpu


@pytest.fixture()
def eng16():
    from hipmonocon.engine import Engine
    e = Engine()
    e.set_precision(1)
    yield e
    e.set_precision(0)


def rnd(seed, name, shape, scale=1.0):
    return torch.from_numpy((synth.normalish(seed, name, shape) * scale).astype(np.float32))


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


CASES = [
    # (name, B, H, W, [Cin...], Cout, k, stride, residual, relu)
    ("b16_s1_64_64", 2, 16, 24, [64], 64, 3, 1, True, True),
    ("b16_s1_128_128", 1, 12, 40, [128], 128, 3, 1, True, True),
    ("b16_s1_cat", 2, 8, 16, [64, 64], 64, 3, 1, False, True),
    ("b16_s1_odd_edges", 1, 6, 10, [32], 64, 3, 1, False, False),
    ("b16_s2_32_64", 2, 16, 32, [32], 64, 3, 2, False, True),
    ("b16_s2_256_512", 1, 8, 8, [256], 512, 3, 2, False, True),
    ("b16_k1_root4", 1, 12, 16, [128, 128, 64, 128], 128, 1, 1, False, True),
    ("b16_head_576", 1, 8, 16, [64], 576, 3, 1, False, False),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_bf16_operands(eng16, case):
    name, B, H, W, cins, cout, k, stride, use_res, relu = case
    seed = 700 + CASES.index(case)
    xs = [rnd(seed, "x%d" % i, (B, c, H, W)) for i, c in enumerate(cins)]
    w = rnd(seed, "w", (cout, sum(cins), k, k), (2.0 / (k * k * sum(cins))) ** 0.5)
    bias = 0.1 * rnd(seed, "bi", (cout,))
    res = rnd(seed, "res", (B, cout, (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1)) if use_res else None

    def ref(xcat, ww):
        r = F.conv2d(xcat, ww, None, stride, k // 2) + bias.double()[None, :, None, None]
        if use_res:
            r = r + res.double()
        return F.relu(r) if relu else r
    exact = ref(torch.cat(xs, 1).double(), w.double())
    # the same computation with both operands rounded to bf16 first: what the kernel is specified to do
    rounded = ref(torch.cat(xs, 1).bfloat16().double(), w.bfloat16().double())
    dev = eng16.device
    got = eng16.op_conv([nhwc(x).to(dev) for x in xs], w.to(dev), stride, None, bias.to(dev),
                        nhwc(res).to(dev) if use_res else None, relu).cpu().permute(0, 3, 1, 2)
    assert rel_err(got, rounded) < 2e-5          # fp32 accumulation of exactly-rounded operands
    e = rel_err(got, exact)
    assert 1e-5 < e < 1e-2, e                    # bf16 rounding is there, and only that


def l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def test_forward_bf16_close_to_fp32(golden_sd):
    """eval forward at 64x128 against the fp32 HIP path on the same inputs.  Every conv contributes ~0.3 %
    relative-L2 of operand rounding and this synthetic (untrained, BN-calibrated) network amplifies a
    perturbation roughly 1.4x per DLA level (the same conditioning shows in the fp32 parity numbers, see
    DESIGN.md section 4), so the bounds are stated per depth: level2 < 2e-2, predictions < 0.3 relative L2."""
    from hipmonocon.engine import Engine
    eng = Engine()
    eng.set_precision(0)          # (the suite may run under MONOCON_HIP_PRECISION=...)
    dsd = {k: v.cuda() for k, v in golden_sd.items()}
    img = synth.make_batch(GOLDEN_SEED + 1, 2, 64, 128, with_labels=False)["img"].cuda()
    eng.bind_state(dsd)
    lv32 = [t.clone() for t in eng.backbone_forward(img)]
    p32 = {k: v.clone() for k, v in eng.forward_infer(img).items()}
    eng.set_precision(1)
    eng.bind_state(dsd)
    lv16 = eng.backbone_forward(img)
    assert torch.equal(lv16[0], lv32[0]) and torch.equal(lv16[1], lv32[1])   # 16-channel layers stay fp32
    e2 = l2(lv16[2], lv32[2])
    assert 1e-4 < e2 < 2e-2, e2
    p16 = eng.forward_infer(img)
    worst = max(l2(p16[k], p32[k]) for k in p32)
    assert 1e-4 < worst < 0.3, worst
    eng.set_precision(0)
    eng.bind_state(dsd)
    back = eng.forward_infer(img)
    for k in p32:
        assert torch.equal(back[k], p32[k]), k     # switching back restores the fp32 path bit for bit


def test_train_step_bf16_runs_and_tracks_fp32(golden_sd):
    """full train step with bf16 conv operands: finite; losses within 25 % of the fp32 step (depth: factor 4)."""
    from model import MonoConDetector
    batch = synth.make_batch(GOLDEN_SEED + 3, 2, 96, 160)
    batch = {"img": batch["img"].cuda(), "label": {k: v.cuda() for k, v in batch["label"].items()},
             "img_metas": batch["img_metas"]}
    out = {}
    for mode in ("fp32", "bf16"):
        m = MonoConDetector(34, pretrained_backbone=False)
        m.load_state_dict(golden_sd, strict=True)
        m = m.cuda().train().set_precision(mode)
        _, loss = m(batch)
        # the depth (exp(-log_var)) and keypoint-heat-map terms are chaotic on this fixture: in pure fp32 a 1e-4 relative
        # weight perturbation already changes the kpt-heat-map head's gradient norm 3.7x (scratch/dbg_sens.py)
        total = sum(v for k, v in loss.items() if k not in ("loss_depth", "loss_kpt_heatmap"))
        total.backward()
        g = {n: p.grad.detach().double().flatten().clone() for n, p in m.named_parameters() if p.grad is not None}
        assert bool(torch.isfinite(total)) and all(bool(torch.isfinite(v).all()) for v in g.values())
        out[mode] = ({k: float(v.detach()) for k, v in loss.items()}, g)
    print({k: (round(v, 4), round(out["bf16"][0][k], 4)) for k, v in out["fp32"][0].items()})
    for k, v in out["fp32"][0].items():
        w = out["bf16"][0][k]
        if k == "loss_depth":      # exp(-log_var)-weighted: the synthetic depth head is the worst-conditioned output
            assert v / 4 < w < v * 4, (k, v, w)
        else:
            assert abs(w - v) <= 0.25 * abs(v) + 1e-3, (k, v, w)
    # B=2 train-mode BN on an untrained network is chaotic (fp32-vs-fp64 already moves backbone gradients by
    # 1e-3 for a 1e-7 perturbation) and a few tiny attention-weight gradients swing by orders of
    # magnitude, so the comparison is per conv weight tensor: a 4e-3 operand rounding leaves heads at cos 0.75-0.95,
    # neck 0.8-0.87, backbone 0.63-0.77; a wrong tap / panel in the bf16 gradient path would drive everything
    # upstream of it to ~0.  (The bf16 data- and weight-gradient kernels are checked exactly at op level.)
    cs = []
    for n, a in out["fp32"][1].items():
        if n.endswith("weight") and a.numel() >= 2048 and "attn" not in n:
            b = out["bf16"][1][n]
            cs.append(float(torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-300)))
    cs = sorted(cs)
    print("per-tensor gradient cosine: min %.3f  10%% %.3f  median %.3f" % (cs[0], cs[len(cs) // 10], cs[len(cs) // 2]))
    assert cs[len(cs) // 2] > 0.55 and cs[len(cs) // 10] > 0.3, (cs[0], cs[len(cs) // 10], cs[len(cs) // 2])


WG_CASES = [
    # (name, B, H, W, [Cin...], Cout, k)
    ("b16wg_64_64", 2, 16, 24, [64], 64, 3),
    ("b16wg_128_128", 2, 12, 40, [128], 128, 3),
    ("b16wg_cat_64_64", 2, 8, 16, [64, 64], 64, 3),
    ("b16wg_odd_edges", 1, 6, 10, [32], 64, 3),
    ("b16wg_head_64_576", 1, 8, 16, [64], 576, 3),
    ("b16wg_k1_root4", 1, 12, 16, [128, 128, 64, 128], 128, 1),
    ("b16wg_k1_32_64", 2, 8, 8, [32], 64, 1),
    ("b16wg_256_256_12x24", 2, 12, 24, [256], 256, 3),
    ("b16wg_128_64", 1, 8, 16, [128], 64, 3),
    ("b16wg_odd_64_128", 1, 6, 10, [64], 128, 3),
    ("b16wg_cat_64_64_128", 1, 10, 20, [64, 64], 128, 3),
]


@pytest.mark.parametrize("case", WG_CASES, ids=[c[0] for c in WG_CASES])
def test_wgrad_bf16_operands(eng16, case):
    """weight gradient with both operands (dY, X) rounded to bf16, fp32 accumulation over the pixels."""
    name, B, H, W, cins, cout, k = case
    seed = 800 + WG_CASES.index(case)
    xs = [rnd(seed, "x%d" % i, (B, c, H, W)) for i, c in enumerate(cins)]
    dy = rnd(seed, "dy", (B, cout, H, W))

    def ref(xcat, d):
        w = torch.zeros(cout, sum(cins), k, k, dtype=torch.float64, requires_grad=True)
        F.conv2d(xcat, w, None, 1, k // 2).backward(d)
        return w.grad
    exact = ref(torch.cat(xs, 1).double(), dy.double())
    rounded = ref(torch.cat(xs, 1).bfloat16().double(), dy.bfloat16().double())
    dev = eng16.device
    got = eng16.op_conv_wgrad([nhwc(x).to(dev) for x in xs], nhwc(dy).to(dev), k, 1).cpu()
    assert got.shape == exact.shape
    assert rel_err(got, rounded) < 2e-5
    e = rel_err(got, exact)
    assert 1e-5 < e < 2e-2, e


# ----------------------------------------------------------------------------------------------- mode 2: fp32 emulation
@pytest.fixture(params=[2, 3], ids=["bf16x3", "f16x2"])
def eng_split(request):
    """the two fp32-emulation modes: 2 = 3-way bf16 split (six partial products), 3 = 2-way fp16 split of the
    power-of-two-scaled operands (three partial products)"""
    from hipmonocon.engine import Engine
    e = Engine()
    e.set_precision(request.param)
    e.split_mode = request.param
    yield e
    e.set_precision(0)


@pytest.mark.parametrize("case", CASES, ids=[c[0].replace("b16", "split") for c in CASES])
def test_conv_split_emulation_is_fp32_accurate(eng_split, case):
    """3-way bf16 split of both operands, six partial products, fp32 accumulation: held to the SAME tolerance as
    the fp32 MFMA kernel (tests/test_hip_ops.py: 5e-6 norm-wise vs fp64), and agrees with that kernel to 2e-6."""
    name, B, H, W, cins, cout, k, stride, use_res, relu = case
    seed = 900 + CASES.index(case)
    xs = [rnd(seed, "x%d" % i, (B, c, H, W)) for i, c in enumerate(cins)]
    w = rnd(seed, "w", (cout, sum(cins), k, k), (2.0 / (k * k * sum(cins))) ** 0.5)
    bias = 0.1 * rnd(seed, "bi", (cout,))
    res = rnd(seed, "res", (B, cout, (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1)) if use_res else None
    ref = F.conv2d(torch.cat(xs, 1).double(), w.double(), None, stride, k // 2) + bias.double()[None, :, None, None]
    if use_res:
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    dev = eng_split.device
    args = ([nhwc(x).to(dev) for x in xs], w.to(dev), stride, None, bias.to(dev), nhwc(res).to(dev) if use_res else None, relu)
    got = eng_split.op_conv(*args).cpu()
    eng_split.set_precision(0)
    base = eng_split.op_conv(*args).cpu()
    eng_split.set_precision(eng_split.split_mode)
    assert rel_err(got.permute(0, 3, 1, 2), ref) < 5e-6
    assert rel_err(got, base) < 2e-6


THIN_CASES = [
    # 16 -> 16, 3x3, stride 1: the full-resolution layers.  (name, B, H, W, residual, relu, scale of x)
    ("two_strips", 2, 16, 128, True, True, 1.0),
    ("ragged_rows_and_strip", 1, 19, 48, False, True, 1.0),         # H % 8 != 0, W < one 64-pixel strip
    ("three_strips_last_partial", 1, 8, 144, False, False, 1.0),
    ("tiny_values", 1, 9, 64, True, False, 1e-12),
    ("one_row", 1, 1, 32, False, True, 3e4),
]


@pytest.mark.parametrize("case", THIN_CASES, ids=[c[0] for c in THIN_CASES])
def test_thin16_conv_on_the_fp16_pipe(case):
    """csrc/conv_thin.hip (mode 3 only: the LDS-staged fp16-split kernel that replaces the fp32 row kernel for the 16-channel
    3x3 layers): fp32-accurate vs fp64 (5e-6, the op-level gate of every fp32 path) and within 2e-6 of the fp32 row kernel"""
    from hipmonocon.engine import Engine
    name, B, H, W, use_res, relu, xs_scale = case
    seed = 1300 + THIN_CASES.index(case)
    x = rnd(seed, "x", (B, 16, H, W)) * xs_scale
    w = rnd(seed, "w", (16, 16, 3, 3), (2.0 / (9 * 16)) ** 0.5)
    bias = 0.1 * xs_scale * rnd(seed, "bi", (16,))
    res = xs_scale * rnd(seed, "res", (B, 16, H, W)) if use_res else None
    ref = F.conv2d(x.double(), w.double(), None, 1, 1) + bias.double()[None, :, None, None]
    if use_res:
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    eng = Engine()
    try:
        args = ([nhwc(x).cuda()], w.cuda(), 1, None, bias.cuda(), nhwc(res).cuda() if use_res else None, relu)
        eng.set_precision(3)
        got = eng.op_conv(*args).cpu()
        eng.set_precision(0)
        base = eng.op_conv(*args).cpu()
    finally:
        eng.close()
    assert rel_err(got.permute(0, 3, 1, 2), ref) < 5e-6
    assert rel_err(got, base) < 2e-6


@pytest.mark.parametrize("case", [(2, 16, 128, 1.0, 1.0), (1, 19, 48, 1.0, 1.0), (1, 8, 144, 1e-10, 3e3), (1, 5, 260, 50.0, 1e-6), (3, 4, 8, 1.0, 1.0)],
                         ids=["one_tile_wide", "ragged", "tiny_x_large_dy", "three_tiles", "smaller_than_a_tile"])
def test_thin16_wgrad_on_the_fp16_pipe(case):
    """csrc/conv_thin.hip wgrad_thin16_kernel (mode 3): weight gradient of the 16 -> 16 3x3 layer vs autograd in fp64 at the
    fp32 kernels' gate (5e-6), and within 2e-6 of the fp32 row kernel it replaces"""
    from hipmonocon.engine import Engine
    B, H, W, xmag, dmag = case
    x = rnd(1500 + H, "x", (B, 16, H, W)) * xmag
    dy = rnd(1500 + H, "dy", (B, 16, H, W)) * dmag
    w = torch.zeros(16, 16, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w, None, 1, 1).backward(dy.double())
    eng = Engine()
    try:
        eng.set_precision(3)
        got = eng.op_conv_wgrad([nhwc(x).cuda()], nhwc(dy).cuda(), 3, 1).cpu()
        eng.set_precision(0)
        base = eng.op_conv_wgrad([nhwc(x).cuda()], nhwc(dy).cuda(), 3, 1).cpu()
    finally:
        eng.close()
    assert rel_err(got, w.grad) < 5e-6, rel_err(got, w.grad)
    assert rel_err(got, base) < 2e-6


@pytest.mark.parametrize("case", [(2, 16, 64, 1.0), (1, 33, 65, 1.0), (1, 8, 200, 1e-9), (1, 21, 130, 2e4), (1, 5, 7, 1.0)],
                         ids=["one_tile", "ragged_33x65", "tiny_values_4_tiles", "large_values", "smaller_than_the_halo"])
def test_stem_on_the_fp16_pipe(case):
    """csrc/stem_f16.hip (mode 3): the 7x7 stem with folded BN + ReLU vs fp64, at the op-level gate of the fp32 kernel (5e-6
    norm-wise) and within 2e-6 of it; the per-workgroup operand scale makes the image's magnitude irrelevant"""
    from hipmonocon.engine import Engine
    B, H, W, mag = case
    seed = 1400 + H
    x = rnd(seed, "x", (B, 3, H, W)) * mag
    w = rnd(seed, "w", (16, 3, 7, 7), (2.0 / 147) ** 0.5)
    sc = 1.0 + 0.1 * rnd(seed, "sc", (16,))
    bi = 0.2 * mag * rnd(seed, "bi", (16,))
    ref = F.relu(F.conv2d(x.double(), w.double(), None, 1, 3) * sc.double()[None, :, None, None] + bi.double()[None, :, None, None])
    eng = Engine()
    try:
        eng.set_precision(3)
        got = eng.op_stem(x.cuda(), w.cuda(), sc.cuda(), bi.cuda()).cpu()
        eng.set_precision(0)
        base = eng.op_stem(x.cuda(), w.cuda(), sc.cuda(), bi.cuda()).cpu()
    finally:
        eng.close()
    assert rel_err(got.permute(0, 3, 1, 2), ref) < 5e-6
    assert rel_err(got, base) < 2e-6


@pytest.mark.parametrize("mode", [2, 3], ids=["bf16x3", "f16x2"])
def test_forward_split_emulation_meets_the_fp32_parity_gate(golden_sd, mode):
    """eval forward at 64x128 in modes 2 / 3 against the REFERENCE's fp64 golden: the same 1e-4 gate as the fp32 path."""
    from conftest import load_golden
    from hipmonocon.engine import Engine
    eng = Engine()
    eng.set_precision(mode)
    eng.bind_state({k: v.cuda() for k, v in golden_sd.items()})
    img = synth.make_batch(GOLDEN_SEED + 1, 2, 64, 128, with_labels=False)["img"].cuda()
    g = load_golden("fwd_small_eval.npz")
    pred = eng.forward_infer(img)
    worst = max(rel_err(v.cpu(), g["f64." + k]) for k, v in pred.items())
    assert worst < 1e-4, worst
    eng.set_precision(0)


@pytest.mark.parametrize("case", WG_CASES, ids=[c[0].replace("b16wg", "splitwg") for c in WG_CASES])
def test_wgrad_split_emulation_is_fp32_accurate(eng_split, case):
    """weight gradient in mode 2: same tolerance as the fp32 MFMA wgrad (tests/test_hip_ops.py: 5e-6 vs fp64)."""
    name, B, H, W, cins, cout, k = case
    seed = 950 + WG_CASES.index(case)
    xs = [rnd(seed, "x%d" % i, (B, c, H, W)) for i, c in enumerate(cins)]
    dy = rnd(seed, "dy", (B, cout, H, W))
    w = torch.zeros(cout, sum(cins), k, k, dtype=torch.float64, requires_grad=True)
    F.conv2d(torch.cat(xs, 1).double(), w, None, 1, k // 2).backward(dy.double())
    dev = eng_split.device
    got = eng_split.op_conv_wgrad([nhwc(x).to(dev) for x in xs], nhwc(dy).to(dev), k, 1).cpu()
    assert rel_err(got, w.grad) < 5e-6


@pytest.mark.parametrize("blocks", [1, 2, 3, 7])
@pytest.mark.parametrize("tiles", [1, 3], ids=["shared4w", "full8w"])
@pytest.mark.parametrize("case", [c for c in WG_CASES if c[6] == 3 and sum(c[4]) % 64 == 0 and c[5] in (64, 128, 256)],
                         ids=lambda c: c[0].replace("b16wg", "pipe"))
def test_wgrad_pipeline_long_pixel_loops(case, blocks, tiles, monkeypatch):
    """mode 3, stride 1, 3x3: the software-pipelined kernel (wgrad_pipe.hip) with so few workgroups that each one walks many
    pixel groups (odd and even counts, the two tile buffers and register sets in steady state), against autograd in fp64
    and against the two-barrier kernel it replaces."""
    from hipmonocon.engine import Engine
    name, B, H, W, cins, cout, k = case
    seed = 990 + WG_CASES.index(case)
    xs = [rnd(seed, "x%d" % i, (B, c, H, W)) for i, c in enumerate(cins)]
    dy = rnd(seed, "dy", (B, cout, H, W))
    w = torch.zeros(cout, sum(cins), k, k, dtype=torch.float64, requires_grad=True)
    F.conv2d(torch.cat(xs, 1).double(), w, None, 1, 1).backward(dy.double())
    eng = Engine()
    try:
        eng.set_precision(3)
        monkeypatch.setenv("MONOCON_HIP_WGRAD_PIPE_BLOCKS", str(blocks))
        monkeypatch.setenv("MONOCON_HIP_WGRAD_PIPE", str(tiles))   # 1: the 4-wave shared-CU tile; 3: every 8-wave tile
        got = eng.op_conv_wgrad([nhwc(x).cuda() for x in xs], nhwc(dy).cuda(), k, 1).cpu()
        monkeypatch.setenv("MONOCON_HIP_WGRAD_PIPE", "0")
        old = eng.op_conv_wgrad([nhwc(x).cuda() for x in xs], nhwc(dy).cuda(), k, 1).cpu()
        eng.set_precision(0)
    finally:
        eng.close()
    assert rel_err(got, w.grad) < 5e-6
    assert rel_err(got, old) < 2e-6


@pytest.mark.parametrize("mode,tol", [(1, 2e-2), (2, 5e-6), (3, 5e-6)], ids=["bf16", "split", "f16x2"])
@pytest.mark.parametrize("case", [(2, 16, 32, [32], 64), (1, 24, 48, [64], 128), (2, 8, 16, [128], 256), (1, 10, 24, [64], 64),
                                  (1, 12, 40, [256], 512), (1, 16, 16, [32, 32], 64)],
                         ids=["32_64", "64_128", "128_256", "ragged_patches", "256_512", "two_sources"])
def test_stride2_wgrad_on_the_bf16_pipe(mode, tol, case):
    """3x3 stride-2 weight gradient (round 3: the odd / even column planes of wgrad_bf16_kernel<3, 2, ...>) vs autograd in
    fp64, in all three modes; incl. output sizes that are not multiples of the 4 x 8 patch and a concatenated input"""
    from hipmonocon.engine import Engine
    B, H, W, cins, cout = case
    xs = [rnd(4321, "x%d" % i, (B, c, H, W)) for i, c in enumerate(cins)]
    dy = rnd(4321, "dy", (B, cout, H // 2, W // 2))
    w = torch.zeros(cout, sum(cins), 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(torch.cat(xs, 1).double(), w, None, 2, 1).backward(dy.double())
    eng = Engine()
    eng.set_precision(mode)
    try:
        got = eng.op_conv_wgrad([nhwc(x).cuda() for x in xs], nhwc(dy).cuda(), 3, 2).cpu()
    finally:
        eng.close()
    assert got.shape == w.grad.shape
    assert rel_err(got, w.grad) < tol, rel_err(got, w.grad)


@pytest.mark.parametrize("mode,tol", [(1, 2e-2), (2, 5e-6), (3, 5e-6)], ids=["bf16", "split", "f16x2"])
@pytest.mark.parametrize("case", [(2, 16, 24, 64, 0, 64, 64, 3, 1), (1, 8, 16, 192, 64, 64, 64, 3, 1),
                                  (2, 16, 32, 32, 0, 32, 64, 3, 2), (1, 24, 48, 64, 0, 64, 128, 3, 2),
                                  (1, 12, 16, 448, 256, 64, 128, 1, 1)],
                         ids=["s1", "s1_slice", "s2_32_64", "s2_64_128", "k1_slice"])
def test_dgrad_on_the_bf16_pipe(mode, tol, case):
    """data gradients (stride-1 panel and the stride-2 parity classes) in modes 1 and 2 vs autograd in fp64."""
    from hipmonocon.engine import Engine
    B, H, W, cin_total, c_off, cs, cout, k, stride = case
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    w = rnd(1234, "w", (cout, cin_total, k, k), (2.0 / (k * k * cin_total)) ** 0.5)
    dy = rnd(1234, "dy", (B, cout, Ho, Wo))
    x = torch.zeros(B, cin_total, H, W, dtype=torch.float64, requires_grad=True)
    F.conv2d(x, w.double(), None, stride, k // 2).backward(dy.double())
    ref = x.grad[:, c_off:c_off + cs]
    eng = Engine()
    eng.set_precision(mode)
    try:
        got = eng.op_conv_dgrad(nhwc(dy).cuda(), w.cuda(), (H, W), c_off, cs, stride)
    finally:
        eng.set_precision(0)
    e = rel_err(got.cpu().permute(0, 3, 1, 2), ref)
    assert e < tol, e
    if mode == 1:
        assert e > 1e-5       # the bf16 kernel really ran



# ----------------------------------------------------------------------------------------------- mode 3: operand scaling
@pytest.mark.parametrize("xscale,wscale", [(1e-7, 1.0), (3e5, 1.0), (1.0, 1e-6), (2e4, 5e3), (1e-20, 1e12)],
                         ids=["tiny_x", "huge_x", "tiny_w", "huge_both", "extreme"])
def test_f16x2_operand_scaling_covers_the_fp32_range(xscale, wscale):
    """the fp16 split scales every operand tensor by the power of two its max |x| dictates: tensors far outside fp16's
    own range (65504 / 6e-8) -- gradients of 1e-7, activations of 1e5 -- go through conv, weight gradient and data
    gradient at the tolerance of the fp32 kernels (5e-6 norm-wise vs fp64)."""
    from hipmonocon.engine import Engine
    B, H, W, cin, cout, k = 2, 16, 24, 64, 64, 3
    x = rnd(77, "x", (B, cin, H, W), xscale)
    w = rnd(77, "w", (cout, cin, k, k), wscale * (2.0 / (k * k * cin)) ** 0.5)
    dy = rnd(77, "dy", (B, cout, H, W), xscale)
    xd = x.double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    y = F.conv2d(xd, wd, None, 1, 1)
    y.backward(dy.double())
    eng = Engine()
    eng.set_precision(3)
    try:
        dev = eng.device
        got = eng.op_conv([nhwc(x).to(dev)], w.to(dev), 1, None, None, None, False).cpu()
        gw = eng.op_conv_wgrad([nhwc(x).to(dev)], nhwc(dy).to(dev), k, 1).cpu()
        gx = eng.op_conv_dgrad(nhwc(dy).to(dev), w.to(dev), (H, W), 0, cin, 1).cpu()
    finally:
        eng.set_precision(0)
    assert rel_err(got.permute(0, 3, 1, 2), y.detach()) < 5e-6
    assert rel_err(gw, wd.grad) < 5e-6
    assert rel_err(gx.permute(0, 3, 1, 2), xd.grad) < 5e-6


def test_retired_mode_4_is_an_error():
    """round 4's experimental mode 4 (mode 3 on pre-split, DMA-staged activations) is gone from the ABI: its weight gradients
    were not bit-reproducible beside the weight-gradient stream"""
    from hipmonocon.engine import Engine
    from hipmonocon.lib import MonoconHipError
    from model import MonoConDetector
    with pytest.raises(MonoconHipError, match="mode must be"):
        Engine().set_precision(4)
    with pytest.raises(ValueError, match="unknown precision mode"):
        MonoConDetector(34, pretrained_backbone=False).set_precision("f16x2p")
