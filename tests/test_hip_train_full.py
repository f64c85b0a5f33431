"""Training parity ON THE HEADLINE WORKLOAD, in every fp32-grade precision mode the library offers.

tests/golden/train_full.npz is the real reference (model/detector/monocon_detector.py:53-61 + total.backward(),
engine/monocon_engine.py:84-86) run in float64 at B=2, 384x1280, train mode, on conditioned parameters / images
(tests/golden/make_golden.py train_full): ten losses, updated BatchNorm buffers, strided samples of the ten prediction
maps, and per-tensor gradient norms + strided samples for all 236 live parameter tensors.

What can be asserted at this size.  2 x 16 x 384 x 1280 activations cannot be screened free of ReLU / max-pool decision
flips (expected flips ~ N * e / sigma >> 1): the reference's OWN fp32 run sits up to 2.4e-2 (median
3.8e-4) from its fp64 run on this fixture, and its fp64 gradients move by up to 1.7e-3 (median 7e-5) under a 3e-7 image
perturbation.  Both yard-sticks are recorded PER TENSOR in the fixture (gerr32.*, gmargin.*).  WHICH tensor a flip
lands on differs between two correct fp32 implementations, so the bounds are on the per-section DISTRIBUTION of the
errors (check_gradients below): median, 90th percentile and worst tensor against the same statistics of the reference's
own fp32 run.  Losses, buffers and predictions are held to 1e-4 against fp64 -- flips do not move them.

The B=32 test is the size-independent property at BASELINE's batch: the pair repeated 16 times has the same batch
statistics and per-object losses, so losses / gradients equal the B=2 ones -- checked in EVERY mode (round 2 ran it in the
native fp32 mode only), and the B=32 gradients are held to the fp64 golden directly as well."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err, grad_rel_l2
from hipmonocon import netspec, synth

pytestmark = pytest.mark.gpu

PRECISIONS = ("fp32", "bf16x3", "f16x2")
LOSS_TOL = 1e-4


def build(sd, precision):
    from model import MonoConDetector
    m = MonoConDetector(34, pretrained_backbone=False)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    m.set_precision(precision)
    return m


def to_cuda(batch):
    d = dict(batch)
    d["img"] = batch["img"].cuda()
    d["label"] = {k: v.cuda() for k, v in batch["label"].items()}
    return d


def check_gradients(m, g, tag):
    """per section (head / neck / backbone): the distribution of the HIP path's per-tensor errors against the fp64
    golden is held to the distribution of the reference's own fp32 run on the same fixture -- median within 2x (+1e-4),
    90th percentile within 3x (+1e-3), worst tensor within 3x the reference's worst (+5e-3: a decision flip the
    reference's run happened not to have -- measured: 2.9e-3 on one 256-element BatchNorm bias of the neck in the native fp32
    mode, where the reference's fp32 run stays at 3e-4)."""
    rows = []
    for n, p in m.named_parameters():
        if n in netspec.DEAD_PARAMS:
            assert p.grad is None, n
            continue
        e = grad_rel_l2(p.grad, g["g64." + n], g["gnorm64." + n], p.numel())
        norm64 = float(g["gnorm64." + n])
        rows.append((n.split(".")[0], n, e, float(g["gerr32." + n]),
                     abs(float(p.grad.double().norm()) - norm64) / max(norm64, 1e-30)))
    assert len(rows) == 236
    fails = []
    for sec in ("head", "neck", "backbone"):
        sel = [r for r in rows if r[0] == sec]
        hip, ref = np.array([r[2] for r in sel]), np.array([r[3] for r in sel])
        worst = max(sel, key=lambda r: r[2])
        print("%s %-9s %3d tensors: hip-vs-fp64 median %.2e q90 %.2e max %.2e (%s) | reference fp32-vs-fp64 median %.2e q90 %.2e "
              "max %.2e" % (tag, sec, len(sel), np.median(hip), np.quantile(hip, 0.9), hip.max(), worst[1], np.median(ref),
                            np.quantile(ref, 0.9), ref.max()))
        if not np.median(hip) <= 2.0 * np.median(ref) + 1e-4: fails.append((sec, "median", float(np.median(hip))))
        if not np.quantile(hip, 0.9) <= 3.0 * np.quantile(ref, 0.9) + 1e-3: fails.append((sec, "q90", float(np.quantile(hip, 0.9))))
        if not hip.max() <= 3.0 * ref.max() + 5e-3: fails.append((sec, "max", worst))
        nerr = max(r[4] for r in sel)
        if not nerr <= 3.0 * ref.max() + 5e-3: fails.append((sec, "norm", nerr))
    assert not fails, fails


@pytest.mark.parametrize("precision", PRECISIONS)
def test_full_size_train_step_vs_reference_fp64(cond_sd, precision):
    g = load_golden("train_full.npz")
    B, H, W = (int(x) for x in g["shape"])
    assert (B, H, W) == (2, 384, 1280)
    m = build(cond_sd, precision)
    pred, loss = m(to_cuda(synth.make_conditioned_batch(int(g["seed"]), B, H, W)))
    sum(loss.values()).backward()
    torch.cuda.synchronize()
    for k, v in loss.items():
        ref = float(g["f64." + k])
        assert abs(float(v.detach()) - ref) <= LOSS_TOL * abs(ref) + 1e-7, (k, float(v.detach()), ref)
    for k, v in pred.items():
        ref = g["pred64." + k]
        step = max(1, v.numel() // 4096)
        e = rel_err(v.detach().cpu().reshape(-1)[::step], ref)
        assert e < 1e-4, (k, e)
    sd = m.state_dict()
    for k in sd:
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert rel_err(sd[k].cpu(), g["buf64." + k]) < 1e-4, k
    check_gradients(m, g, "B=2 %s" % precision)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_b32_train_step_is_the_full_size_pair_times_16(cond_sd, precision):
    """BASELINE configs[2] shape in the mode under test: losses equal the fp64 reference of the pair at 1e-4 (they are
    means over objects / pixels: 16 copies change nothing), the flat gradient equals the B=2 gradient of the same mode
    to 1e-3 relative L2, and every tensor meets the fp64 golden under the same per-tensor bounds as at B=2."""
    g = load_golden("train_full.npz")
    b2 = synth.make_conditioned_batch(int(g["seed"]), 2, 384, 1280)
    m2 = build(cond_sd, precision)
    _, l2 = m2(to_cuda(b2))
    sum(l2.values()).backward()
    g2 = torch.cat([p.grad.flatten() for p in m2.parameters() if p.grad is not None]).clone()
    del m2, l2
    b32 = {"img": b2["img"].repeat(16, 1, 1, 1).cuda(),
           "label": {k: v.repeat(16, *([1] * (v.dim() - 1))).cuda() for k, v in b2["label"].items()},
           "img_metas": {"pad_shape": [(384, 1280)] * 32}}
    m = build(cond_sd, precision)
    _, l32 = m(b32)
    sum(l32.values()).backward()
    torch.cuda.synchronize()
    for k, v in l32.items():
        ref = float(g["f64." + k])
        assert abs(float(v.detach()) - ref) <= LOSS_TOL * abs(ref) + 1e-7, (k, float(v.detach()), ref)
    g32 = torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None])
    assert bool(torch.isfinite(g32).all())
    e = float((g32.double() - g2.double()).norm() / g2.double().norm())
    print("B=32 vs B=2 flat gradient (%s): rel L2 %.2e" % (precision, e))
    assert e < 1e-3, e
    sd = m.state_dict()
    # (running_var carries the unbiased n/(n-1) factor, which depends on the batch size: not compared here)
    for k in sd:
        if k.endswith("running_mean"):
            assert rel_err(sd[k].cpu(), g["buf64." + k]) < 1e-4, k
    check_gradients(m, g, "B=32 %s" % precision)


def test_b32_bf16_literal_config_at_its_own_size(cond_sd):
    """BASELINE configs[2] "as written" (bf16 operands) exercised at ITS size, B=32 x 3x384x1280 -- a
    property test, not a parity claim (the reference forces fp32, train.py:12-15; mode `bf16` keeps fp32 activations /
    master weights / statistics and rounds only the MFMA operands to 8 mantissa bits):
      * every loss and every gradient element finite;
      * losses within a STATED bound of the f16x2 (fp32-grade) step on the same batch: 5 % for the nine regression /
        heat-map terms, a factor 1.5 for loss_depth (exp(-s)-weighted, the worst-conditioned output) -- measured on the
        conditioned fixtures at small size: 0.3-1.2 % (scratch/bf16_probe.py);
      * the flat gradient keeps cosine > 0.85 with the f16x2 gradient (measured 0.92-0.97 vs fp64 at small size);
      * two runs of the same step are bit-identical (fixed accumulation orders: determinism does not depend on the mode)."""
    g = load_golden("train_full.npz")
    b2 = synth.make_conditioned_batch(int(g["seed"]), 2, 384, 1280)
    b32 = {"img": b2["img"].repeat(16, 1, 1, 1).cuda(),
           "label": {k: v.repeat(16, *([1] * (v.dim() - 1))).cuda() for k, v in b2["label"].items()},
           "img_metas": {"pad_shape": [(384, 1280)] * 32}}

    def run(precision):
        m = build(cond_sd, precision)
        _, loss = m(b32)
        sum(loss.values()).backward()
        torch.cuda.synchronize()
        flat = torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None]).clone()
        return {k: float(v.detach()) for k, v in loss.items()}, flat

    l_ref, g_ref = run("f16x2")
    l_a, g_a = run("bf16")
    l_b, g_b = run("bf16")
    assert all(np.isfinite(v) for v in l_a.values()) and bool(torch.isfinite(g_a).all())
    assert l_a == l_b and torch.equal(g_a, g_b)                                # deterministic
    for k, v in l_ref.items():
        if k == "loss_depth":
            assert v / 1.5 < l_a[k] < v * 1.5, (k, v, l_a[k])
        else:
            assert abs(l_a[k] - v) <= 0.05 * abs(v) + 1e-4, (k, v, l_a[k])
    cos = float(torch.dot(g_a.double(), g_ref.double()) / (g_a.double().norm() * g_ref.double().norm()))
    print("bf16-literal B=32: losses", {k: (round(l_ref[k], 4), round(l_a[k], 4)) for k in l_ref}, "gradient cosine %.4f" % cos)
    assert cos > 0.85, cos
