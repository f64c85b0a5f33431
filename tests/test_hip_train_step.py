"""Whole train step on the GPU through the drop-in API: train-mode forward (batch-stat BN, AttnBN),
targets, losses, backward -- against the golden vectors recorded from the real reference
(tests/golden/train_step.npz: losses, per-parameter gradient norms + samples, updated BN buffers).
GPU-only."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err, grad_rel_l2, GOLDEN_SEED
from hipmonocon import netspec, synth

pytestmark = pytest.mark.gpu

# The parity tests below run in all three fp32 modes of the library: "fp32" = v_mfma_f32_32x32x2_f32 (native fp32 matrix
# pipe), "bf16x3" = fp32 emulated on the bf16 matrix pipe (each fp32 operand split into three bf16 pieces, six partial
# products, fp32 accumulation), "f16x2" = emulated on the fp16 matrix pipe (two fp16 pieces of the power-of-two-scaled
# operand, three partial products).  All must meet the SAME fp32 tolerances.
PRECISIONS = ("fp32", "bf16x3", "f16x2")
LOSS_TOL = 1e-4          # BASELINE.json north_star: fp32 losses within 1e-4 relative (judged against the fp64 reference)


def build(sd, precision="fp32"):
    from model import MonoConDetector
    m = MonoConDetector(34, pretrained_backbone=False)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    m.set_precision(precision)
    return m


def oracle_losses_fp64(sd, batch):
    """the CPU oracle's train forward in float64: the yard-stick for loss parity at arbitrary shapes"""
    from oracle import monocon_oracle as O
    sd64 = {k: (v.double().clone() if v.dtype == torch.float32 else v.clone()) for k, v in sd.items()}
    b64 = dict(batch)
    b64["img"] = batch["img"].double()
    with torch.no_grad():
        _, _, L, _ = O.train_forward(sd64, b64)
    return {k: float(v) for k, v in L.items()}


def to_cuda(batch):
    d = dict(batch)
    d["img"] = batch["img"].cuda()
    d["label"] = {k: v.cuda() for k, v in batch["label"].items()}
    return d


@pytest.fixture(scope="module", params=PRECISIONS)
def stepped(golden_sd, request):
    m = build(golden_sd, request.param)
    batch = to_cuda(synth.make_batch(GOLDEN_SEED + 4, 2, 192, 384))
    pred, loss = m(batch)
    total = sum(v for v in loss.values())
    total.backward()
    torch.cuda.synchronize()
    return m, pred, loss, total


def test_losses_match_reference(stepped):
    m, pred, loss, total = stepped
    g = load_golden("train_step.npz")
    assert list(loss.keys()) == list(netspec.LOSS_KEYS)
    tot64 = 0.0
    for k, v in loss.items():
        assert v.dim() == 0 and v.requires_grad
        ref64 = float(g["f64." + k])                 # the reference run in float64 (measured: <= 4.4e-5)
        tot64 += ref64
        assert abs(float(v.detach()) - ref64) <= LOSS_TOL * abs(ref64) + 1e-7, (k, float(v.detach()), ref64)
    assert abs(float(total.detach()) - tot64) <= LOSS_TOL * abs(tot64)
    for k, v in pred.items():
        assert rel_err(v.detach().cpu().reshape(-1)[::31], g["pred." + k + ".sample"]) < 2e-4, k


def test_running_statistics_match_reference(stepped):
    m = stepped[0]
    g = load_golden("train_step.npz")
    sd = m.state_dict()
    for k in sd:
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert rel_err(sd[k].cpu(), g["buf." + k]) < 2e-4, k
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == 1, k


def test_gradients_match_reference(stepped):
    m = stepped[0]
    g = load_golden("train_step.npz")
    dead = set(g["dead"].tolist())

    def err(a, b, norm, numel):
        # relative L2 over the strided sample; the scale never drops below the tensor's RMS entry (a
        # bias whose true gradient is 0 shows 1e-8 cancellation noise in the reference).  L2 rather
        # than max: one ReLU / max-pool decision flipping under fp32 round-off moves a single row of a
        # weight gradient by several percent (seen for the reference's own fp32 run too).
        a, b = torch.as_tensor(np.asarray(a)).double().reshape(-1), torch.as_tensor(np.asarray(b)).double().reshape(-1)
        scale = max(float(b.norm()), norm / numel ** 0.5 * len(b) ** 0.5, 1e-30)
        return float((a - b).norm() / scale)

    # Yard-stick: the reference run in fp64.  On this B=2 train-mode-BN fixture fp32 round-off is
    # amplified on the way down: the reference's OWN fp32 CPU gradients sit 1e-5 (heads) .. 2.5e-2
    # (backbone) from its fp64 gradients, and which tensor lands where is arbitrary.  The HIP kernels
    # accumulate each output in one long fp32 FMA chain (MKLDNN blocks its sums), which measures as
    # 1.0-2.6x the CPU's fp32 noise at equal algorithm (measured, both precision modes).  Bounds: every
    # tensor within 5x the worst reference-fp32 deviation of its section; section medians within 2.5x the
    # reference medians.  (The cause is ReLU / max-pool decisions flipping under round-off -- see
    # test_conditioned_gradients_vs_reference_fp64 for the flip-free fixtures that are held to 1e-3.)
    rows = []
    for n, p in m.named_parameters():
        if n in dead:
            assert p.grad is None, n
            continue
        assert p.grad is not None, n
        norm64, ne = float(g["gnorm64." + n]), p.numel()
        e_hip = err(p.grad.cpu().reshape(-1)[::101], g["gsample64." + n], norm64, ne)
        e_ref = err(g["gsample." + n], g["gsample64." + n], norm64, ne)
        got_norm = float(p.grad.double().norm())
        rows.append((n.split(".")[0], n, e_hip, e_ref, abs(got_norm - norm64) / max(norm64, 1e-30)))
    assert len(rows) == 236
    for sec in ("head", "neck", "backbone"):
        sel = [r for r in rows if r[0] == sec]
        ref_max = max(r[3] for r in sel)
        ref_med, hip_med = float(np.median([r[3] for r in sel])), float(np.median([r[2] for r in sel]))
        print("%-9s tensors %3d  ref32-vs-fp64 max %.2e med %.2e | hip-vs-fp64 max %.2e med %.2e"
              % (sec, len(sel), ref_max, ref_med, max(r[2] for r in sel), hip_med))
        for r in sel:
            assert r[2] <= 5.0 * ref_max + 1e-3, r
            assert r[4] <= 5.0 * ref_max + 2e-3, r
        assert hip_med <= 2.5 * ref_med + 1e-4, (sec, hip_med, ref_med)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_conditioned_gradients_vs_reference_fp64(cond_sd, case, precision):
    """Gradient parity on well-conditioned fixtures (tests/golden/train_cond_*.npz, make_golden.py cond_train):
    small head output weights (no e^{-s}/sigmoid blow-up in the depth loss), per-image contrast (the BatchNorm over
    the batch inside AttnBN is not normalising round-off), and seeds SELECTED so that the reference itself has no
    ReLU / max-pool decision within a few fp32 round-offs of its threshold -- its own fp32 run then sits <= 1.4e-4
    from its fp64 run on every tensor, instead of the 1e-2 a single flipped decision causes.

    Fixtures 0 and 1: every loss within 1e-4 and EVERY one of the 236 gradient tensors within 1e-3 relative L2 of the
    reference's fp64 gradients (measured: <= 3.4e-4 fp32, <= 2.2e-4 bf16x3).  Fixtures 2 and 3 document the flip
    mechanism: the HIP kernels round differently from MKLDNN, one decision lands on the other side (fixture 3: in
    both precision modes), and all upstream tensors shift together (max 1.1e-2) while the losses, the BN buffers
    and the median tensor stay at the 1e-5 level; for those two only the statistical bounds are asserted."""
    g = load_golden("train_cond_%d.npz" % case)
    B, H, W = (int(x) for x in g["shape"])
    m = build(cond_sd, precision)
    _, loss = m(to_cuda(synth.make_conditioned_batch(int(g["seed"]), B, H, W)))
    sum(loss.values()).backward()
    torch.cuda.synchronize()
    for k, v in loss.items():
        ref = float(g["f64." + k])
        assert abs(float(v.detach()) - ref) <= LOSS_TOL * abs(ref) + 1e-7, (k, float(v.detach()), ref)
    errs = {}
    for n, p in m.named_parameters():
        if n in netspec.DEAD_PARAMS:
            assert p.grad is None, n
            continue
        errs[n] = grad_rel_l2(p.grad, g["g64." + n], g["gnorm64." + n], p.numel())
    assert len(errs) == 236
    e = np.array(list(errs.values()))
    worst = max(errs, key=errs.get)
    print("cond fixture %d %s: max %.2e (%s) median %.2e, reference fp32-vs-fp64 max %.2e"
          % (case, precision, e.max(), worst, np.median(e), float(g["ref32_max_err"])))
    if case in (0, 1):
        assert float(np.median(e)) <= 1e-4, np.median(e)
        assert e.max() <= 1e-3, (worst, e.max())
    else:       # a flipped decision shifts every tensor upstream of it together: only gross bounds here
        assert float(np.median(e)) <= 5e-3, np.median(e)
        assert e.max() <= 5e-2, (worst, e.max())
    sd = m.state_dict()
    for k in sd:
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert rel_err(sd[k].cpu(), g["buf64." + k]) < 1e-4, k


def test_train_step_with_fused_optimizer_reduces_loss(golden_sd):
    """three full steps (forward, backward, fused clip+AdamW, cyclic schedule) run end to end."""
    from model import MonoConDetector
    from solver import AdamW, CyclicScheduler
    m = MonoConDetector(34, pretrained_backbone=False)
    m.load_state_dict(golden_sd, strict=True)
    m = m.cuda().train()
    opt = AdamW(m.parameters(), lr=2.25e-4, weight_decay=1e-5, betas=(0.95, 0.99), max_grad_norm=35.0)
    sch = CyclicScheduler(opt, total_steps=100)
    batch = to_cuda(synth.make_batch(77, 4, 192, 384))
    hist = []
    for _ in range(3):
        opt.zero_grad()
        _, loss = m(batch)
        total = sum(v for v in loss.values())
        total.backward()
        opt.step()
        sch.step()
        hist.append(float(total))
    assert all(np.isfinite(hist))
    assert float(opt.last_grad_norm) > 0
    m.eval()
    out = m({"img": batch["img"]})
    assert all(torch.isfinite(v).all() for v in out.values())


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("shape", [(3, 64, 128), (2, 128, 512), (5, 96, 160), (2, 96, 1248)], ids=lambda s: "B%d_%dx%d" % s)
def test_train_forward_shape_sweep_vs_oracle(golden_sd, shape, precision):
    """odd batches / other resolutions through the train plan (autotuned conv shapes, 16-channel row kernels
    where the width allows, parity-class stride-2 data gradients): losses vs the CPU oracle's train forward,
    gradients finite and of the oracle's total norm."""
    from model import MonoConDetector
    from oracle import monocon_oracle as O
    B, H, W = shape
    batch = synth.make_batch(2000 + B + H + W, B, H, W)
    live = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and "running" not in k else v.clone())
            for k, v in golden_sd.items()}
    _, _, L, _ = O.train_forward(live, batch)
    sum(L.values()).backward()
    ref_norm = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in live.values()
                                    if getattr(p, "grad", None) is not None)))
    L64 = oracle_losses_fp64(golden_sd, batch)
    m = build(golden_sd, precision)
    _, loss = m(to_cuda(batch))
    sum(loss.values()).backward()
    for k, v in loss.items():       # measured: <= 5.7e-5 (the oracle's own fp32 run sits up to 6.3e-5 from its fp64 run)
        assert abs(float(v.detach()) - L64[k]) <= LOSS_TOL * abs(L64[k]) + 1e-7, (k, float(v.detach()), L64[k])
    g = torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None])
    assert bool(torch.isfinite(g).all())
    assert abs(float(g.double().norm()) - ref_norm) <= 0.05 * ref_norm, (float(g.double().norm()), ref_norm)


def test_image_without_objects_inside_a_batch(golden_sd):
    """one image of the batch has no valid object (mask row all zero, stale label values left in place): the
    regression losses average over the objects of the other images only, as in the reference (the batch as a whole
    is not empty, so nothing raises)."""
    from model import MonoConDetector
    from oracle import monocon_oracle as O
    batch = synth.make_batch(4242, 3, 64, 128)
    batch["label"]["mask"][1] = 0
    live = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and "running" not in k else v.clone())
            for k, v in golden_sd.items()}
    _, T, L, _ = O.train_forward(live, batch)
    sum(L.values()).backward()
    ref_norm = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in live.values()
                                    if getattr(p, "grad", None) is not None)))
    L64 = oracle_losses_fp64(golden_sd, batch)
    m = build(golden_sd)
    _, loss = m(to_cuda(batch))
    sum(loss.values()).backward()
    for k, v in loss.items():
        assert abs(float(v.detach()) - L64[k]) <= LOSS_TOL * abs(L64[k]) + 1e-7, (k, float(v.detach()), L64[k])
    g = torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None])
    assert bool(torch.isfinite(g).all())
    assert abs(float(g.double().norm()) - ref_norm) <= 0.05 * ref_norm, (float(g.double().norm()), ref_norm)


def test_train_step_is_independent_of_the_conv_tiling(golden_sd):
    """losses, gradients and updated BN buffers are bit-identical whether the conv workgroup shapes are autotuned
    or forced to one tiling: the accumulation order of an output element and the per-patch statistics partials do
    not depend on the shape (so plan-build timing noise can never change a training run)."""
    from model import MonoConDetector
    batch = to_cuda(synth.make_batch(GOLDEN_SEED + 9, 2, 96, 160))
    res = []
    for cfg in (0, 6, 22):          # autotuned; 128 px x 32 ch everywhere; its wave-specialised variant
        m = MonoConDetector(34, pretrained_backbone=False)
        m.load_state_dict(golden_sd, strict=True)
        m = m.cuda().train()
        eng = m._engine()
        eng.set_conv_cfg(cfg)
        _, loss = m(batch)
        sum(loss.values()).backward()
        res.append(([v.detach().clone() for v in loss.values()],
                    torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None]).clone(),
                    torch.cat([b.flatten().float() for b in m.buffers()]).clone()))
        eng.set_conv_cfg(0)
    for other in res[1:]:
        for a, b in zip(res[0][0], other[0]):
            assert torch.equal(a, b)
        assert torch.equal(res[0][1], other[1])
        assert torch.equal(res[0][2], other[2])



def test_lazy_activations_track_the_stored_ones(golden_sd, monkeypatch):
    """Round 6: in mode f16x2 the post-BatchNorm activations without residual are never stored -- every consumer
    (conv / weight-gradient staging, pools, deconvs, the residual input of the next block) forms max(a * y + b, 0) on load
    (model/backbone/dla.py:34-51, dla_neck.py:30-38 under train-mode BatchNorm).  With the same operand scale the staged
    fp16 pieces are bit-identical to those of a stored map; the scale itself comes from a bound of max |z| instead of its
    exact value, so the step agrees to round-off of the 22-bit operand split, not bit for bit.  MONOCON_HIP_LAZY_MIN=0 makes
    every eligible map lazy at this small size.  The bit-packed ReLU mask of the residual layers (MONOCON_HIP_ZBITS) is the
    same mask: bit-identical.  `feat`, the input of the heads, is stored by default (its consumers are the two longest launches of
    the step); MONOCON_HIP_LAZY_FEAT=1 makes it lazy like the other nodes."""
    from model import MonoConDetector
    batch = to_cuda(synth.make_batch(GOLDEN_SEED + 12, 3, 128, 224))
    res = {}
    for tag, env in (("stored", {"MONOCON_HIP_LAZY_Z": "0"}),
                     ("lazy", {"MONOCON_HIP_LAZY_Z": "3", "MONOCON_HIP_LAZY_MIN": "0"}),
                     ("lazy_nobits", {"MONOCON_HIP_LAZY_Z": "3", "MONOCON_HIP_LAZY_MIN": "0", "MONOCON_HIP_ZBITS": "0"}),
                     ("lazy_feat", {"MONOCON_HIP_LAZY_Z": "3", "MONOCON_HIP_LAZY_MIN": "0", "MONOCON_HIP_LAZY_FEAT": "1"}),
                     ("relu_only", {"MONOCON_HIP_LAZY_Z": "1", "MONOCON_HIP_LAZY_MIN": "0"})):
        for k in ("MONOCON_HIP_LAZY_Z", "MONOCON_HIP_LAZY_MIN", "MONOCON_HIP_ZBITS", "MONOCON_HIP_LAZY_FEAT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)              # read when the train plan is built
        m = MonoConDetector(34, pretrained_backbone=False)
        m.load_state_dict(golden_sd, strict=True)
        m = m.cuda().train().set_precision("f16x2")
        _, loss = m(batch)
        sum(loss.values()).backward()
        torch.cuda.synchronize()
        res[tag] = ({k: float(v.detach()) for k, v in loss.items()},
                    {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None},
                    {n: b.detach().clone().double() for n, b in m.named_buffers()}, m._engine().workspace_bytes())
    for k in res["lazy"][1]:                                   # the same ReLU masks, read from bits or from z
        assert torch.equal(res["lazy"][1][k], res["lazy_nobits"][1][k]), k
    assert res["lazy"][0] == res["lazy_nobits"][0]
    assert res["lazy_feat"][3] < res["lazy"][3] < res["relu_only"][3] < res["stored"][3]      # maps that are never allocated (at B = 32, 384x1280: 30.3 -> 24.4 GB)
    for tag in ("lazy", "lazy_feat", "relu_only"):
        for k, v in res["stored"][0].items():
            assert abs(res[tag][0][k] - v) <= 2e-5 * abs(v) + 1e-7, (tag, k, res[tag][0][k], v)
        worst = 0.0
        for n, g in res["stored"][1].items():
            a, b = res[tag][1][n].double(), g.double()
            e = float((a - b).norm() / b.norm().clamp_min(1e-3 * (b.numel() ** 0.5) * float(b.abs().max().clamp_min(1e-30))))
            worst = max(worst, e)
            assert e <= 2e-3, (tag, n, e)
        for n, b in res["stored"][2].items():
            assert float((res[tag][2][n] - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-7, (tag, n)


@pytest.mark.parametrize("precision", ("fp32", "f16x2"))
def test_head_backward_without_the_stored_gradient_is_bit_identical(golden_sd, precision, monkeypatch):
    """the AttnBN backward of the heads forms the ReLU-masked gradient of the nine 1x1 convs again from the raw prediction
    gradients (launch_head_dx, the default) instead of reading what head_bwd_kernel stored (MONOCON_HIP_HEAD_DX_FUSE=0):
    the same fma chain per element, so every gradient is bit-identical -- except the biases of the heads' 3x3 convs, whose
    column sums are folded over a different partition (monocon_heads.py:114-131, attentive_norm.py:79-91 under autograd)"""
    from model import MonoConDetector
    batch = to_cuda(synth.make_batch(GOLDEN_SEED + 11, 3, 96, 224))
    res = []
    for fuse in ("0", "1"):
        monkeypatch.setenv("MONOCON_HIP_HEAD_DX_FUSE", fuse)      # read when the train plan is built
        m = MonoConDetector(34, pretrained_backbone=False)
        m.load_state_dict(golden_sd, strict=True)
        m = m.cuda().train().set_precision(precision)
        _, loss = m(batch)
        sum(loss.values()).backward()
        torch.cuda.synchronize()
        res.append({n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
    stored, fused = res
    assert stored.keys() == fused.keys()
    n_bias = 0
    for n in stored:
        if n.startswith("head.") and n.endswith(".0.bias"):
            n_bias += 1
            assert rel_err(fused[n].cpu(), stored[n].cpu()) < 1e-4, n      # (sums of 10^5 terms that cancel to ~1e-5 of their magnitude)
        else:
            assert torch.equal(stored[n], fused[n]), n
    assert n_bias >= 9            # the nine 3x3 convs (dir_cls / dir_reg: their 1x1 convs are called ".0" too and are bit-identical)


def test_fused_stride2_thin_data_gradient_matches_the_four_class_launches(golden_sd, monkeypatch):
    """level1's data gradient (32 -> 16 channels, stride 2, full resolution) as ONE pass of dgrad_s2_thin_kernel (the default)
    against the four output-parity launches of the tiled kernel (MONOCON_HIP_DGRAD_S2_THIN=0): the same f16x2 products
    accumulated in a different association -- only the gradients below level1 may differ, and only at rounding level
    (model/backbone/dla.py:280-298 under autograd)"""
    from model import MonoConDetector
    batch = to_cuda(synth.make_batch(GOLDEN_SEED + 12, 3, 96, 224))        # 224: a half-filled last strip
    res = []
    for flag in ("0", "1"):
        monkeypatch.setenv("MONOCON_HIP_DGRAD_S2_THIN", flag)             # read when the train plan is built
        m = MonoConDetector(34, pretrained_backbone=False)
        m.load_state_dict(golden_sd, strict=True)
        m = m.cuda().train().set_precision("f16x2")
        _, loss = m(batch)
        sum(loss.values()).backward()
        torch.cuda.synchronize()
        res.append({n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
    four, one = res
    below = differing = 0
    for n in four:
        if n.startswith("backbone.base_layer") or n.startswith("backbone.level0"):
            below += 1
            differing += int(not torch.equal(four[n], one[n]))
            assert rel_err(one[n].cpu(), four[n].cpu()) < 1e-6, n
        else:
            assert torch.equal(four[n], one[n]), n
    # (it IS the other kernel: the six tensors below level1 differ at rounding level -- a 16-element BatchNorm gradient can
    #  round to the same floats by chance, the 3x3 / 7x7 weight gradients cannot all do so)
    assert below == 6 and differing >= 2


def test_full_size_train_step_is_deterministic(golden_sd):
    """size-independent property at the full 384x1280 resolution: the same state and batch give bit-identical losses,
    gradients and BN buffers twice in a row (fixed accumulation orders everywhere, two streams included)."""
    from model import MonoConDetector
    batch = to_cuda(synth.make_batch(GOLDEN_SEED + 21, 2, 384, 1280))
    outs = []
    for _ in range(2):
        m = MonoConDetector(34, pretrained_backbone=False)
        m.load_state_dict(golden_sd, strict=True)
        m = m.cuda().train()
        _, loss = m(batch)
        sum(loss.values()).backward()
        torch.cuda.synchronize()
        outs.append((torch.stack([v.detach() for v in loss.values()]).clone(),
                     torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None]).clone(),
                     torch.cat([b.flatten().float() for b in m.buffers()]).clone()))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    assert bool(torch.isfinite(outs[0][1]).all())


def test_a_few_optimizer_steps_reduce_the_loss(golden_sd):
    """end-to-end sanity of forward + backward + fused clip/AdamW + cyclic schedule: on one fixed batch the total loss
    falls over ten steps (sign and scale of every gradient path)."""
    from model import MonoConDetector
    from solver import AdamW, CyclicScheduler
    batch = to_cuda(synth.make_batch(GOLDEN_SEED + 22, 4, 96, 320))
    m = MonoConDetector(34, pretrained_backbone=False)
    m.load_state_dict(golden_sd, strict=True)
    m = m.cuda().train()
    opt = AdamW(m.parameters(), lr=2.25e-4, weight_decay=1e-5, betas=(0.95, 0.99), max_grad_norm=35.0)
    sch = CyclicScheduler(opt, total_steps=100)
    totals = []
    for _ in range(10):
        opt.zero_grad()
        _, loss = m(batch)
        t = sum(loss.values())
        t.backward()
        opt.step()
        sch.step()
        totals.append(float(t))
    assert all(np.isfinite(totals))
    assert totals[-1] < 0.9 * totals[0], totals


def test_bf16x3_and_fp32_training_trajectories_agree(cond_sd):
    """the two fp32 modes as TRAINING paths: five optimizer steps (forward, backward, fused clip + AdamW, schedule) from
    the same state on the same batches.  The first step's ten losses agree to 1e-5 (round-off only; measured 1.7e-6) and the parameters
    after it to 1e-3 of their norm (see the comment at the assertion).  From then on the runs separate the way any two fp32 implementations do -- AdamW's
    first updates are sign-like (m / sqrt(v) ~ +-1), so a gradient that is ~0 in one mode and ~-0 in the other moves
    that weight by 2*lr -- measured: losses within 2e-2 after five steps; bound asserted: 1e-1, both runs finite."""
    from solver import AdamW, CyclicScheduler
    batches = [to_cuda(synth.make_conditioned_batch(900 + i, 4, 64, 128)) for i in range(5)]
    runs = {}
    for prec in ("fp32", "bf16x3"):
        m = build(cond_sd, prec)
        opt = AdamW(m.parameters(), lr=2.25e-4, weight_decay=1e-5, betas=(0.95, 0.99), max_grad_norm=35.0)
        sch = CyclicScheduler(opt, total_steps=100)
        hist, p1 = [], None
        for i, b in enumerate(batches):
            opt.zero_grad()
            _, loss = m(b)
            sum(loss.values()).backward()
            opt.step()
            sch.step()
            hist.append([float(v.detach()) for v in loss.values()])
            if i == 0:
                p1 = torch.cat([p.detach().flatten() for p in m.parameters()]).double().cpu()
        runs[prec] = (np.array(hist), p1)
    ha, pa = runs["fp32"]
    hb, pb = runs["bf16x3"]
    assert np.all(np.isfinite(ha)) and np.all(np.isfinite(hb))
    rel = np.abs(ha - hb) / np.maximum(np.abs(ha), 1e-12)
    assert rel[0].max() < 1e-5, rel[0].max()
    # AdamW's first step moves every weight by ~lr * sign(g): the ~0.03 % of weights whose gradient is zero up to round-off
    # take opposite signs in the two modes (measured 2e-4 of the parameter norm = 4 % of the update norm)
    assert float((pa - pb).norm() / pa.norm()) < 1e-3
    assert rel.max() < 1e-1, rel.max()
