"""Whole train step on the GPU through the drop-in API: train-mode forward (batch-stat BN, AttnBN),
targets, losses, backward -- against the golden vectors recorded from the real reference
(tests/golden/train_step.npz: losses, per-parameter gradient norms + samples, updated BN buffers).
GPU-only."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err, GOLDEN_SEED
from hipmonocon import netspec, synth

pytestmark = pytest.mark.gpu


def to_cuda(batch):
    d = dict(batch)
    d["img"] = batch["img"].cuda()
    d["label"] = {k: v.cuda() for k, v in batch["label"].items()}
    return d


@pytest.fixture(scope="module")
def stepped(golden_sd):
    from model import MonoConDetector
    m = MonoConDetector(34, pretrained_backbone=False)
    m.load_state_dict(golden_sd, strict=True)
    m = m.cuda().train()
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
    for k, v in loss.items():
        assert v.dim() == 0 and v.requires_grad
        assert abs(float(v) - float(g[k])) <= 2e-4 * abs(float(g[k])) + 1e-6, (k, float(v), float(g[k]))
    assert abs(float(total) - float(g["total"])) <= 2e-4 * abs(float(g["total"]))
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
    # 1.2-2.5x the CPU's fp32 noise at equal algorithm.  Bounds: every tensor within 6x the worst
    # reference-fp32 deviation of its section; section medians within 3x the reference medians.
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
            assert r[2] <= 6.0 * ref_max + 1e-3, r
            assert r[4] <= 6.0 * ref_max + 2e-3, r
        assert hip_med <= 3.0 * ref_med + 1e-4, (sec, hip_med, ref_med)


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


@pytest.mark.parametrize("shape", [(3, 64, 128), (2, 128, 512), (5, 96, 160), (2, 96, 1248)], ids=lambda s: "B%d_%dx%d" % s)
def test_train_forward_shape_sweep_vs_oracle(golden_sd, shape):
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
    m = MonoConDetector(34, pretrained_backbone=False)
    m.load_state_dict(golden_sd, strict=True)
    m = m.cuda().train()
    _, loss = m(to_cuda(batch))
    sum(loss.values()).backward()
    for k, v in loss.items():
        assert abs(float(v) - float(L[k])) <= 2e-3 * abs(float(L[k])) + 1e-4, (k, float(v), float(L[k]))
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
    m = MonoConDetector(34, pretrained_backbone=False)
    m.load_state_dict(golden_sd, strict=True)
    m = m.cuda().train()
    _, loss = m(to_cuda(batch))
    sum(loss.values()).backward()
    for k, v in loss.items():
        assert abs(float(v) - float(L[k])) <= 2e-3 * abs(float(L[k])) + 1e-4, (k, float(v), float(L[k]))
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
