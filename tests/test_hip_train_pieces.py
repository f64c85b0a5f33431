"""Training-side pieces through the C-ABI: target generation (integer-exact indices / masks),
the ten losses and their gradients with respect to the prediction maps.  GPU-only."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err, GOLDEN_SEED
from hipmonocon import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from hipmonocon.engine import Engine
    return Engine()


def test_targets_vs_reference_golden(eng):
    g = load_golden("targets.npz")
    label = {k[3:]: torch.from_numpy(g[k]).cuda() for k in g.files if k.startswith("in.")}
    T = eng.make_targets(label, (384, 1280), (96, 320))
    for name in ("indices", "indices_kpt", "mask_target"):
        assert np.array_equal(T[name].cpu().numpy(), g[name]), name          # integer outputs: exact
    for name in ("mask_center2kpt_offset", "mask_kpt_heatmap_offset", "alpha_cls_target", "wh_target", "offset_target",
                 "dim_target", "depth_target", "center2kpt_offset_target", "kpt_heatmap_offset_target",
                 "alpha_offset_target"):
        assert np.array_equal(T[name].cpu().numpy(), g[name]), name          # same fp32 op sequence: exact
    for name in ("center_heatmap_target", "kpt_heatmap_target"):
        got, ref = T[name].cpu().numpy(), g[name]
        assert np.array_equal(got == 1.0, ref == 1.0), name                   # which pixels are positives
        assert np.array_equal(got > 0, ref > 0), name                         # support of every splat
        assert np.abs(got - ref).max() < 2e-7, name                           # expf vs torch.exp: <= 1 ulp


def test_targets_other_resolution_vs_oracle(eng):
    from oracle import monocon_oracle as O
    lab = synth.make_labels(55, 3, 192, 384)
    T = eng.make_targets({k: torch.from_numpy(v).cuda() for k, v in lab.items()}, (192, 384), (48, 96))
    ref = O.make_targets({k: torch.from_numpy(v) for k, v in lab.items()}, (192, 384), (3, 64, 48, 96))
    for k, v in ref.items():
        got = T[k].cpu()
        if v.dtype in (torch.long, torch.bool):
            assert torch.equal(got, v), k
        else:
            assert (got - v).abs().max() < 2e-7, k


@pytest.mark.parametrize("max_objs,max_gen", [(8, 8), (50, 50)], ids=["all_8_slots_used", "50_slots"])
def test_targets_other_max_objs_vs_oracle(eng, max_objs, max_gen):
    """MODEL.HEAD.MAX_OBJS other than 30 (every slot filled / many overlapping splats on one map): the per-slot
    ranks, the integer max-splat and all masks stay bit-exact."""
    from oracle import monocon_oracle as O
    lab = synth.make_labels(91, 2, 192, 384, max_objs=max_objs, min_objs=max_gen, max_gen=max_gen)
    assert int(lab["mask"].sum()) == 2 * max_gen
    T = eng.make_targets({k: torch.from_numpy(v).cuda() for k, v in lab.items()}, (192, 384), (48, 96), max_objs=max_objs)
    ref = O.make_targets({k: torch.from_numpy(v) for k, v in lab.items()}, (192, 384), (2, 64, 48, 96), max_objs=max_objs)
    for k, v in ref.items():
        got = T[k].cpu()
        if v.dtype in (torch.long, torch.bool):
            assert torch.equal(got, v), k
        else:
            assert (got - v).abs().max() < 2e-7, k


def _train_case(seed=GOLDEN_SEED + 4):
    """predictions from the oracle's train-mode forward (the reference-pinned fixture inputs)."""
    from oracle import monocon_oracle as O
    stats = load_golden("bn_calib_seed%d.npz" % GOLDEN_SEED)
    sd = synth.make_state_dict(GOLDEN_SEED, bn_stats={k: stats[k] for k in stats.files})
    batch = synth.make_batch(seed, 2, 192, 384)
    with torch.no_grad():
        preds, T, L, _ = O.train_forward(sd, batch)
    return batch, preds, T, L


def test_losses_vs_reference_golden(eng):
    g = load_golden("train_step.npz")
    batch, preds, Tref, Lref = _train_case()
    label = {k: v.cuda() for k, v in batch["label"].items()}
    T = eng.make_targets(label, (192, 384), (48, 96))
    L = eng.losses({k: v.cuda().contiguous() for k, v in preds.items()}, T).cpu()
    from hipmonocon.netspec import LOSS_KEYS
    for i, k in enumerate(LOSS_KEYS):
        assert abs(float(L[i]) - float(g[k])) <= 1e-4 * abs(float(g[k])) + 1e-6, (k, float(L[i]), float(g[k]))
        assert abs(float(L[i]) - float(Lref[k])) <= 1e-4 * abs(float(Lref[k])) + 1e-6, k


def test_loss_gradients_vs_autograd(eng):
    """d(sum w_i loss_i)/d(raw head outputs) vs torch autograd through the oracle's loss code."""
    from oracle import monocon_oracle as O
    batch, preds, Tref, _ = _train_case(GOLDEN_SEED + 9)
    # raw (pre-activation) maps as autograd leaves
    raw = {}
    for k, v in preds.items():
        if k in ("center_heatmap_pred", "kpt_heatmap_pred"):
            x = torch.logit(v)
            # entries sitting on the clamp were produced by logits strictly beyond it: rebuild them
            # that way (a logit that lands exactly ON the boundary is a measure-zero case where
            # torch.clamp's inclusive mask and a mask derived from the clamped value differ)
            x = torch.where(v <= 1e-4, x - 1.0, torch.where(v >= 1 - 1e-4, x + 1.0, x))
            raw[k] = x.clone().requires_grad_(True)
        elif k == "depth_pred":
            x0 = torch.logit(1.0 / (v[:, 0:1] + 1.0))
            raw[k] = torch.cat([x0, v[:, 1:2]], 1).clone().requires_grad_(True)
        else:
            raw[k] = v.clone().requires_grad_(True)
    act = {}
    for k, v in raw.items():
        if k in ("center_heatmap_pred", "kpt_heatmap_pred"):
            act[k] = torch.clamp(torch.sigmoid(v), 1e-4, 1 - 1e-4)
        elif k == "depth_pred":
            act[k] = torch.cat([1.0 / (torch.sigmoid(v[:, 0:1]) + 1e-12) - 1.0, v[:, 1:2]], 1)
        else:
            act[k] = v
    L = O.losses(act, Tref)
    w = torch.tensor([1.0, 0.5, 2.0, 1.5, 1.0, 0.7, 1.0, 3.0, 1.0, 0.25])
    total = sum(w[i] * v for i, v in enumerate(L.values()))
    total.backward()
    label = {k: v.cuda() for k, v in batch["label"].items()}
    T = eng.make_targets(label, (192, 384), (48, 96))
    d = eng.losses_backward({k: v.detach().cuda().contiguous() for k, v in act.items()}, T, w.cuda())
    for k in raw:
        ref = raw[k].grad
        assert rel_err(d[k].cpu(), ref) < 2e-4, (k, rel_err(d[k].cpu(), ref))


def test_fused_clip_adamw_vs_reference_golden(golden_sd):
    """three optimizer steps on the reference's recorded (pre-clip) gradients, schedule included."""
    from solver import AdamW, CyclicScheduler
    g = load_golden("adamw.npz")
    names = [k[len("step0."):] for k in g.files if k.startswith("step0.")]
    params = [torch.nn.Parameter(golden_sd[n].clone().cuda()) for n in names]
    opt = AdamW(params, lr=2.25e-4, weight_decay=1e-5, betas=(0.95, 0.99), max_grad_norm=None)
    sch = CyclicScheduler(opt, total_steps=1000, target_lr_ratio=(10, 1e-4),
                          target_momentum_ratio=(0.85 / 0.95, 1.0), period_up=0.4)
    for step in range(3):
        lr, b1, norm = g["sched"][step]
        assert opt.param_groups[0]["lr"] == pytest.approx(lr, rel=1e-12)
        assert opt.param_groups[0]["betas"][0] == pytest.approx(b1, rel=1e-12)
        coef = min(1.0, 35.0 / (norm + 1e-6))          # the golden norm is over ALL model gradients
        for p, n in zip(params, names):
            p.grad = (torch.from_numpy(g["grad%d.%s" % (step, n)]) * np.float32(coef)).cuda()
        v0 = params[0]._version
        opt.step()
        sch.step()
        assert params[0]._version > v0
        for p, n in zip(params, names):
            assert rel_err(p.detach().cpu(), g["step%d.%s" % (step, n)]) < 2e-6, (step, n)


def test_fused_clip_matches_torch_clip(golden_sd):
    """global-norm clipping inside the fused step == clip_grad_norm_ followed by an unclipped step."""
    from solver import AdamW
    torch.manual_seed(0)
    shapes = [(64, 32, 3, 3), (128,), (10, 64), (70000,)]
    base = [torch.randn(s) for s in shapes]
    grads = [torch.randn(s) * 3 for s in shapes]
    pa = [torch.nn.Parameter(b.clone().cuda()) for b in base]
    pb = [torch.nn.Parameter(b.clone()) for b in base]
    oa = AdamW(pa, lr=1e-3, weight_decay=1e-2, betas=(0.9, 0.99), max_grad_norm=5.0)
    ob = torch.optim.AdamW(pb, lr=1e-3, weight_decay=1e-2, betas=(0.9, 0.99))
    for it in range(3):
        for p, q, gr in zip(pa, pb, grads):
            p.grad = (gr * (it + 1)).cuda()
            q.grad = (gr * (it + 1)).clone()
        ref_norm = torch.nn.utils.clip_grad_norm_(pb, 5.0)
        oa.step(); ob.step()
        assert float(oa.last_grad_norm) == pytest.approx(float(ref_norm), rel=1e-5)
        for p, q in zip(pa, pb):
            assert rel_err(p.detach().cpu(), q.detach()) < 2e-6
