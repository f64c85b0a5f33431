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
