"""pytest wiring: path set-up, the ``gpu`` marker and shared fixtures."""
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "monocon-pytorch_amd")
for p in (PKG, REPO):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(REPO, "tests", "golden")
GOLDEN_SEED = 7


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # torch 2.10's own DataLoader pin thread passes the deprecated argument, once per tensor of every batch
    config.addinivalue_line("filterwarnings", r"ignore:The argument 'device' of Tensor\.(pin_memory|is_pinned)\(\) is deprecated:DeprecationWarning")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden_sd():
    """Synthetic parameters (seed 7) with the calibrated BN statistics fixture."""
    from hipmonocon import synth
    stats = load_golden("bn_calib_seed%d.npz" % GOLDEN_SEED)
    return synth.make_state_dict(GOLDEN_SEED, bn_stats={k: stats[k] for k in stats.files})


def rel_err(a, b):
    """norm-wise error used throughout: max|a-b| / max|b|."""
    import torch
    a = torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a.detach().cpu()).double()
    b = torch.as_tensor(np.asarray(b) if not torch.is_tensor(b) else b.detach().cpu()).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def sample_step(numel, target=2048):
    """stride of the per-tensor gradient samples in train_cond_*.npz / dp_shards.npz (make_golden.py gsample)."""
    return max(1, numel // target)


def gsample(t):
    return t.detach().reshape(-1)[::sample_step(t.numel())]


def grad_rel_l2(got, ref_sample, ref_norm, numel):
    """relative L2 error of a gradient tensor on the golden's strided sample.  The scale never drops below what the
    tensor's full fp64 norm implies for a sample of this size (a bias whose true gradient is ~0 carries only
    cancellation noise)."""
    import torch
    a = gsample(got).detach().cpu().double().reshape(-1)
    b = torch.as_tensor(np.asarray(ref_sample)).double().reshape(-1)
    scale = max(float(b.norm()), float(ref_norm) / numel ** 0.5 * len(b) ** 0.5, 1e-30)
    return float((a - b).norm() / scale)


@pytest.fixture(scope="session")
def cond_sd():
    """Conditioned parameters (small head output weights) for the flip-free gradient fixtures."""
    from hipmonocon import synth
    stats = load_golden("bn_calib_seed%d.npz" % GOLDEN_SEED)
    return synth.make_conditioned_state_dict(GOLDEN_SEED, bn_stats={k: stats[k] for k in stats.files})
