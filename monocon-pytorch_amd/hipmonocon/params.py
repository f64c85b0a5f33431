"""Parameter-holder modules.

The product modules keep the reference's ``nn.Module`` tree (attribute names, ``state_dict``
keys, OIHW layouts) but own no arithmetic: these holders only carry ``nn.Parameter`` /
buffers with the shapes ``nn.Conv2d`` / ``nn.BatchNorm2d`` / ``nn.ConvTranspose2d`` would have,
so checkpoints interchange with the reference, optimizers see real parameters, and every
device computation goes to libmonocon_hip.so.  Calling a holder is an error by design
(there is no CPU or eager fallback).
"""
import math

import torch
import torch.nn as nn


class _Holder(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError("%s holds parameters only; computation runs in libmonocon_hip.so through the "
                           "owning DLA / DLAUp / MonoConDenseHeads / MonoConDetector module" % type(self).__name__)


class ConvParams(_Holder):
    """shape-compatible with nn.Conv2d(in_ch, out_ch, k, bias=bias)."""

    def __init__(self, in_ch, out_ch, kernel_size, bias=False):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = in_ch, out_ch, (kernel_size, kernel_size)
        self.weight = nn.Parameter(torch.empty(out_ch, in_ch, kernel_size, kernel_size))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_ch))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        # torch.nn.Conv2d defaults (kaiming_uniform(a=sqrt(5)) + uniform bias)
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            nn.init.uniform_(self.bias, -bound, bound)

    def extra_repr(self):
        return "%d, %d, kernel_size=%s, bias=%s" % (self.in_channels, self.out_channels, self.kernel_size,
                                                    self.bias is not None)


class BNParams(_Holder):
    """shape-compatible with nn.BatchNorm2d(c, eps, momentum, affine)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True):
        super().__init__()
        self.num_features, self.eps, self.momentum, self.affine = num_features, eps, momentum, affine
        if affine:
            self.weight = nn.Parameter(torch.ones(num_features))
            self.bias = nn.Parameter(torch.zeros(num_features))
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def extra_repr(self):
        return "%d, eps=%g, momentum=%g, affine=%s" % (self.num_features, self.eps, self.momentum, self.affine)


class DeconvParams(_Holder):
    """shape-compatible with nn.ConvTranspose2d(c, c, 4, stride=2, padding=1, groups=c, bias=False)."""

    def __init__(self, channels, kernel_size=4):
        super().__init__()
        self.channels = channels
        self.weight = nn.Parameter(torch.empty(channels, 1, kernel_size, kernel_size))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))


class Holders(nn.Module):
    """nn.Sequential-style numbered container (children named '0', '1', ...); not callable.
    ``None`` entries keep their index free (e.g. the parameter-less ReLU at index 2)."""

    def __init__(self, *mods):
        super().__init__()
        for i, m in enumerate(mods):
            if m is not None:
                self.add_module(str(i), m)

    def __getitem__(self, i):
        n = len(self._modules)
        keys = sorted(int(k) for k in self._modules)
        if i < 0:
            i = keys[i]
        return self._modules[str(i)]

    def forward(self, *a, **k):
        raise RuntimeError("Holders is a parameter container; computation runs in libmonocon_hip.so")


# ---------------------------------------------------------------------------------- runtime
def module_state(module, prefix):
    """state_dict of ``module`` with ``prefix`` prepended (tensors are the live storage)."""
    return {prefix + k: v for k, v in module.state_dict(keep_vars=True).items()}


class HipRuntime:
    """Lazily created per-root-module Engine + binding bookkeeping."""

    def __init__(self):
        self.engine = None
        self.precision = 0     # 0 fp32 MFMA, 1 bf16 operands, 2 fp32 as a 3-way bf16 split, 3 fp32 as a 2-way fp16 split

    PRECISION_MODES = {"fp32": 0, "f32": 0, "bf16": 1, "bf16x3": 2, "fp32_split": 2, "f16x2": 3, "fp16x2": 3}

    def set_precision(self, mode):
        if isinstance(mode, str) and mode not in self.PRECISION_MODES:
            raise ValueError("unknown precision mode %r (one of %s)" % (mode, sorted(self.PRECISION_MODES)))
        self.precision = self.PRECISION_MODES.get(mode, mode)
        if self.engine is not None:
            self.engine.set_precision(self.precision)

    def get(self, state):
        from .engine import Engine
        from . import lib
        dev = next(iter(state.values())).device
        if dev.type != "cuda":
            raise lib.MonoconHipError(
                "parameters live on %s: move the model to a HIP device (.to('cuda')); libmonocon_hip has no "
                "CPU path" % dev)
        if self.engine is None or self.engine.device != dev:
            self.engine = Engine(dev.index if dev.index is not None else torch.cuda.current_device())
            if self.precision:
                self.engine.set_precision(self.precision)
        self.engine.bind_state(state)   # Parameters are passed as-is: their _version tracks in-place updates
        return self.engine
