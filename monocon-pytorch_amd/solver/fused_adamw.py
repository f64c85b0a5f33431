"""Fused gradient-clip + AdamW on MI355X.

Drop-in for ``torch.optim.AdamW`` in the reference train loop (engine/monocon_engine.py:35-55,
94-102).  The class is deliberately named ``AdamW`` (CyclicScheduler asserts on the class name,
solver/cyclic_scheduler.py:16-17).  ``step()`` launches libmonocon_hip's multi-tensor kernels:
global L2 gradient norm, clip coefficient and the decoupled-weight-decay Adam update in two passes
over the parameters instead of hundreds of per-tensor launches.  Parameters without a gradient
(the six tensors the reference never back-propagates into, SURVEY §8a quirk i) are skipped, as
torch does.
"""
import ctypes as C

import torch
from torch.optim.optimizer import Optimizer

from hipmonocon import lib as _lib


class AdamW(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=None,
                 engine=None):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise NotImplementedError("the fused optimizer handles one parameter group")
        self.max_grad_norm = max_grad_norm
        self._engine = engine
        self._bound_sig = None
        self._step_cache = None         # host-side copy of the common step count (None: re-read from the state)
        self.last_grad_norm = None      # device scalar of the most recent pre-clip norm

    def set_engine(self, engine):
        self._engine = engine
        self._bound_sig = None

    def load_state_dict(self, state_dict):
        """Restores the moments and the per-parameter ``step`` (torch.optim.AdamW's checkpoint layout, which
        is what the reference saves, engine/base_engine.py:155-219).  The kernels hold raw pointers to the moment
        tensors, which torch replaces here: the binding is dropped and rebuilt on the next step()."""
        super().load_state_dict(state_dict)
        self._bound_sig = None
        self._step_cache = None
        # torch hands the caller's own 'step' (and same-device moment) tensors through uncopied; step() updates
        # them in place, so take private copies -- loading one checkpoint dict into two optimizers stays safe
        for st in self.state.values():
            for key in ('step', 'exp_avg', 'exp_avg_sq'):
                if key in st and torch.is_tensor(st[key]):
                    st[key] = st[key].clone()

    def __setstate__(self, state):
        super().__setstate__(state)
        self._bound_sig = None
        self._step_cache = None

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self._bound_sig = None
        self._step_cache = None

    def zero_state(self):
        """forget the moments and step counts (a fresh optimizer over the same parameters)"""
        self.state.clear()
        self._bound_sig = None
        self._step_cache = None

    def _common_step(self, plist):
        """The bias-correction step lives in ``state[p]['step']`` only (so it survives state_dict round trips);
        the fused kernel applies one step number to all tensors, as they always share it in this train loop.  Reading
        it back is one ``int()`` per tensor -- a host sync each if a resumed checkpoint left the step tensors on the
        device -- so it is read once (after construction / load_state_dict / a change of the parameter list) and
        tracked on the host from then on."""
        # keyed on the IDENTITY of the stepped tensors and of their step objects (the length alone let a changed
        # parameter set of equal size, a reset state or a user-edited state[p]['step'] reuse a stale host copy)
        key = tuple((id(p), id(self.state[p].get('step'))) for p in plist)
        if self._step_cache is not None and self._step_cache[0] == key:
            return self._step_cache[1]
        steps = {int(self.state[p]['step']) for p in plist}
        if len(steps) != 1:
            raise NotImplementedError("fused AdamW: parameters at different step counts %s" % sorted(steps))
        self._step_cache = (key, steps.pop())
        return self._step_cache[1]

    def _bind(self, plist):
        eng = self._engine
        if eng is None:
            from hipmonocon.engine import Engine
            eng = self._engine = Engine(plist[0].device.index)
        for p in plist:
            st = self.state[p]
            if len(st) == 0:
                st['step'] = torch.tensor(0.0)
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
        sig = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]['exp_avg'].data_ptr(),
                     self.state[p]['exp_avg_sq'].data_ptr()) for p in plist)
        if sig == self._bound_sig:
            return eng
        n = len(plist)
        P, G, M, V = ((C.c_void_p * n)() for _ in range(4))
        N = (C.c_int64 * n)()
        for i, p in enumerate(plist):
            if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() or not p.grad.is_contiguous():
                raise _lib.MonoconHipError("fused AdamW needs contiguous float32 HIP parameters and gradients")
            st = self.state[p]
            for key in ('exp_avg', 'exp_avg_sq'):
                m = st[key]
                if m.dtype != torch.float32 or m.device != p.device or not m.is_contiguous():
                    raise _lib.MonoconHipError("fused AdamW: state['%s'] must be a contiguous float32 tensor on the "
                                               "parameter's device" % key)
            P[i], G[i], M[i], V[i] = p.data_ptr(), p.grad.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr()
            N[i] = p.numel()
        _lib.check(eng.h, eng.lib.mc_optim_bind(eng.h, n, P, G, M, V, N), "mc_optim_bind")
        self._bound_sig = sig
        return eng

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        group = self.param_groups[0]
        plist = [p for p in group['params'] if p.grad is not None]
        if not plist:
            return loss
        eng = self._bind(plist)
        step = self._common_step(plist) + 1
        for p in plist:
            self.state[p]['step'] += 1          # in place: the tensors' identities (the cache key) survive
        self._step_cache = (tuple((id(p), id(self.state[p]['step'])) for p in plist), step)
        if self.last_grad_norm is None:
            self.last_grad_norm = torch.zeros(1, dtype=torch.float32, device=plist[0].device)
        beta1, beta2 = group['betas']
        with torch.cuda.device(plist[0].device):
            rc = eng.lib.mc_clip_adamw_step(eng.h, float(group['lr']), float(beta1), float(beta2), float(group['eps']),
                                            float(group['weight_decay']),
                                            float(self.max_grad_norm) if self.max_grad_norm else 0.0, step,
                                            C.c_void_p(self.last_grad_norm.data_ptr()),
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(eng.h, rc, "mc_clip_adamw_step")
        # the kernels updated the parameters behind torch's back: advance their version counters
        # (host-only bookkeeping) so autograd and the engine's packed-weight cache see the change
        torch.autograd.graph.increment_version(plist)
        return loss
