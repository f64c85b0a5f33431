"""One-cycle cosine schedule on the learning rate and on AdamW's beta1.

Same constructor and behaviour as reference solver/cyclic_scheduler.py:8-76 -- including that
phases are keyed on ``_step_count``, which is already 1 when the first optimizer step runs
(``_LRScheduler.__init__`` performs an initial ``step()``), and that the optimizer class must be
called ``AdamW`` (torch.optim.AdamW or solver.AdamW, the fused HIP optimizer).  Pure host code.
"""
import math
from typing import List, Tuple

from torch.optim.lr_scheduler import _LRScheduler
from torch.optim.optimizer import Optimizer


def _cos_anneal(start: float, end: float, factor: float) -> float:
    return end + 0.5 * (start - end) * (math.cos(math.pi * factor) + 1)


class CyclicScheduler(_LRScheduler):
    def __init__(self, optimizer: Optimizer, total_steps: int, target_lr_ratio: Tuple[float, float] = (10, 1e-4),
                 target_momentum_ratio: Tuple[float, float] = (0.85 / 0.95, 1.), period_up: float = 0.4):
        assert optimizer.__class__.__name__ == 'AdamW', "Currently, this scheduler only supports 'AdamW' optimizer."
        self.total_steps = total_steps
        self.target_lr_ratio = target_lr_ratio
        self.target_momentum_ratio = target_momentum_ratio
        self.period_up = period_up
        self.steps_up = int(total_steps * period_up)
        for group in optimizer.param_groups:
            group.setdefault('initial_momentum', group['betas'][0])
        self.base_momentum = [group['initial_momentum'] for group in optimizer.param_groups]
        super().__init__(optimizer, last_epoch=-1)

    def _phase(self):
        """(is_up, factor) for the current ``_step_count``."""
        if self._step_count < self.steps_up:
            return True, self._step_count / self.steps_up
        return False, (self._step_count - self.steps_up) / (self.total_steps - self.steps_up)

    def get_lr(self) -> List[float]:
        self.set_momentum()
        up, f = self._phase()
        hi, lo = self.target_lr_ratio
        if up:
            return [_cos_anneal(b, b * hi, f) for b in self.base_lrs]
        return [_cos_anneal(b * hi, b * lo, f) for b in self.base_lrs]

    def set_momentum(self):
        up, f = self._phase()
        lo, hi = self.target_momentum_ratio
        for group, m in zip(self.optimizer.param_groups, self.base_momentum):
            mom = _cos_anneal(m, m * lo, f) if up else _cos_anneal(m * lo, m * hi, f)
            group['betas'] = (mom, group['betas'][1])
