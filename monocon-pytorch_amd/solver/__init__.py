"""One-cycle LR / momentum schedule and the fused clip + AdamW step."""
from .fused_adamw import AdamW
from .cyclic_scheduler import CyclicScheduler

__all__ = ("AdamW", "CyclicScheduler")
