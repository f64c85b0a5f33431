from .cyclic_scheduler import CyclicScheduler
from .fused_adamw import AdamW

__all__ = ['CyclicScheduler', 'AdamW']
