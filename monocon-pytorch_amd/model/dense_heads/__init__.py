"""MonoCon dense heads: prediction maps, targets + losses, top-K decode."""
from .monocon_heads import MonoConDenseHeads

__all__ = ("MonoConDenseHeads",)
