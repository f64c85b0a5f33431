from .monocon_heads import MonoConDenseHeads

__all__ = ['MonoConDenseHeads']
