"""Attentive normalisation module of the dense heads (parameter container; the math lives in the HIP head kernels)."""
from .attentive_norm import AttnBatchNorm2d

__all__ = ("AttnBatchNorm2d",)
