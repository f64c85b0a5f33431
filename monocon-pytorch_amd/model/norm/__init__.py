from .attentive_norm import AttnBatchNorm2d

__all__ = ['AttnBatchNorm2d']
