from .base_transforms import BaseTransform, Compose
from .default_transforms import Normalize, Pad, ToTensor, GpuNormalizePad

__all__ = ['BaseTransform', 'Compose', 'Normalize', 'Pad', 'ToTensor', 'GpuNormalizePad']
