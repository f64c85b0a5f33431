from .base_transforms import BaseTransform, Compose
from .default_transforms import Normalize, Pad, ToTensor, GpuNormalizePad, DeferImage, DeferredImage
from .augmentations import (Resize3D, PhotometricDistortion, RandomShift, RandomHorizontalFlip, Convert_3D_to_4D,
                            RandomCrop3D, RandomRangeCrop3D)

__all__ = ['BaseTransform', 'Compose', 'Normalize', 'Pad', 'ToTensor', 'GpuNormalizePad', 'DeferImage', 'DeferredImage',
           'Resize3D', 'PhotometricDistortion', 'RandomShift', 'RandomHorizontalFlip', 'Convert_3D_to_4D',
           'RandomCrop3D', 'RandomRangeCrop3D']
