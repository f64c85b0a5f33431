"""The deterministic tail of the reference's transform lists -- Normalize, Pad, ToTensor
(transforms/default_transforms.py:375-452; dataset/monocon_dataset.py:32-33,39-40) -- on the host, plus
``GpuNormalizePad`` which does all three on the device through ``mc_preprocess``.

The random training augmentations live in transforms/augmentations.py.  Pinned (round 6) to the reference's own output
(tests/golden/f4_transforms.npz, tests/test_f4_reference_golden.py: bit-equal); the arithmetic: ``(uint8 -> float32 - mean(float64)) / std(float64)`` is a float64 image, zero-padded to a
multiple of ``size_divisor`` and rounded ONCE to float32 by ``torch.Tensor(...)``.
"""
from numbers import Number
from typing import Any, Dict, List

import numpy as np
import torch

from .base_transforms import BaseTransform


class Normalize(BaseTransform):
    def __init__(self, mean: List[float], std: List[float], keep_origin: bool = False):
        super().__init__(True, False, False, False)
        self.mean = [mean] * 3 if isinstance(mean, Number) else list(mean)
        self.std = [std] * 3 if isinstance(std, Number) else list(std)
        self.keep_origin = keep_origin

    def __call__(self, data_dict: Dict[str, Any]) -> Dict[str, Any]:
        img = data_dict['img'].astype(np.float32)
        if self.keep_origin:
            data_dict['ori_img'] = img.copy()
        data_dict['img'] = (img - np.asarray(self.mean).reshape(1, 1, -1)) / np.asarray(self.std).reshape(1, 1, -1)
        return data_dict


class Pad(BaseTransform):
    def __init__(self, size_divisor: int):
        super().__init__(True, True, False, False)
        self.size_divisor = int(size_divisor)

    def __call__(self, data_dict: Dict[str, Any]) -> Dict[str, Any]:
        img = data_dict['img']
        h, w = img.shape[:2]
        d = self.size_divisor
        ph, pw = -(-h // d) * d, -(-w // d) * d
        canvas = np.zeros((ph, pw, 3), dtype=img.dtype)
        canvas[:h, :w] = img
        data_dict['img'] = canvas
        data_dict['img_metas']['pad_shape'] = (ph, pw)
        return data_dict


class ToTensor(BaseTransform):
    def __init__(self):
        super().__init__(True, False, False, True)

    def __call__(self, data_dict: Dict[str, Any]) -> Dict[str, Any]:
        if 'img' in data_dict:
            data_dict['img'] = torch.Tensor(np.ascontiguousarray(data_dict['img'])).permute(2, 0, 1)
        if 'label' in data_dict:
            # every label field becomes a float32 tensor with a leading batch axis of 1 (collate_fn concatenates them)
            data_dict['label'] = {k: torch.Tensor(np.asarray(v)).unsqueeze(0) for k, v in data_dict['label'].items()}
        return data_dict


class GpuNormalizePad(BaseTransform):
    """Normalize + Pad + ToTensor for the image in ONE device kernel (``mc_preprocess``: bit-identical to the three host
    transforms above, tests/test_input_pipeline.py): the DataLoader then ships 1 byte per pixel channel instead of 4.
    Use it as the last transform of a dataset whose ``__getitem__`` runs in the training process (num_workers=0), or
    call ``Engine.preprocess`` on a collated uint8 batch."""

    def __init__(self, engine, mean: List[float], std: List[float], size_divisor: int = 32):
        super().__init__(True, True, False, True)
        self.engine, self.mean, self.std, self.size_divisor = engine, list(mean), list(std), int(size_divisor)

    def __call__(self, data_dict: Dict[str, Any]) -> Dict[str, Any]:
        img = torch.from_numpy(np.ascontiguousarray(data_dict['img'])).to(self.engine.device)
        out, pads = self.engine.preprocess([img], self.mean, self.std, self.size_divisor)
        data_dict['img'] = out[0]
        data_dict['img_metas']['pad_shape'] = pads[0]
        if 'label' in data_dict:
            data_dict['label'] = {k: torch.Tensor(np.asarray(v)).unsqueeze(0) for k, v in data_dict['label'].items()}
        return data_dict
