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


# ---------------------------------------------------------------------------------------------- image work on the device
AUG_PARAMS = 24
AUG_COLOUR, AUG_BRIGHTNESS, AUG_CONTRAST_BEFORE, AUG_SATURATION, AUG_HUE, AUG_CONTRAST_AFTER, AUG_PERMUTATION = 1, 2, 4, 8, 16, 32, 64
AUG_SHIFT, AUG_FLIP, AUG_WINDOW = 128, 256, 512
_AUG_ORDER = ('colour', 'shift', 'flip', 'window')       # the order the kernel composes them in = the reference's train list


class DeferImage(BaseTransform):
    """FIRST transform of a list whose image work runs on the device: from here on PhotometricDistortion, RandomShift,
    RandomHorizontalFlip and the crops draw their random numbers and move labels and calibration as always, but leave the
    frame alone and note what they would have done to it (``data_dict['img_ops']``).  DeferredImage, the LAST transform,
    turns the notes into the 24 parameters of ``mc_preprocess_augmented``."""

    def __init__(self):
        super().__init__(True, False, False, False)

    def __call__(self, data_dict: Dict[str, Any]) -> Dict[str, Any]:
        data_dict['img_ops'] = []
        return data_dict


class DeferredImage(BaseTransform):
    """stands where Normalize + Pad + ToTensor stand in the host lists: the sample's ``img`` becomes the RAW uint8 frame,
    zero-padded to the padded size as (Hp, Wp, 3), and ``img_aug`` the float32 parameters the device kernel needs to produce
    -- bit for bit -- the float32 CHW frame those three (and the augmentations in front of them) would have
    (``Engine.preprocess_augmented``; hipmonocon.feed.DevicePrefetcher calls it on the uploaded batch).  Labels become tensors
    as under ToTensor."""

    def __init__(self, size_divisor: int = 32):
        super().__init__(True, True, False, True)
        self.size_divisor = int(size_divisor)

    def __call__(self, data_dict: Dict[str, Any]) -> Dict[str, Any]:
        img = data_dict['img']
        if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 3:
            raise TypeError("DeferredImage needs the decoded uint8 (H, W, 3) frame; a host transform has changed it "
                            "(put DeferImage first in the list)")
        ops = data_dict.pop('img_ops', [])
        names = [n for n, _ in ops]
        if len(set(names)) != len(names) or names != [n for n in _AUG_ORDER if n in names]:
            raise NotImplementedError("deferred image operations %s: the device kernel composes at most one each of %s, in "
                                      "that order" % (names, list(_AUG_ORDER)))
        h, w = img.shape[:2]
        prm = np.zeros(AUG_PARAMS, np.float32)
        prm[0], prm[1] = h, w
        flags = 0
        for name, val in ops:
            if name == 'colour':
                flags |= AUG_COLOUR
                for key, bit, at in (('brightness', AUG_BRIGHTNESS, 3), ('contrast_before', AUG_CONTRAST_BEFORE, 4),
                                     ('saturation', AUG_SATURATION, 5), ('hue', AUG_HUE, 6), ('contrast_after', AUG_CONTRAST_AFTER, 7)):
                    if val[key] is not None:
                        flags |= bit
                        prm[at] = val[key]
                if val['permutation'] is not None:
                    flags |= AUG_PERMUTATION
                    prm[8:11] = val['permutation']
            elif name == 'shift':
                flags |= AUG_SHIFT
                prm[11], prm[12] = val
            elif name == 'flip':
                flags |= AUG_FLIP
            else:
                flags |= AUG_WINDOW
                prm[13:17] = val
        prm[2] = flags
        d = self.size_divisor
        ph, pw = -(-h // d) * d, -(-w // d) * d
        canvas = np.zeros((ph, pw, 3), np.uint8)
        canvas[:h, :w] = img
        data_dict['img'] = torch.from_numpy(canvas)
        data_dict['img_aug'] = torch.from_numpy(prm)
        data_dict['img_metas']['pad_shape'] = (ph, pw)
        if 'label' in data_dict:
            data_dict['label'] = {k: torch.Tensor(np.asarray(v)).unsqueeze(0) for k, v in data_dict['label'].items()}
        return data_dict
