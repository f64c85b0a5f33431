"""Transform protocol of the input pipeline: a callable ``data_dict -> data_dict`` over the keys 'img' (HWC ndarray
until ToTensor), 'img_metas', 'calib', 'label' (reference transforms/base_transforms.py:4-44)."""
from typing import Any, Dict, List


class BaseTransform:
    def __init__(self, change_img: bool, change_metas: bool, change_calib: bool, change_label: bool):
        self._change_img, self._change_metas = change_img, change_metas
        self._change_calib, self._change_label = change_calib, change_label

    def __call__(self, data_dict: Dict[str, Any]) -> Dict[str, Any]:
        raise NotImplementedError

    def __repr__(self):
        args = ', '.join('%s=%s' % (k, v) for k, v in self.__dict__.items() if not k.startswith('_') and not callable(v))
        return '%s(%s)' % (self.__class__.__name__, args)


class Compose:
    def __init__(self, transforms: List[BaseTransform]):
        self.transforms = list(transforms)

    def __call__(self, data_dict: Dict[str, Any]) -> Dict[str, Any]:
        for t in self.transforms:
            data_dict = t(data_dict)
        return data_dict

    def __repr__(self):
        return 'Compose(%s)' % ', '.join(repr(t) for t in self.transforms)
