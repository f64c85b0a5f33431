"""Host-side helpers of the engine (reference utils/engine_utils.py:19-108), rank-aware."""
import pickle
import random
from contextlib import redirect_stdout
from datetime import datetime
from typing import Any, Dict

import numpy as np
import torch
import torch.nn as nn

from config.cfgnode import CfgNode
from config.monocon_configs import _C as cfg


def generate_random_seed(seed: int = None) -> int:
    return seed if (seed is not None and seed != -1) else int(np.random.randint(2 ** 31))


def set_random_seed(seed: int) -> None:
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def count_trainable_params(model: nn.Module):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def get_default_cfg() -> CfgNode:
    return cfg.clone()


def load_cfg(cfg_file: str) -> CfgNode:
    c = get_default_cfg()
    c.set_new_allowed(True)
    c.merge_from_file(cfg_file)
    return c


def export_cfg(cfg: CfgNode, save_path: str) -> None:
    with open(save_path, 'w') as f, redirect_stdout(f):
        print(cfg.dump())


def move_data_device(data_dict: Dict[str, Any], device: str = None) -> Dict[str, Any]:
    if device is None or not torch.cuda.is_available():
        device = 'cpu'
    for k, v in data_dict.items():
        if isinstance(v, torch.Tensor):
            data_dict[k] = v.to(device, non_blocking=True)
    if 'label' in data_dict:
        data_dict['label'] = {k: v.to(device, non_blocking=True) for k, v in data_dict['label'].items()}
    return data_dict


def reduce_loss_dict(loss_dict: Dict[str, torch.Tensor]) -> torch.Tensor:
    return sum(v for v in loss_dict.values())


def tprint(message: str, indent: bool = False) -> None:
    stamp = str(datetime.now())[:-7]
    print(('\n' if indent else '') + '[%s] %s' % (stamp, message))


def progress_to_string_bar(current_prog: int, total_prog: int, bins: int = 10, non_filled_chr: str = ' ',
                           filled_chr: str = '#') -> str:
    frac = current_prog / total_prog
    assert 0.0 <= frac <= 1.0
    filled = int(frac / (1 / bins))
    return '[%s%s][%5.2f%%]' % (filled_chr * filled, non_filled_chr * (bins - filled), frac * 100)


# ---- checkpoints written by the reference ---------------------------------------------------------------------------
class OpaqueReferenceObject:
    """stands in for an object of a class this repository does not have, found inside a reference checkpoint"""

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {'state': state})


class _TolerantUnpickler(pickle.Unpickler):
    """The reference's ``save_checkpoint`` pickles engine attributes next to the state dicts (engine/base_engine.py:155-189)
    and -- through a missing comma in its exclusion list (:171) -- the whole ``test_dataset`` object with its transforms.
    Most of those classes exist here under the same module paths (dataset.monocon_dataset.MonoConDataset,
    transforms.default_transforms.Normalize / Pad / ToTensor, transforms.base_transforms.Compose); the rest (e.g.
    transforms.geo_aware_transforms.*, which needs cv2 in the reference) must not make the load fail: a class that cannot
    be resolved becomes an :class:`OpaqueReferenceObject` subclass carrying the pickled attributes."""

    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except (ImportError, AttributeError):
            return type(name, (OpaqueReferenceObject,), {'__module__': module})


class _TolerantPickle:
    """``pickle_module`` argument for ``torch.load``"""
    __name__ = 'tolerant_pickle'
    Unpickler = _TolerantUnpickler
    load = staticmethod(lambda f, **kw: _TolerantUnpickler(f, **kw).load())
    loads = staticmethod(pickle.loads)
    dumps = staticmethod(pickle.dumps)
    dump = staticmethod(pickle.dump)
    Pickler = pickle.Pickler
    HIGHEST_PROTOCOL = pickle.HIGHEST_PROTOCOL


def load_checkpoint_file(ckpt_file: str) -> Dict[str, Any]:
    """``torch.load`` of a checkpoint in the reference's layout ({'engine_attrs': ..., 'state_dict': {'model', 'optimizer',
    'scheduler'}}), written by this repository's engine OR by the reference's: full unpickling (the file holds plain
    Python objects, torch >= 2.6 needs weights_only=False for that), tensors mapped to the CPU, unknown classes opaque."""
    return torch.load(ckpt_file, map_location='cpu', weights_only=False, pickle_module=_TolerantPickle)
