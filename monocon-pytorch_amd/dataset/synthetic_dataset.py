"""KITTI-shaped synthetic dataset with the reference's ``collate_fn`` output contract
(dataset/monocon_dataset.py:160-200).  Used when ``cfg.DATA.ROOT == 'synthetic'`` (no KITTI files
and no cv2 in this image); the real MonoConDataset stays out of scope."""
from typing import Any, Dict, List

import torch
from torch.utils.data import Dataset

from hipmonocon import synth


class SyntheticMonoConDataset(Dataset):
    def __init__(self, length: int = 64, height: int = 384, width: int = 1280, max_objs: int = 30, seed: int = 0,
                 rank: int = 0, world: int = 1):
        self.length, self.height, self.width, self.max_objs, self.seed = length, height, width, max_objs, seed
        self.rank, self.world = rank, world

    def __len__(self):
        return self.length

    def __getitem__(self, idx: int) -> Dict[str, Any]:
        b = synth.make_batch(self.seed * 100003 + idx, 1, self.height, self.width)
        return {'img': b['img'][0], 'label': {k: v[0] for k, v in b['label'].items()}, 'calib': b['calib'][0],
                'img_metas': {'pad_shape': (self.height, self.width), 'ori_shape': (self.height, self.width),
                              'sample_idx': idx}}

    @staticmethod
    def collate_fn(batched: List[Dict[str, Any]]) -> Dict[str, Any]:
        return {'img': torch.stack([d['img'] for d in batched]),
                'label': {k: torch.stack([d['label'][k] for d in batched]) for k in batched[0]['label']},
                'calib': [d['calib'] for d in batched],
                'img_metas': {k: [d['img_metas'][k] for d in batched] for k in batched[0]['img_metas']}}

    def evaluate(self, results, eval_classes=None, verbose=False):
        """no KITTI ground truth to score against: report detection counts only."""
        n3d = sum(len(r['boxes_3d']) if isinstance(r, dict) and 'boxes_3d' in r else 0 for r in results.get('img_bbox', []))
        return {'num_results': float(len(results.get('img_bbox', []))), 'num_boxes_3d': float(n3d)}


class PooledSyntheticDataset(Dataset):
    """``length`` samples handed out as copies of ``pool`` pre-drawn ones (drawing a 384x1280 sample costs ~0.4 s: too slow to
    time a feed with).  Same sample dicts and collate as SyntheticMonoConDataset."""

    def __init__(self, length: int, height: int = 384, width: int = 1280, pool: int = 8, seed: int = 0):
        base = SyntheticMonoConDataset(length=pool, height=height, width=width, seed=seed)
        self.length, self.pool = length, [base[i] for i in range(pool)]

    collate_fn = staticmethod(SyntheticMonoConDataset.collate_fn)

    def __len__(self):
        return self.length

    def __getitem__(self, idx: int) -> Dict[str, Any]:
        d = self.pool[idx % len(self.pool)]
        return {'img': d['img'].clone(), 'label': {k: v.clone() for k, v in d['label'].items()}, 'calib': d['calib'],
                'img_metas': dict(d['img_metas'], sample_idx=idx)}
