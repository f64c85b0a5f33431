"""File-backed KITTI dataset with the reference's sample contract (dataset/base_dataset.py:19-91,
dataset/monocon_dataset.py:43-200): ``__getitem__`` -> {'img' (3,Hp,Wp) float32, 'img_metas', 'calib', 'label'};
``collate_fn`` -> the ``data_dict`` the detector consumes.

Differences from the reference, all on the host side of the hot path:
  * PNG frames are decoded with PIL (no cv2 in the image): a PNG is lossless, so the RGB uint8 array equals
    ``cv2.cvtColor(cv2.imread(path), COLOR_BGR2RGB)``;
  * the split lists (ImageSets/{train,val,trainval,test}.txt -- data the reference ships inside its source tree) are
    looked up under ``<base_root>/ImageSets``, ``$MONOCON_IMAGESETS`` or an explicit ``imageset_dir``; without one
    the sample ids are taken from the image directory listing;
  * the random training augmentations (transforms/augmentations.py) are numpy restatements without cv2 and can draw from
    an explicit ``np.random.Generator`` (``aug_rng=``) so that data-parallel ranks / loader workers are seeded apart;
  * ``evaluate`` runs the AP evaluator of engine/kitti_eval (HIP overlap kernels + native matching instead of numba);
    ``write_kitti_results`` additionally writes the benchmark's txt submission files.
The label assembly restates monocon_dataset.py:89-158 by reading (the reference module needs cv2 to import): the
per-object quantities it copies are pinned against the reference's data classes (tests/golden/kitti_objects.npz), the
filter rules are covered by a fixture built so that each rule fires exactly once (tests/test_input_pipeline.py).
"""
import os
from typing import Any, Dict, List, Optional

import numpy as np
import torch
from torch.utils.data import Dataset

from transforms import (Compose, DeferImage, DeferredImage, Normalize, Pad, PhotometricDistortion, RandomCrop3D, RandomHorizontalFlip, RandomShift,
                        ToTensor)
from utils.data_classes import KITTICalibration, KITTIMultiObjects

DEFAULT_FILTER_CONFIG = {'min_height': 25, 'min_depth': 2, 'max_depth': 65, 'max_truncation': 0.5, 'max_occlusion': 2}
IMG_MEAN, IMG_STD = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]


def default_transforms(device_image: bool = False):
    """the test / validation list (dataset/monocon_dataset.py:38-42).  ``device_image``: the image work is left to the device
    (transforms.DeferredImage: the sample carries the raw uint8 frame + 24 parameters for ``mc_preprocess_augmented``)"""
    if device_image:
        return [DeferImage(), DeferredImage(size_divisor=32)]
    return [Normalize(mean=IMG_MEAN, std=IMG_STD), Pad(size_divisor=32), ToTensor()]


def default_train_transforms(rng=None, device_image: bool = False):
    """the reference's training list (dataset/monocon_dataset.py:22-35).  ``rng``: one np.random.Generator for all the random
    decisions (seed it per rank / worker / epoch); None = numpy's global state.  ``device_image``: the same random decisions,
    labels and calibration, the pixels left to the device (the float32 colour round trip alone costs a worker ~80 ms per frame)"""
    aug = [PhotometricDistortion(brightness_delta=32, contrast_range=(0.5, 1.5), saturation_range=(0.5, 1.5), hue_delta=18, rng=rng),
           RandomShift(prob=0.5, shift_range=(-32, 32), hide_kpts_in_shift_area=True, rng=rng),
           RandomHorizontalFlip(prob=0.5, rng=rng),
           RandomCrop3D(prob=0.5, crop_size=(320, 960), hide_kpts_in_crop_area=True, rng=rng)]
    if device_image:
        return [DeferImage()] + aug + [DeferredImage(size_divisor=32)]
    return aug + default_transforms()


def _find_imageset(base_root: str, split: str, imageset_dir: Optional[str]) -> Optional[str]:
    for d in (imageset_dir, os.path.join(base_root, 'ImageSets'), os.environ.get('MONOCON_IMAGESETS')):
        if d and os.path.isfile(os.path.join(d, split + '.txt')):
            return os.path.join(d, split + '.txt')
    return None


class BaseKITTIMono3DDataset(Dataset):
    def __init__(self, base_root: str, split: str, pad_divisor: int = 32, imageset_dir: Optional[str] = None,
                 file_prefix: Optional[List[str]] = None):
        super().__init__()
        if not os.path.isdir(base_root):
            raise FileNotFoundError("KITTI root %r is not a directory" % base_root)
        if split not in ('train', 'val', 'trainval', 'test'):
            raise ValueError("split must be one of train / val / trainval / test, got %r" % split)
        self.base_root, self.split, self.pad_divisor = base_root, split, pad_divisor
        sub = 'testing' if split == 'test' else 'training'
        self.image_dir = os.path.join(base_root, sub, 'image_2')
        self.calib_dir = os.path.join(base_root, sub, 'calib')
        self.label_dir = None if split == 'test' else os.path.join(base_root, sub, 'label_2')
        if file_prefix is None:
            lst = _find_imageset(base_root, split, imageset_dir)
            if lst is not None:
                with open(lst) as f:
                    file_prefix = [ln.strip() for ln in f if ln.strip()]
            else:
                file_prefix = sorted(os.path.splitext(f)[0] for f in os.listdir(self.image_dir) if f.endswith('.png'))
        self.file_prefix = list(file_prefix)
        self.image_files = [os.path.join(self.image_dir, p + '.png') for p in self.file_prefix]
        self.calib_files = [os.path.join(self.calib_dir, p + '.txt') for p in self.file_prefix]
        self.label_files = [] if self.label_dir is None else [os.path.join(self.label_dir, p + '.txt') for p in self.file_prefix]

    def __len__(self):
        return len(self.file_prefix)

    def load_image(self, idx: int):
        from PIL import Image
        with Image.open(self.image_files[idx]) as im:
            arr = np.asarray(im.convert('RGB'), dtype=np.uint8)
        metas = {'idx': idx, 'split': self.split, 'sample_idx': int(os.path.basename(self.image_files[idx]).split('.')[0]),
                 'image_path': self.image_files[idx], 'ori_shape': arr.shape[:2]}
        return arr, metas

    def load_calib(self, idx: int) -> KITTICalibration:
        return KITTICalibration(self.calib_files[idx])

    def load_label(self, idx: int) -> KITTIMultiObjects:
        return KITTIMultiObjects.get_objects_from_label(self.label_files[idx], self.load_calib(idx))


class MonoConDataset(BaseKITTIMono3DDataset):
    def __init__(self, base_root: str, split: str, max_objs: int = 30, transforms=None, filter_configs: Dict[str, Any] = None,
                 aug_rng=None, device_image: bool = False, **kwargs):
        super().__init__(base_root=base_root, split=split, **kwargs)
        self.max_objs = max_objs
        if transforms is None:          # as the reference: augmentations for 'train' only (monocon_dataset.py:58-63)
            transforms = default_train_transforms(aug_rng, device_image) if split == 'train' else default_transforms(device_image)
        self.transforms = Compose(transforms)
        # One Generator serves every random transform.  A DataLoader with num_workers > 0 COPIES it into each forked
        # worker, which would then all draw the same augmentation sequence (the engine's worker_init_fn only
        # reseeds numpy's global state): the first __getitem__ inside a worker re-seeds the shared object in place from
        # (its current stream, worker id, the worker's torch seed).
        self._aug_rng, self._aug_rng_worker = aug_rng, None
        cfg = dict(DEFAULT_FILTER_CONFIG)
        if filter_configs is not None:
            unknown = [k for k in filter_configs if k not in DEFAULT_FILTER_CONFIG]
            if unknown:
                raise ValueError("unknown filter keys %s (valid: %s)" % (unknown, list(DEFAULT_FILTER_CONFIG)))
            cfg.update(filter_configs)
        self.filter_configs = cfg
        for k, v in cfg.items():
            setattr(self, k, v)

    def _empty_labels(self) -> Dict[str, np.ndarray]:
        m = self.max_objs
        return {'gt_bboxes': np.zeros((m, 4), np.float32), 'gt_labels': np.zeros(m, np.uint8),
                'gt_bboxes_3d': np.zeros((m, 7), np.float32), 'gt_labels_3d': np.zeros(m, np.uint8),
                'centers2d': np.zeros((m, 2), np.float32), 'depths': np.zeros(m, np.float32),
                'gt_kpts_2d': np.zeros((m, 18), np.float32), 'gt_kpts_valid_mask': np.zeros((m, 9), np.uint8),
                'mask': np.zeros((m,), np.bool_)}

    def build_labels(self, objects: KITTIMultiObjects, input_hw) -> Dict[str, np.ndarray]:
        """camera-2 / local-yaw labels of the objects that pass the filter; an object keeps ITS index among the
        non-DontCare objects of the file as its row (rows of rejected objects stay zero, mask False)"""
        objects.convert_cam(src_cam=0, dst_cam=2)
        objects.convert_yaw(src_type='global', dst_type='local')
        L = self._empty_labels()
        H, W = input_hw
        for row, obj in enumerate(objects):
            if row >= self.max_objs:
                raise IndexError("more than max_objs=%d labelled objects in one frame" % self.max_objs)
            if obj.occlusion > self.max_occlusion or obj.truncation > self.max_truncation:
                continue
            if obj.box2d[3] - obj.box2d[1] < self.min_height:
                continue
            center = obj.projected_center
            if not (self.min_depth <= center[-1] <= self.max_depth):
                continue
            kpts = obj.projected_kpts                         # (9,3): u, v, in-front flag
            inside = (kpts[:, 0] >= 0) & (kpts[:, 0] <= W) & (kpts[:, 1] >= 0) & (kpts[:, 1] <= H)
            kpts[inside, 2] = 2
            L['gt_bboxes'][row] = obj.box2d
            L['gt_labels'][row] = L['gt_labels_3d'][row] = obj.cls_num
            L['gt_bboxes_3d'][row] = np.concatenate([obj.loc, obj.dim, [obj.ry]])
            L['centers2d'][row], L['depths'][row] = center[:2], center[2]
            L['gt_kpts_2d'][row], L['gt_kpts_valid_mask'][row] = kpts[:, :2].reshape(-1), kpts[:, 2]
            L['mask'][row] = True
        return L

    def _reseed_in_worker(self):
        info = torch.utils.data.get_worker_info()
        rng = getattr(self, '_aug_rng', None)      # (an instance restored from a foreign pickle has no such attribute)
        if rng is None or info is None or getattr(self, '_aug_rng_worker', None) == info.id:
            return
        base = int(self._aug_rng.bit_generator.random_raw())          # same in every copy: the parent's stream position
        ss = np.random.SeedSequence([base, info.id, int(info.seed) % (2 ** 63)])
        self._aug_rng.bit_generator.state = type(self._aug_rng.bit_generator)(ss).state
        self._aug_rng_worker = info.id

    def __getitem__(self, idx: int) -> Dict[str, Any]:
        self._reseed_in_worker()
        image, metas = self.load_image(idx)
        out = {'img': image, 'img_metas': metas, 'calib': self.load_calib(idx)}
        if self.label_files:
            out['label'] = self.build_labels(self.load_label(idx), image.shape[:2])
        return self.transforms(out)

    @staticmethod
    def collate_fn(batched: List[Dict[str, Any]]) -> Dict[str, Any]:
        out = {'img': torch.stack([d['img'] for d in batched]),
               'img_metas': {k: [d['img_metas'][k] for d in batched] for k in batched[0]['img_metas']},
               'calib': [d['calib'] for d in batched]}
        if 'label' in batched[0]:
            out['label'] = {k: torch.cat([d['label'][k] for d in batched], dim=0) for k in batched[0]['label']}
        if 'img_aug' in batched[0]:          # transforms.DeferredImage: raw frames + the device kernel's parameters
            out['img_aug'] = torch.stack([d['img_aug'] for d in batched])
        return out

    def collect_gt_infos(self, verbose: bool = False) -> List[Dict[str, Any]]:
        """per frame {'image', 'calib', 'annos'}: the annotation dict of ALL labelled objects, DontCare included -- the
        evaluator needs them (base_dataset.py:85-115)"""
        out = []
        for idx in range(len(self)):
            _, metas = self.load_image(idx)
            objs = self.load_label(idx)
            if objs.ignore_dontcare:
                objs = objs.original_objects
            out.append({'image': metas, 'calib': self.load_calib(idx).get_info_dict(), 'annos': objs.info_dict})
        return out

    def write_kitti_results(self, kitti_format_results: Dict[str, Any], save_dir: str) -> None:
        """one txt per frame and result set in the benchmark's submission format (not in the reference, which only keeps
        the dicts in memory)"""
        for name, results in kitti_format_results.items():
            d = os.path.join(save_dir, name)
            os.makedirs(d, exist_ok=True)
            for r in results:
                ids = np.asarray(r['sample_idx']).reshape(-1)
                sid = int(ids[0]) if len(ids) else 0
                with open(os.path.join(d, '%06d.txt' % sid), 'w') as f:
                    for i in range(len(r['name'])):
                        bb, dm, lc = r['bbox'][i], r['dimensions'][i], r['location'][i]
                        f.write('%s -1 -1 %.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f\n'
                                % (r['name'][i], r['alpha'][i], bb[0], bb[1], bb[2], bb[3], dm[1], dm[2], dm[0],
                                   lc[0], lc[1], lc[2], r['rotation_y'][i], r['score'][i]))

    def evaluate(self, kitti_format_results: Dict[str, Any], eval_classes: List[str] = ('Pedestrian', 'Cyclist', 'Car'),
                 eval_types: List[str] = ('bbox', 'bev', '3d'), verbose: bool = True, save_path: str = None) -> Dict[str, float]:
        """KITTI AP40 of every result set in ``kitti_format_results`` ({'img_bbox': [...], 'img_bbox2d': [...]}) against the
        labels of this split -> {'<set>/KITTI/<Class>_<3D|BEV|2D>_AP40_<difficulty>_<strict|loose>': value}, written as
        json to ``save_path`` when given (base_dataset.py:117-152).  The overlap kernels run on the GPU and the matching in
        native host code (engine/kitti_eval)."""
        import json
        from engine.kitti_eval import kitti_eval
        if self.split == 'test' or not self.label_files:
            raise RuntimeError("evaluate() needs labels: the %r split has none" % self.split)
        if getattr(self, 'gt_annos', None) is None:
            self.gt_annos = [info['annos'] for info in self.collect_gt_infos(verbose=verbose)]
        ap_dict = {}
        for name, result in kitti_format_results.items():
            if len(result) != len(self.gt_annos):
                raise ValueError("%s: %d result frames for %d labelled frames" % (name, len(result), len(self.gt_annos)))
            types = ['bbox'] if '2d' in name else list(eval_types)
            result_string, result_dict = kitti_eval(gt_annos=self.gt_annos, dt_annos=result, current_classes=list(eval_classes),
                                                    eval_types=types)
            for ap_type, ap_value in result_dict.items():
                ap_dict['%s/%s' % (name, ap_type)] = float('%.4f' % ap_value)
            if verbose and '2d' not in name:
                print(result_string)
        if save_path is not None:
            with open(save_path, 'w') as f:
                json.dump(ap_dict, f)
        return ap_dict


class RepeatedDataset(torch.utils.data.Dataset):
    """``length`` samples out of a shorter dataset, index modulo its length (each access runs the dataset's own __getitem__:
    decode, labels, transforms).  For timing a feed on a small tree of real frames (bench.py); picklable like its base."""

    def __init__(self, base, length: int):
        self.base, self.length = base, int(length)
        self.collate_fn = base.collate_fn

    def __len__(self):
        return self.length

    def __getitem__(self, idx: int):
        return self.base[idx % len(self.base)]
