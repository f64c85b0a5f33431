"""The random training augmentations of the reference's default train pipeline (dataset/monocon_dataset.py:22-35):
PhotometricDistortion, RandomShift, RandomHorizontalFlip (transforms/default_transforms.py:52-373), RandomCrop3D and
RandomRangeCrop3D (transforms/geo_aware_transforms.py:14-418), plus Resize3D and Convert_3D_to_4D
(default_transforms.py:15-49, :460-482).  Same class names, constructor arguments, ``img_metas`` keys and label
bookkeeping; host-side numpy, no cv2:

  * the BGR <-> HSV conversions of PhotometricDistortion are written out in numpy after OpenCV's float32 formulas
    (hue in degrees, saturation and value unscaled);
  * Resize3D resamples with torch's bilinear interpolation (half-pixel centres, like cv2.INTER_LINEAR on floats);
  * every transform draws from ONE source: an ``np.random.Generator`` handed in as ``rng=`` (what a data-parallel
    loader wants: seed it per (rank, worker, epoch)) or, by default, numpy's global state -- the reference mixes
    numpy's and python's global generators.  The ORDER and kind of the random decisions follow the reference.

Parity unpinned: the reference modules import cv2 (absent here), and the transforms are random.  tests/
test_augmentations.py checks the invariants instead (projection consistency of labels and calibration after shift /
flip, identity when a transform does not fire, colour round trips, mask bookkeeping).
"""
from typing import Any, Dict, Tuple, Union

import numpy as np

from .base_transforms import BaseTransform

LABEL_ROWS_2D = ('gt_bboxes', 'gt_bboxes_3d', 'centers2d', 'gt_kpts_2d', 'gt_kpts_valid_mask')
LABEL_ROWS_1D = ('gt_labels', 'gt_labels_3d', 'depths')


class _Draw:
    """the random decisions a transform needs, from a Generator or from numpy's global state"""

    def __init__(self, rng=None):
        self.rng = rng

    def coin(self) -> bool:                                   # reference: random.randint(2)
        return bool(self.rng.integers(2) if self.rng is not None else np.random.randint(2))

    def unit(self) -> float:                                  # reference: random.random()
        return float(self.rng.random() if self.rng is not None else np.random.random())

    def uniform(self, lo, hi) -> float:
        return float(self.rng.uniform(lo, hi) if self.rng is not None else np.random.uniform(lo, hi))

    def between(self, lo: int, hi: int) -> int:               # inclusive on both ends, like python's random.randint
        return int(self.rng.integers(lo, hi + 1) if self.rng is not None else np.random.randint(lo, hi + 1))

    def permutation(self, n: int):
        return self.rng.permutation(n) if self.rng is not None else np.random.permutation(n)


def keep_objects(label: Dict[str, np.ndarray], keep: np.ndarray) -> None:
    """zero the rows of every label field whose object is dropped and store the new mask (the bookkeeping the reference
    repeats in RandomShift / RandomCrop3D / RandomRangeCrop3D)"""
    keep = np.asarray(keep, dtype=bool)
    for k in LABEL_ROWS_2D:
        if k in label:
            label[k] = label[k] * keep[:, None].astype(label[k].dtype)
    for k in LABEL_ROWS_1D:
        if k in label:
            label[k] = label[k] * keep.astype(label[k].dtype)
    label['mask'] = keep


# ------------------------------------------------------------------------------------------------------ colour
_EPS = np.float32(np.finfo(np.float32).eps)


def bgr_to_hsv(img: np.ndarray) -> np.ndarray:
    """float32 BGR -> (H in degrees [0, 360), S, V); V = max, S = (max - min) / V"""
    img = img.astype(np.float32)
    b, g, r = img[..., 0], img[..., 1], img[..., 2]
    v = np.maximum(np.maximum(b, g), r)
    diff = v - np.minimum(np.minimum(b, g), r)
    s = diff / (np.abs(v) + _EPS)
    k = np.float32(60.0) / (diff + _EPS)
    h = np.where(v == r, (g - b) * k, np.where(v == g, (b - r) * k + np.float32(120.0), (r - g) * k + np.float32(240.0)))
    h = np.where(h < 0, h + np.float32(360.0), h)
    return np.stack([h, s, v], axis=-1).astype(np.float32)


def hsv_to_bgr(img: np.ndarray) -> np.ndarray:
    img = img.astype(np.float32)
    h, s, v = img[..., 0] / np.float32(60.0), img[..., 1], img[..., 2]
    h = np.mod(h, np.float32(6.0))
    sector = np.floor(h)
    f = h - sector
    sector = sector.astype(np.int64) % 6
    tab = np.stack([v, v * (1 - s), v * (1 - s * f), v * (1 - s * (1 - f))], axis=-1)       # v, p, q, t
    order = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])     # (b, g, r) per sector
    idx = order[sector]                                                                      # (..., 3)
    out = np.take_along_axis(tab, idx, axis=-1)
    return np.where((s == 0)[..., None], v[..., None], out).astype(np.float32)


class PhotometricDistortion(BaseTransform):
    """brightness, contrast (before or after the HSV stage), saturation, hue, channel swap -- each with probability 1/2"""

    def __init__(self, brightness_delta: int = 32, contrast_range: Tuple[float, float] = (0.5, 1.5),
                 saturation_range: Tuple[float, float] = (0.5, 1.5), hue_delta: int = 18, rng=None):
        super().__init__(True, False, False, False)
        self.brightness_delta = brightness_delta
        self.contrast_lower, self.contrast_upper = contrast_range
        self.saturation_lower, self.saturation_upper = saturation_range
        self.hue_delta = hue_delta
        self._draw = _Draw(rng)

    def draw(self) -> Dict[str, Any]:
        """the random decisions of one call, in the reference's order (they do not depend on the image): None = not applied"""
        d = self._draw
        p = {'brightness': None, 'contrast_before': None, 'saturation': None, 'hue': None, 'contrast_after': None, 'permutation': None}
        if d.coin():
            p['brightness'] = np.float32(d.uniform(-self.brightness_delta, self.brightness_delta))
        contrast_first = d.coin()
        if contrast_first and d.coin():
            p['contrast_before'] = np.float32(d.uniform(self.contrast_lower, self.contrast_upper))
        if d.coin():
            p['saturation'] = np.float32(d.uniform(self.saturation_lower, self.saturation_upper))
        if d.coin():
            p['hue'] = np.float32(d.uniform(-self.hue_delta, self.hue_delta))
        if not contrast_first and d.coin():
            p['contrast_after'] = np.float32(d.uniform(self.contrast_lower, self.contrast_upper))
        if d.coin():
            p['permutation'] = np.asarray(d.permutation(3))
        return p

    @staticmethod
    def apply(img: np.ndarray, p: Dict[str, Any]) -> np.ndarray:
        img = img.astype(np.float32)[:, :, ::-1]                              # RGB -> BGR
        if p['brightness'] is not None:
            img = img + p['brightness']
        if p['contrast_before'] is not None:
            img = img * p['contrast_before']
        hsv = bgr_to_hsv(img)
        if p['saturation'] is not None:
            hsv[..., 1] = hsv[..., 1] * p['saturation']
        if p['hue'] is not None:
            hue = hsv[..., 0] + p['hue']
            hue = np.where(hue > 360, hue - 360, hue)
            hsv[..., 0] = np.where(hue < 0, hue + 360, hue)
        img = hsv_to_bgr(hsv)
        if p['contrast_after'] is not None:
            img = img * p['contrast_after']
        if p['permutation'] is not None:
            img = img[..., p['permutation']]
        return np.ascontiguousarray(img[:, :, ::-1])                          # BGR -> RGB

    def __call__(self, data_dict: Dict[str, Any]) -> Dict[str, Any]:
        p = self.draw()
        if 'img_ops' in data_dict:                                            # DeferImage: the device does the pixels
            data_dict['img_ops'].append(('colour', p))
        else:
            data_dict['img'] = self.apply(data_dict['img'], p)
        return data_dict


# ------------------------------------------------------------------------------------------------------ geometry
class RandomShift(BaseTransform):
    """integer translation of image, principal point and 2D labels; objects whose clipped box collapses are dropped.
    (``hide_kpts_in_shift_area`` is accepted and, as in the reference, has no effect: its filter is never invoked.)"""

    def __init__(self, prob: float = 0.50, shift_range: Tuple[float, float] = (-32.0, 32.0),
                 hide_kpts_in_shift_area: bool = True, rng=None):
        super().__init__(True, True, True, True)
        assert 0.0 <= prob <= 1.0
        assert len(shift_range) == 2, "Argument 'shift_range' must be given as a tuple of length 2."
        self.prob, self.shift_range, self.hide_kpts_in_shift_area = prob, shift_range, hide_kpts_in_shift_area
        self._draw = _Draw(rng)

    @staticmethod
    def _unchanged(data_dict):
        data_dict['img_metas']['is_shifted'] = False
        data_dict['img_metas']['shift_params'] = (0, 0)
        return data_dict

    def __call__(self, data_dict: Dict[str, Any]) -> Dict[str, Any]:
        d = self._draw
        if d.unit() >= self.prob:
            return self._unchanged(data_dict)
        metas, label = data_dict['img_metas'], data_dict['label']
        H, W = metas['ori_shape']
        sx, sy = int(d.uniform(*self.shift_range)), int(d.uniform(*self.shift_range))      # truncated towards zero
        boxes = label['gt_bboxes'].copy()
        boxes[..., 0::2] = np.clip(boxes[..., 0::2] + sx, 0, W)
        boxes[..., 1::2] = np.clip(boxes[..., 1::2] + sy, 0, H)
        alive = ((boxes[..., 2] - boxes[..., 0]) > 1) & ((boxes[..., 3] - boxes[..., 1]) > 1)
        if not alive.any():
            return self._unchanged(data_dict)
        metas['is_shifted'], metas['shift_params'] = True, (sx, sy)
        keep = alive & label['mask'].astype(bool)
        label['gt_bboxes'] = boxes
        centers = label['centers2d'].copy()
        centers[..., 0] += sx; centers[..., 1] += sy
        label['centers2d'] = centers
        kpts = label['gt_kpts_2d'].copy()
        kpts[..., 0::2] += sx; kpts[..., 1::2] += sy
        label['gt_kpts_2d'] = kpts
        keep_objects(label, keep)
        calib = data_dict['calib']                       # the camera sees the same rays: move the principal point
        calib.P2[0, 2] += sx
        calib.P2[1, 2] += sy
        if hasattr(calib, '_refresh_intrinsics'):
            calib._refresh_intrinsics()
        if 'img_ops' in data_dict:
            data_dict['img_ops'].append(('shift', (sx, sy)))
            return data_dict
        img = data_dict['img']
        canvas = np.zeros_like(img)
        h, w = H - abs(sy), W - abs(sx)
        canvas[max(0, sy):max(0, sy) + h, max(0, sx):max(0, sx) + w] = img[max(0, -sy):max(0, -sy) + h, max(0, -sx):max(0, -sx) + w]
        data_dict['img'] = canvas
        return data_dict


class RandomHorizontalFlip(BaseTransform):
    """mirror image, principal point / baseline term of P2, boxes, centres, and the left/right pairing of the 8 corners"""

    def __init__(self, prob: float = 0.50, rng=None):
        super().__init__(True, True, True, True)
        assert 0.0 <= prob <= 1.0
        self.prob = prob
        self._draw = _Draw(rng)

    def __call__(self, data_dict: Dict[str, Any]) -> Dict[str, Any]:
        metas = data_dict['img_metas']
        if self._draw.unit() >= self.prob:
            metas['is_flipped'] = False
            return data_dict
        img = data_dict['img']
        w = img.shape[1]
        if 'img_ops' in data_dict:
            data_dict['img_ops'].append(('flip', True))
        else:
            data_dict['img'] = img[:, ::-1, :]
        metas['is_flipped'] = True
        calib = data_dict['calib']
        calib.P2[0, 2] = w - calib.P2[0, 2] - 1
        calib.P2[0, 3] = -calib.P2[0, 3]
        if hasattr(calib, '_refresh_intrinsics'):
            calib._refresh_intrinsics()
        label = data_dict['label']
        on = label['mask'].astype(label['centers2d'].dtype)
        label['centers2d'][..., 0] = (w - label['centers2d'][..., 0] - 1) * on
        boxes = label['gt_bboxes']
        ref_w = metas['ori_shape'][1]                     # (the reference mirrors boxes about w, points about w - 1)
        flipped = boxes.copy()
        flipped[..., 0], flipped[..., 2] = ref_w - boxes[..., 2], ref_w - boxes[..., 0]
        label['gt_bboxes'] = flipped * on[:, None]
        b3 = label['gt_bboxes_3d']
        b3[..., 0] = -b3[..., 0]
        b3[..., -1] = -b3[..., -1] + np.pi
        label['gt_bboxes_3d'] = b3 * on[:, None]
        swap = [1, 0, 3, 2, 5, 4, 7, 6]                   # corner i <-> its mirror partner; the centre (8) stays
        kp = label['gt_kpts_2d'].copy()
        kp[..., 0::2] = (w - kp[..., 0::2] - 1) * on[:, None]
        kp = kp.reshape(kp.shape[0], -1, 2)
        kp[:, :8] = kp[:, swap]
        label['gt_kpts_2d'] = kp.reshape(kp.shape[0], -1)
        vm = label['gt_kpts_valid_mask'].copy()
        vm[:, :8] = vm[:, swap]
        label['gt_kpts_valid_mask'] = vm
        return data_dict


def _box_in_frame(frame: np.ndarray, box: np.ndarray):
    """'within' (box unchanged), 'out', or 'inters' with the clipped box"""
    inter = np.array([max(frame[0], box[0]), max(frame[1], box[1]), min(frame[2], box[2]), min(frame[3], box[3])])
    if np.allclose(inter, box):
        return 'within', box
    if inter[2] <= inter[0] or inter[3] <= inter[1]:
        return 'out', None
    return 'inters', inter


class _CropBase(BaseTransform):
    """blank everything outside a window (the image keeps its size and geometry, so the calibration is untouched);
    objects leave the labels when less than ``area_filter_thres`` of their 2D box remains"""

    def __init__(self, prob, hide_kpts_in_crop_area, area_filter_thres, rng):
        super().__init__(True, True, False, True)
        assert 0.0 <= prob <= 1.0
        assert 0.0 <= area_filter_thres < 1.0
        self.prob, self.hide_kpts_in_crop_area, self.area_filter_thres = prob, hide_kpts_in_crop_area, area_filter_thres
        self._draw = _Draw(rng)
        self._keep_original_if_empty = False

    def _window(self, ori_hw):
        raise NotImplementedError

    def __call__(self, data_dict: Dict[str, Any]) -> Dict[str, Any]:
        metas = data_dict['img_metas']
        if self._draw.unit() >= self.prob:
            metas['is_cropped'], metas['crop_coord'] = False, (0, 0, 0, 0)
            return data_dict
        ch, cw = self._window(metas['ori_shape'])
        H, W = metas['ori_shape']
        y0, x0 = self._draw.between(0, H - ch), self._draw.between(0, W - cw)
        frame = (x0, y0, x0 + cw, y0 + ch)
        metas['is_cropped'], metas['crop_coord'] = True, frame
        label = data_dict['label']
        boxes = label['gt_bboxes']
        was = label['mask'].astype(bool)
        now = np.zeros_like(was)
        for i in np.nonzero(was)[0]:
            kind, clipped = _box_in_frame(np.array(frame), boxes[i])
            if kind == 'within':
                now[i] = True
            elif kind == 'inters':
                ratio = ((clipped[2] - clipped[0]) * (clipped[3] - clipped[1])) / ((boxes[i][2] - boxes[i][0]) * (boxes[i][3] - boxes[i][1]))
                if ratio >= self.area_filter_thres:
                    now[i] = True
                    boxes[i] = clipped
        if self._keep_original_if_empty and not now.any():
            return data_dict                              # nothing would be left to learn from: hand the frame on as it is
        keep_objects(label, was & now)
        if self.hide_kpts_in_crop_area:                   # corners outside the window: "outside the image" (flag 1)
            kp = label['gt_kpts_2d'].reshape(len(was), 9, 2)
            inside = (kp[..., 0] >= frame[0]) & (kp[..., 0] <= frame[2]) & (kp[..., 1] >= frame[1]) & (kp[..., 1] <= frame[3])
            vm = label['gt_kpts_valid_mask']
            rows = label['mask'].astype(bool)
            vm[rows] = np.where(inside[rows], vm[rows], 1).astype(vm.dtype)
        if 'img_ops' in data_dict:
            data_dict['img_ops'].append(('window', tuple(int(v) for v in frame)))
            return data_dict
        img = data_dict['img']
        canvas = np.zeros_like(img)
        canvas[frame[1]:frame[3], frame[0]:frame[2], :] = img[frame[1]:frame[3], frame[0]:frame[2], :]
        data_dict['img'] = canvas
        return data_dict


class RandomCrop3D(_CropBase):
    def __init__(self, prob: float = 0.50, crop_size: Union[int, Tuple[int, int]] = (320, 960),
                 hide_kpts_in_crop_area: bool = False, area_filter_thres: float = 0.20, rng=None):
        super().__init__(prob, hide_kpts_in_crop_area, area_filter_thres, rng)
        self.crop_size = (crop_size, crop_size) if isinstance(crop_size, int) else tuple(crop_size)
        self._keep_original_if_empty = True

    def __call__(self, data_dict):
        shape = data_dict['img_metas']['ori_shape']
        assert self.crop_size[0] <= shape[0] and self.crop_size[1] <= shape[1], \
            "Crop size should be smaller than image size. (crop size: %s, image size: %s)" % (self.crop_size, shape)
        return super().__call__(data_dict)

    def _window(self, ori_hw):
        return self.crop_size


class RandomRangeCrop3D(_CropBase):
    def __init__(self, prob: float = 0.50, height_range: Union[int, Tuple[int, int]] = (256, 320), aspect_ratio: float = 3.0,
                 hide_kpts_in_crop_area: bool = True, area_filter_thres: float = 0.20, rng=None):
        super().__init__(prob, hide_kpts_in_crop_area, area_filter_thres, rng)
        self.height_range = (height_range, height_range) if isinstance(height_range, int) else tuple(height_range)
        self.width_range = (int(self.height_range[0] * aspect_ratio), int(self.height_range[1] * aspect_ratio))

    def _window(self, ori_hw):
        return self._draw.between(*self.height_range), self._draw.between(*self.width_range)


# ------------------------------------------------------------------------------------------------------ misc
class Resize3D(BaseTransform):
    def __init__(self, target_hw: Union[int, Tuple[int, int]] = None):
        super().__init__(True, True, True, True)
        self.target_hw = (target_hw, target_hw) if isinstance(target_hw, int) else target_hw

    def __call__(self, data_dict: Dict[str, Any]) -> Dict[str, Any]:
        if self.target_hw is None:
            return data_dict
        if data_dict.get('img_ops'):
            raise NotImplementedError("Resize3D behind a deferred image operation: the device kernel maps whole pixels only")
        import torch
        import torch.nn.functional as F
        img = data_dict['img']
        ori_hw = img.shape[:2]
        t = torch.from_numpy(np.array(img, copy=True)).permute(2, 0, 1)[None].float()
        out = F.interpolate(t, size=tuple(self.target_hw), mode='bilinear', align_corners=False)[0].permute(1, 2, 0).numpy()
        data_dict['img'] = np.clip(np.rint(out), 0, 255).astype(np.uint8) if img.dtype == np.uint8 else out.astype(img.dtype)
        scale_hw = np.array(self.target_hw) / np.array(ori_hw)
        data_dict['img_metas']['scale_hw'] = scale_hw
        data_dict['img_metas']['ori_shape'] = tuple(self.target_hw)
        data_dict['calib'].rescale(*scale_hw[::-1])
        if 'label' in data_dict:
            sx, sy = scale_hw[1], scale_hw[0]
            lab = data_dict['label']
            lab['gt_bboxes'] = lab['gt_bboxes'] * np.array([sx, sy, sx, sy], dtype=lab['gt_bboxes'].dtype)
            lab['centers2d'] = lab['centers2d'] * np.array([sx, sy], dtype=lab['centers2d'].dtype)
            lab['gt_kpts_2d'] = lab['gt_kpts_2d'] * np.tile(np.array([sx, sy], dtype=lab['gt_kpts_2d'].dtype), 9)
        return data_dict


class Convert_3D_to_4D(BaseTransform):
    """a single transformed sample -> a batch of one (image tensors gain a batch axis, metas / calib become lists)"""

    def __init__(self):
        super().__init__(True, True, False, False)

    def __call__(self, data_dict: Dict[str, Any]) -> Dict[str, Any]:
        import torch
        for k, v in list(data_dict.items()):
            if isinstance(v, torch.Tensor) and v.dim() == 3:
                data_dict[k] = v.unsqueeze(0)
        data_dict['img_metas'] = {k: [v] for k, v in data_dict['img_metas'].items()}
        data_dict['calib'] = [data_dict['calib']]
        return data_dict
