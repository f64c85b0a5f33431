"""Decoded boxes -> KITTI annotation dicts (the step right after decode; host-side numpy).

Same outputs as reference utils/kitti_convert_utils.py:16-249 (``convert_to_kitti_2d`` /
``convert_to_kitti_3d``), written vectorised: 3D boxes (x, y_bottom, z, l, h, w, rot_y in camera
coordinates) are turned into their 8 corners (utils/geometry_ops.py:7-45,126-163), projected with
P2, the enclosing 2D box is clipped to the image and boxes entirely outside the image are dropped.
"""
from typing import Any, Dict, List

import numpy as np
import torch

CLASSES = ('Pedestrian', 'Cyclist', 'Car')
_KEYS = ('name', 'truncated', 'occluded', 'alpha', 'bbox', 'dimensions', 'location', 'rotation_y', 'score')


def _empty_anno():
    return {'name': np.array([]), 'truncated': np.array([]), 'occluded': np.array([]), 'alpha': np.array([]),
            'bbox': np.zeros([0, 4]), 'dimensions': np.zeros([0, 3]), 'location': np.zeros([0, 3]),
            'rotation_y': np.array([]), 'score': np.array([])}


def _inv_scale(img_metas):
    s = img_metas['scale_hw'][0] if img_metas.get('scale_hw') else (1., 1.)
    return np.reciprocal(np.array([s[1], s[0], s[1], s[0]], dtype=np.float64))


def _np(x):
    return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


def project_boxes_3d(boxes: np.ndarray, P2: np.ndarray) -> np.ndarray:
    """(N,7) camera boxes with bottom-centre origin -> (N,4) enclosing image boxes x1,y1,x2,y2."""
    boxes = np.asarray(boxes, np.float32)
    unit = np.array([[sx, sy, sz] for sx in (-0.5, 0.5) for sy in (-1.0, 0.0) for sz in (-0.5, 0.5)], np.float32)
    c = boxes[:, None, 3:6] * unit[None]                                     # (N,8,3) l,h,w extents
    sin, cos = np.sin(boxes[:, 6])[:, None], np.cos(boxes[:, 6])[:, None]
    x = c[..., 0] * cos + c[..., 2] * sin
    z = -c[..., 0] * sin + c[..., 2] * cos
    pts = np.stack([x, c[..., 1], z], -1) + boxes[:, None, :3]              # float32, like the reference
    P = np.eye(4, dtype=np.float32)
    P[:3, :4] = np.asarray(P2, np.float32).reshape(3, 4)
    p = np.concatenate([pts.astype(np.float64), np.ones(pts.shape[:-1] + (1,))], -1) @ P.T
    uv = p[..., :2] / p[..., 2:3]
    return np.concatenate([uv.min(1), uv.max(1)], 1)


def convert_to_kitti_3d(results_3d: List[Dict[str, torch.Tensor]], img_metas: Dict[str, Any], calibs) -> List[Dict[str, Any]]:
    inv = _inv_scale(img_metas)
    out = []
    for i, r in enumerate(results_3d):
        boxes, scores, labels = _np(r['boxes_3d']), _np(r['scores_3d']), _np(r['labels_3d'])
        h, w = img_metas['ori_shape'][i]
        anno = _empty_anno()
        if len(boxes):
            b2 = project_boxes_3d(boxes, calibs[i].P2)
            ok = (b2[:, 0] < w) & (b2[:, 1] < h) & (b2[:, 2] > 0) & (b2[:, 3] > 0)
            if ok.any():
                boxes, scores, labels, b2 = boxes[ok], scores[ok], labels[ok], b2[ok]
                b2[:, 2:] = np.minimum(b2[:, 2:], [w, h])
                b2[:, :2] = np.maximum(b2[:, :2], [0, 0])
                anno = {'name': np.array([CLASSES[int(l)] for l in labels]), 'truncated': np.zeros(len(boxes)),
                        'occluded': np.zeros(len(boxes), dtype=np.int64),
                        'alpha': -np.arctan2(boxes[:, 0], boxes[:, 2]) + boxes[:, 6], 'bbox': b2 * inv,
                        'dimensions': boxes[:, 3:6], 'location': boxes[:, :3], 'rotation_y': boxes[:, 6], 'score': scores}
        anno['sample_idx'] = np.array([img_metas['sample_idx'][i]] * len(anno['score']), dtype=np.int64)
        out.append(anno)
    return out


def convert_to_kitti_2d(results_2d: List[List[np.ndarray]], img_metas: Dict[str, Any]) -> List[Dict[str, Any]]:
    assert len(results_2d[0]) == len(CLASSES)
    inv = _inv_scale(img_metas)
    out = []
    for i, per_class in enumerate(results_2d):
        n = sum(b.shape[0] for b in per_class)
        anno = _empty_anno()
        if n:
            cls = np.concatenate([np.full(b.shape[0], c) for c, b in enumerate(per_class)])
            allb = np.concatenate([np.asarray(b).reshape(-1, 5) for b in per_class], 0)
            anno = {'name': np.array([CLASSES[c] for c in cls]), 'truncated': np.zeros(n), 'occluded': np.zeros(n, dtype=np.int64),
                    'alpha': np.full(n, -10), 'bbox': allb[:, :4] * inv, 'dimensions': np.zeros((n, 3), np.float32),
                    'location': np.full((n, 3), -1000.0, np.float32), 'rotation_y': np.zeros(n), 'score': allb[:, 4]}
        anno['sample_idx'] = np.array([img_metas['sample_idx'][i]] * n, dtype=np.int64)
        out.append(anno)
    return out
