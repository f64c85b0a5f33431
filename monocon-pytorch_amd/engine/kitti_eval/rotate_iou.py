"""Rotated-box overlaps on the GPU for the KITTI evaluator.

Mirror of the reference's engine/kitti_eval/rotate_iou.py (rotate_iou_gpu_eval, :337-378): numpy in, numpy out, the
pairwise work on the device -- here through the HIP kernel behind mc_rotate_iou_eval / mc_box3d_overlap
(csrc/kitti_eval.hip) instead of a numba.cuda JIT kernel.  There is no CPU path: without a HIP device these raise.
"""
import numpy as np
import torch

_ENGINES = {}


def _engine(device_id=0):
    from hipmonocon.engine import Engine
    eng = _ENGINES.get(device_id)
    if eng is None:
        eng = _ENGINES[device_id] = Engine(device_id)
    return eng


def rotate_iou_gpu_eval(boxes, query_boxes, criterion=-1, device_id=0):
    """boxes (N,5), query_boxes (K,5): [center x, center y, dim x, dim y, angle (clockwise positive)].
    criterion -1: IoU; 0: intersection / area(query box); 1: intersection / area(box); 2: the intersection area.
    Returns (N,K) in the dtype of ``boxes`` (computed in float32 like the reference, rotate_iou.py:356-378)."""
    boxes = np.asarray(boxes)
    query_boxes = np.asarray(query_boxes)
    N, K = boxes.shape[0], query_boxes.shape[0]
    if N == 0 or K == 0:
        return np.zeros((N, K), dtype=np.float32).astype(boxes.dtype)
    eng = _engine(device_id)
    b = torch.from_numpy(np.ascontiguousarray(boxes, dtype=np.float32)).to(eng.device)
    q = torch.from_numpy(np.ascontiguousarray(query_boxes, dtype=np.float32)).to(eng.device)
    return eng.rotate_iou(b, q, criterion).cpu().numpy().astype(boxes.dtype)


def box3d_overlap_gpu(boxes, query_boxes, criterion=-1, device_id=0):
    """camera-frame boxes (N,7), (K,7) = [x, y, z, l, h, w, ry] -> (N,K) float64 3D overlap in one launch (what the
    reference computes in two steps, engine/kitti_eval/eval.py:128-164)."""
    boxes = np.ascontiguousarray(boxes, dtype=np.float64)
    query_boxes = np.ascontiguousarray(query_boxes, dtype=np.float64)
    N, K = boxes.shape[0], query_boxes.shape[0]
    if N == 0 or K == 0:
        return np.zeros((N, K), dtype=np.float64)
    eng = _engine(device_id)
    return eng.box3d_overlap(torch.from_numpy(boxes).to(eng.device), torch.from_numpy(query_boxes).to(eng.device),
                             criterion).cpu().numpy()
