"""KITTI average precision (AP40 / AP11, 2D box, bird's-eye view, 3D, orientation) for the MonoCon engine.

Mirror of the reference's engine/kitti_eval/eval.py: same entry points, same annotation dicts in, same result string
and result dict out -- ``kitti_eval(gt_annos, dt_annos, current_classes, eval_types)`` (eval.py:666-812) is what
``MonoConDataset.evaluate`` calls.  Where the reference JIT-compiles python loops with numba and runs one numba.cuda
kernel, this package calls native code behind the C-ABI (include/monocon_hip.h, csrc/kitti_eval.hip):

    rotated BEV IoU, 3D IoU          HIP kernels     mc_rotate_iou_eval, mc_box3d_overlap   (need a HIP device)
    2D box overlap                    C++ (host)      mc_kitti_image_overlap
    matching / tp-fp-fn counting      C++ (host)      mc_kitti_statistics_part

The orchestration (class / difficulty / threshold loops, ignore rules, recall sampling) is numpy.  There is no python
fallback for the native parts: a missing library raises.
"""
import ctypes as C

import numpy as np

from hipmonocon import lib as _lib

from .rotate_iou import rotate_iou_gpu_eval, box3d_overlap_gpu

N_SAMPLE_PTS = 41
CLASS_NAMES = ('car', 'pedestrian', 'cyclist')
MIN_HEIGHT = (40, 25, 25)
MAX_OCCLUSION = (0, 1, 2)
MAX_TRUNCATION = (0.15, 0.3, 0.5)
_DP, _LLP = C.POINTER(C.c_double), C.POINTER(C.c_longlong)


def _f64(a, cols=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a.reshape(-1, cols) if cols else a


def _dp(a):
    return a.ctypes.data_as(_DP) if a.size else _DP()


def _llp(a):
    return a.ctypes.data_as(_LLP) if a.size else _LLP()


# ------------------------------------------------------------------------------------------------------ small pieces
def get_thresholds(scores, num_gt, num_sample_pts=N_SAMPLE_PTS):
    """Scores at which recall crosses the next of ``num_sample_pts`` equally spaced levels (eval.py:14-32)."""
    scores = np.sort(np.asarray(scores, dtype=np.float64))[::-1]
    n = len(scores)
    picked, level = [], 0
    step = 1 / (num_sample_pts - 1.0)
    for i in range(n):
        here = (i + 1) / num_gt
        nxt = (i + 2) / num_gt if i < n - 1 else here
        if i < n - 1 and (nxt - level) < (level - here):
            continue
        picked.append(scores[i])
        level += step
    return picked


def clean_data(gt_anno, dt_anno, current_class, difficulty):
    """-> num_valid_gt, ignored_gt, ignored_dt, dc_bboxes (eval.py:35-87).  Flags: 0 evaluated, 1 ignored (neighbouring
    class, or harder than this difficulty; detections: too small), -1 other class."""
    cls = CLASS_NAMES[current_class]
    gnames = np.array([str(n).lower() for n in gt_anno['name']], dtype=object)
    gbox = _f64(gt_anno['bbox'], 4)
    same = gnames == cls
    near = np.zeros(len(gnames), dtype=bool)
    if cls == 'pedestrian':
        near = gnames == 'person_sitting'
    elif cls == 'car':
        near = gnames == 'van'
    hard = ((np.asarray(gt_anno['occluded'], dtype=np.float64).reshape(-1) > MAX_OCCLUSION[difficulty]) |
            (np.asarray(gt_anno['truncated'], dtype=np.float64).reshape(-1) > MAX_TRUNCATION[difficulty]) |
            ((gbox[:, 3] - gbox[:, 1]) <= MIN_HEIGHT[difficulty])) if len(gnames) else np.zeros(0, dtype=bool)
    ignored_gt = np.full(len(gnames), -1, dtype=np.int64)
    ignored_gt[near | (same & hard)] = 1
    ignored_gt[same & ~hard] = 0
    dc = gbox[np.array([str(n) == 'DontCare' for n in gt_anno['name']], dtype=bool)] if len(gnames) else gbox

    dnames = np.array([str(n).lower() for n in dt_anno['name']], dtype=object)
    dbox = _f64(dt_anno['bbox'], 4)
    ignored_dt = np.full(len(dnames), -1, dtype=np.int64)
    if len(dnames):
        ignored_dt[dnames == cls] = 0
        ignored_dt[np.abs(dbox[:, 3] - dbox[:, 1]) < MIN_HEIGHT[difficulty]] = 1
    return int((ignored_gt == 0).sum()), ignored_gt, ignored_dt, dc


def image_box_overlap(boxes, query_boxes, criterion=-1):
    """axis-aligned (x1, y1, x2, y2) boxes -> (N,K) (eval.py:90-119); native host code"""
    boxes, query_boxes = _f64(boxes, 4), _f64(query_boxes, 4)
    out = np.zeros((len(boxes), len(query_boxes)), dtype=np.float64)
    if out.size:
        rc = _lib.load().mc_kitti_image_overlap(_dp(boxes), len(boxes), _dp(query_boxes), len(query_boxes), int(criterion), _dp(out))
        if rc != 0:
            raise _lib.MonoconHipError("mc_kitti_image_overlap failed (rc=%d)" % rc)
    return out


def bev_box_overlap(boxes, qboxes, criterion=-1):
    """(N,5) / (K,5) [x, z, l, w, ry] -> rotated IoU on the GPU (eval.py:122-125)"""
    return rotate_iou_gpu_eval(boxes, qboxes, criterion)


def d3_box_overlap(boxes, qboxes, criterion=-1):
    """(N,7) / (K,7) camera-frame [x, y, z, l, h, w, ry] -> 3D IoU on the GPU (eval.py:160-164)"""
    return box3d_overlap_gpu(boxes, qboxes, criterion)


def get_split_parts(num, num_part):
    """frames per part (eval.py:288-294)"""
    same, rest = num // num_part, num % num_part
    return [same] * num_part + ([rest] if rest else [])


# ------------------------------------------------------------------------------------------------------ overlaps by part
def _boxes_of(annos, metric):
    if metric == 0:
        return np.concatenate([_f64(a['bbox'], 4) for a in annos], 0)
    loc = np.concatenate([_f64(a['location'], 3) for a in annos], 0)
    dims = np.concatenate([_f64(a['dimensions'], 3) for a in annos], 0)
    rots = np.concatenate([_f64(a['rotation_y']).reshape(-1, 1) for a in annos], 0)
    if metric == 1:
        return np.concatenate([loc[:, [0, 2]], dims[:, [0, 2]], rots], axis=1)
    return np.concatenate([loc, dims, rots], axis=1)


def calculate_iou_partly(gt_annos, dt_annos, metric, num_parts=50):
    """Overlap matrices, one launch per part of consecutive frames (eval.py:347-422).  Returns (per-frame blocks, per-part
    matrices, #boxes per frame of the first argument, of the second).  NOTE: eval_class passes detections first, as the
    reference does (eval.py:483), so rows are detections and columns ground truth."""
    assert len(gt_annos) == len(dt_annos)
    n_first = np.array([len(a['name']) for a in gt_annos], dtype=np.int64)
    n_second = np.array([len(a['name']) for a in dt_annos], dtype=np.int64)
    if metric not in (0, 1, 2):
        raise ValueError('unknown metric')
    parted, blocks = [], []
    at = 0
    for num in get_split_parts(len(gt_annos), num_parts):
        first, second = gt_annos[at:at + num], dt_annos[at:at + num]
        a, b = _boxes_of(first, metric), _boxes_of(second, metric)
        if metric == 0:
            part = image_box_overlap(a, b)
        elif metric == 1:
            part = bev_box_overlap(a, b).astype(np.float64)
        else:
            part = d3_box_overlap(a, b).astype(np.float64)
        parted.append(part)
        r = c = 0
        for i in range(num):
            nr, nc = n_first[at + i], n_second[at + i]
            blocks.append(part[r:r + nr, c:c + nc])
            r += nr; c += nc
        at += num
    return blocks, parted, n_first, n_second


# ------------------------------------------------------------------------------------------------------ statistics
def _statistics_part(mode, overlaps, gt_nums, dt_nums, dc_nums, gt_datas, dt_datas, dontcares, ignored_gts, ignored_dets,
                     metric, min_overlap, thresholds=None, compute_aos=False, pr=None):
    """mc_kitti_statistics_part over the frames of one part.  mode 0 -> scores of the true positives; mode 1 -> pr += ..."""
    overlaps = _f64(overlaps)
    gt_nums, dt_nums, dc_nums = (np.ascontiguousarray(x, dtype=np.int64) for x in (gt_nums, dt_nums, dc_nums))
    gt_datas, dt_datas, dontcares = _f64(gt_datas, 5), _f64(dt_datas, 6), _f64(dontcares, 4)
    ignored_gts = np.ascontiguousarray(ignored_gts, dtype=np.int64)
    ignored_dets = np.ascontiguousarray(ignored_dets, dtype=np.int64)
    assert overlaps.size == int(gt_nums.sum()) * int(dt_nums.sum())
    thr = _f64(thresholds if thresholds is not None else [])
    scores = np.zeros(max(int(gt_nums.sum()), 1), dtype=np.float64)
    n_scores = C.c_longlong(0)
    if mode == 1:
        assert pr is not None and pr.dtype == np.float64 and pr.flags.c_contiguous and pr.shape == (len(thr), 4)
    elif pr is not None:
        assert pr.dtype == np.float64 and pr.flags.c_contiguous and pr.size >= 4
    rc = _lib.load().mc_kitti_statistics_part(
        int(mode), _dp(overlaps), len(gt_nums), _llp(gt_nums), _llp(dt_nums), _llp(dc_nums), _dp(gt_datas), _dp(dt_datas),
        _dp(dontcares), _llp(ignored_gts), _llp(ignored_dets), int(metric), float(min_overlap), _dp(thr), len(thr),
        int(bool(compute_aos)), _dp(pr) if pr is not None else _DP(), _dp(scores), C.byref(n_scores))
    if rc != 0:
        raise _lib.MonoconHipError("mc_kitti_statistics_part failed (rc=%d)" % rc)
    return scores[:n_scores.value]


def compute_statistics_jit(overlaps, gt_datas, dt_datas, ignored_gt, ignored_det, dc_bboxes, metric, min_overlap, thresh=0,
                           compute_fp=False, compute_aos=False):
    """One frame (eval.py:167-285) -> tp, fp, fn, similarity, scores of the true positives.  Native; kept under the
    reference's name for callers that analyse single frames."""
    overlaps = _f64(overlaps)
    one = lambda n: np.array([n], dtype=np.int64)     # noqa: E731
    args = (overlaps, one(len(gt_datas)), one(len(dt_datas)), one(len(dc_bboxes)), gt_datas, dt_datas, dc_bboxes,
            ignored_gt, ignored_det, metric, min_overlap)
    if not compute_fp:
        # (without false-positive accounting the reference reports tp and fn only.  fn comes from the native pass itself:
        #  a valid ground truth matched to a detection with ignored_det == 1 is neither tp nor fn, eval.py:228-247 --
        #  "valid ground truths minus tp" over-counted such frames)
        pr0 = np.zeros((1, 4))
        s = _statistics_part(0, *args, pr=pr0)
        return int(pr0[0, 0]), 0, int(pr0[0, 2]), 0, s
    pr = np.zeros((1, 4))
    _statistics_part(1, *args, thresholds=[thresh], compute_aos=compute_aos, pr=pr)
    return int(pr[0, 0]), int(pr[0, 1]), int(pr[0, 2]), pr[0, 3], np.zeros(0)


def fused_compute_statistics(overlaps, pr, gt_nums, dt_nums, dc_nums, gt_datas, dt_datas, dontcares, ignored_gts,
                             ignored_dets, metric, min_overlap, thresholds, compute_aos=False):
    """pr[t] += (tp, fp, fn, similarity) over the frames of a part for every score threshold (eval.py:297-344)"""
    _statistics_part(1, overlaps, gt_nums, dt_nums, dc_nums, gt_datas, dt_datas, dontcares, ignored_gts, ignored_dets, metric,
                     min_overlap, thresholds=thresholds, compute_aos=compute_aos, pr=pr)


def _prepare_data(gt_annos, dt_annos, current_class, difficulty):
    """per-frame arrays for one class / difficulty (eval.py:425-453)"""
    gt_datas, dt_datas, ig_gts, ig_dts, dcs, dc_nums = [], [], [], [], [], []
    total_valid = 0
    for g, d in zip(gt_annos, dt_annos):
        nvalid, ig, idt, dc = clean_data(g, d, current_class, difficulty)
        total_valid += nvalid
        ig_gts.append(ig); ig_dts.append(idt); dcs.append(dc); dc_nums.append(len(dc))
        gt_datas.append(np.concatenate([_f64(g['bbox'], 4), _f64(g['alpha']).reshape(-1, 1)], 1))
        dt_datas.append(np.concatenate([_f64(d['bbox'], 4), _f64(d['alpha']).reshape(-1, 1), _f64(d['score']).reshape(-1, 1)], 1))
    return gt_datas, dt_datas, ig_gts, ig_dts, dcs, np.array(dc_nums, dtype=np.int64), total_valid


def eval_class(gt_annos, dt_annos, current_classes, difficultys, metric, min_overlaps, compute_aos=False, num_parts=200):
    """precision / recall / orientation at 41 recall positions, [class, difficulty, overlap threshold, 41]
    (eval.py:456-574).  metric 0: 2D box, 1: bird's-eye view, 2: 3D; min_overlaps [num_overlap, metric, class]."""
    assert len(gt_annos) == len(dt_annos)
    n = len(gt_annos)
    num_parts = min(num_parts, n)
    parts = get_split_parts(n, num_parts) if n else []
    _, parted_overlaps, n_dt, n_gt = calculate_iou_partly(dt_annos, gt_annos, metric, num_parts) if n else ([], [], [], [])
    shape = [len(current_classes), len(difficultys), len(min_overlaps), N_SAMPLE_PTS]
    precision, recall, aos = np.zeros(shape), np.zeros(shape), np.zeros(shape)
    for m, cls in enumerate(current_classes):
        for l, difficulty in enumerate(difficultys):
            gt_datas, dt_datas, ig_gts, ig_dts, dcs, dc_nums, total_valid = _prepare_data(gt_annos, dt_annos, cls, difficulty)
            cat = []                    # the per-part concatenations do not depend on the overlap threshold
            at = 0
            for num in parts:
                sl = slice(at, at + num)
                cat.append((n_gt[sl], n_dt[sl], dc_nums[sl],
                            np.concatenate(gt_datas[sl], 0) if num else np.zeros((0, 5)),
                            np.concatenate(dt_datas[sl], 0) if num else np.zeros((0, 6)),
                            np.concatenate(dcs[sl], 0) if num else np.zeros((0, 4)),
                            np.concatenate(ig_gts[sl], 0) if num else np.zeros(0, np.int64),
                            np.concatenate(ig_dts[sl], 0) if num else np.zeros(0, np.int64)))
                at += num
            for k, min_overlap in enumerate(min_overlaps[:, metric, m]):
                scores = [_statistics_part(0, parted_overlaps[j], *cat[j], metric, min_overlap) for j in range(len(parts))]
                scores = np.concatenate(scores) if scores else np.zeros(0)
                thresholds = np.array(get_thresholds(scores, total_valid), dtype=np.float64)
                pr = np.zeros([len(thresholds), 4])
                for j in range(len(parts)):
                    fused_compute_statistics(parted_overlaps[j], pr, *cat[j], metric, min_overlap=min_overlap,
                                             thresholds=thresholds, compute_aos=compute_aos)
                nt = len(thresholds)
                with np.errstate(divide='ignore', invalid='ignore'):
                    recall[m, l, k, :nt] = pr[:, 0] / (pr[:, 0] + pr[:, 2])
                    precision[m, l, k, :nt] = pr[:, 0] / (pr[:, 0] + pr[:, 1])
                    if compute_aos:
                        aos[m, l, k, :nt] = pr[:, 3] / (pr[:, 0] + pr[:, 1])
                # monotone envelope from the right (np.max semantics: a nan to the right propagates, as in the reference)
                for arr in (precision, recall) + ((aos,) if compute_aos else ()):
                    for i in range(nt):
                        arr[m, l, k, i] = np.max(arr[m, l, k, i:])
    return {'recall': recall, 'precision': precision, 'orientation': aos}


def get_mAP11(prec):
    return prec[..., ::4].sum(axis=-1) / 11 * 100


def get_mAP40(prec):
    return prec[..., 1:].sum(axis=-1) / 40 * 100


def do_eval(gt_annos, dt_annos, current_classes, min_overlaps, eval_types=('bbox', 'bev', '3d')):
    """-> (mAP11 bbox, bev, 3d, aos, mAP40 bbox, bev, 3d, aos), None where not requested (eval.py:600-643)"""
    difficultys = [0, 1, 2]
    out11 = {'bbox': None, 'bev': None, '3d': None, 'aos': None}
    out40 = dict(out11)
    for key, metric in (('bbox', 0), ('bev', 1), ('3d', 2)):
        if key not in eval_types:
            continue
        with_aos = key == 'bbox' and 'aos' in eval_types
        ret = eval_class(gt_annos, dt_annos, current_classes, difficultys, metric, min_overlaps, compute_aos=with_aos)
        out11[key], out40[key] = get_mAP11(ret['precision']), get_mAP40(ret['precision'])
        if with_aos:
            out11['aos'], out40['aos'] = get_mAP11(ret['orientation']), get_mAP40(ret['orientation'])
    return (out11['bbox'], out11['bev'], out11['3d'], out11['aos'], out40['bbox'], out40['bev'], out40['3d'], out40['aos'])


CLASS_TO_NAME = {0: 'Car', 1: 'Pedestrian', 2: 'Cyclist', 3: 'Van', 4: 'Person_sitting'}
NAME_TO_CLASS = {v: k for k, v in CLASS_TO_NAME.items()}


def _class_ids(current_classes):
    if not isinstance(current_classes, (list, tuple)):
        current_classes = [current_classes]
    return [NAME_TO_CLASS[c] if isinstance(c, str) else c for c in current_classes]


def kitti_eval(gt_annos, dt_annos, current_classes, eval_types=('bbox', 'bev', '3d')):
    """KITTI evaluation -> (result string, result dict) (eval.py:666-812).  AP40 at the strict (0.7 / 0.5 / 0.5) and
    loose (0.5 / 0.25 / 0.25) overlap thresholds for every class and difficulty, plus the mean over classes."""
    eval_types = list(eval_types)
    assert len(eval_types) > 0, 'must contain at least one evaluation type'
    if 'aos' in eval_types:
        assert 'bbox' in eval_types, 'must evaluate bbox when evaluating aos'
    strict = np.array([[0.7, 0.5, 0.5, 0.7, 0.5]] * 3)
    loose = np.array([[0.7, 0.5, 0.5, 0.7, 0.5], [0.5, 0.25, 0.25, 0.5, 0.25], [0.5, 0.25, 0.25, 0.5, 0.25]])
    current_classes = _class_ids(current_classes)
    min_overlaps = np.stack([strict, loose], axis=0)[:, :, current_classes]            # [2, metric, class]
    # orientation similarity only when both sides carry observation angles (-10 = absent)
    pred_alpha = any((np.asarray(a['alpha']) != -10).any() for a in dt_annos)
    valid_alpha_gt = any(len(np.asarray(a['alpha']).reshape(-1)) and np.asarray(a['alpha']).reshape(-1)[0] != -10 for a in gt_annos)
    compute_aos = bool(pred_alpha and valid_alpha_gt)
    if compute_aos and 'aos' not in eval_types:
        eval_types.append('aos')

    _, _, _, _, ap_bbox, ap_bev, ap_3d, ap_aos = do_eval(gt_annos, dt_annos, current_classes, min_overlaps, eval_types)

    ret_dict = {}
    difficulty = ['easy', 'moderate', 'hard']
    result = '\n----------- Eval Results ------------\n'
    for j, cls in enumerate(current_classes):
        name = CLASS_TO_NAME[cls]
        for i in range(min_overlaps.shape[0]):
            result += '{} AP40@{:.2f}, {:.2f}, {:.2f}:\n'.format(name, *min_overlaps[i, :, j])
            if ap_bbox is not None:
                result += 'bbox AP40:{:.4f}, {:.4f}, {:.4f}\n'.format(*ap_bbox[j, :, i])
            if ap_bev is not None:
                result += 'bev  AP40:{:.4f}, {:.4f}, {:.4f}\n'.format(*ap_bev[j, :, i])
            if ap_3d is not None:
                result += '3d   AP40:{:.4f}, {:.4f}, {:.4f}\n'.format(*ap_3d[j, :, i])
            if compute_aos and ap_aos is not None:
                result += 'aos  AP40:{:.2f}, {:.2f}, {:.2f}\n'.format(*ap_aos[j, :, i])
            for idx in range(3):
                postfix = '%s_%s' % (difficulty[idx], 'strict' if i == 0 else 'loose')
                prefix = 'KITTI/%s' % name
                if ap_3d is not None:
                    ret_dict['%s_3D_AP40_%s' % (prefix, postfix)] = ap_3d[j, idx, i]
                if ap_bev is not None:
                    ret_dict['%s_BEV_AP40_%s' % (prefix, postfix)] = ap_bev[j, idx, i]
                if ap_bbox is not None:
                    ret_dict['%s_2D_AP40_%s' % (prefix, postfix)] = ap_bbox[j, idx, i]
    if len(current_classes) > 1:
        result += '\nOverall AP40@{}, {}, {}:\n'.format(*difficulty)
        if ap_bbox is not None:
            ap_bbox = ap_bbox.mean(axis=0)
            result += 'bbox AP40:{:.4f}, {:.4f}, {:.4f}\n'.format(*ap_bbox[:, 0])
        if ap_bev is not None:
            ap_bev = ap_bev.mean(axis=0)
            result += 'bev  AP40:{:.4f}, {:.4f}, {:.4f}\n'.format(*ap_bev[:, 0])
        if ap_3d is not None:
            ap_3d = ap_3d.mean(axis=0)
            result += '3d   AP40:{:.4f}, {:.4f}, {:.4f}\n'.format(*ap_3d[:, 0])
        if compute_aos and ap_aos is not None:
            ap_aos = ap_aos.mean(axis=0)
            result += 'aos  AP40:{:.2f}, {:.2f}, {:.2f}\n'.format(*ap_aos[:, 0])
        for idx in range(3):
            if ap_3d is not None:
                ret_dict['KITTI/Overall_3D_AP40_%s' % difficulty[idx]] = ap_3d[idx, 0]
            if ap_bev is not None:
                ret_dict['KITTI/Overall_BEV_AP40_%s' % difficulty[idx]] = ap_bev[idx, 0]
            if ap_bbox is not None:
                ret_dict['KITTI/Overall_2D_AP40_%s' % difficulty[idx]] = ap_bbox[idx, 0]
    result += '-------------------------------------'
    return result, ret_dict


def do_coco_style_eval(gt_annos, dt_annos, current_classes, overlap_ranges, compute_aos):
    """AP11 averaged over ten overlap thresholds per (metric, class) (eval.py:646-663)"""
    min_overlaps = np.zeros([10, *overlap_ranges.shape[1:]])
    for i in range(overlap_ranges.shape[1]):
        for j in range(overlap_ranges.shape[2]):
            lo, hi, num = overlap_ranges[:, i, j]
            min_overlaps[:, i, j] = np.linspace(lo, hi, int(num))
    types = ['bbox', 'bev', '3d'] + (['aos'] if compute_aos else [])
    bbox, bev, d3, aos = do_eval(gt_annos, dt_annos, current_classes, min_overlaps, types)[:4]
    return bbox.mean(-1), bev.mean(-1), d3.mean(-1), (aos.mean(-1) if aos is not None else None)


def kitti_eval_coco_style(gt_annos, dt_annos, current_classes):
    """COCO-style summary string (mean AP over 0.5:0.05:0.95 for Car / Van, 0.25:0.05:0.7 otherwise)"""
    ranges = {0: [0.5, 0.95, 10], 1: [0.25, 0.7, 10], 2: [0.25, 0.7, 10], 3: [0.5, 0.95, 10], 4: [0.25, 0.7, 10]}
    current_classes = _class_ids(current_classes)
    overlap_ranges = np.zeros([3, 3, len(current_classes)])
    for i, c in enumerate(current_classes):
        overlap_ranges[:, :, i] = np.array(ranges[c])[:, np.newaxis]
    compute_aos = False
    for anno in dt_annos:
        a = np.asarray(anno['alpha']).reshape(-1)
        if a.shape[0] != 0:
            compute_aos = bool(a[0] != -10)
            break
    bbox, bev, d3, aos = do_coco_style_eval(gt_annos, dt_annos, current_classes, overlap_ranges, compute_aos)
    result = ''
    for j, c in enumerate(current_classes):
        lo, hi, num = ranges[c]
        result += '%s coco AP@%.2f:%.2f:%.2f:\n' % (CLASS_TO_NAME[c], lo, (hi - lo) / (num - 1), hi)
        result += 'bbox AP:%.2f, %.2f, %.2f\n' % tuple(bbox[j, :3])
        result += 'bev  AP:%.2f, %.2f, %.2f\n' % tuple(bev[j, :3])
        result += '3d   AP:%.2f, %.2f, %.2f\n' % tuple(d3[j, :3])
        if compute_aos:
            result += 'aos  AP:%.2f, %.2f, %.2f\n' % tuple(aos[j, :3])
    return result
