"""CPU restatement of the reference's KITTI AP evaluator -- TEST INFRASTRUCTURE ONLY (imported by tests/, never by the
product package).

Restates, by reading:
  * engine/kitti_eval/rotate_iou.py  (the numba.cuda rotated-IoU kernel: :19-32 triangle fan area, :35-72 vertex sort,
    :75-115 segment intersection, :157-173 point-in-quadrilateral, :176-197 polygon vertices, :200-223 corners,
    :226-249 intersection, :252-277 criterion, :280-378 kernel + host wrapper)
  * engine/kitti_eval/eval.py  (:14-32 get_thresholds, :35-87 clean_data, :90-119 image_box_overlap, :128-164 3D overlap,
    :167-285 compute_statistics_jit, :297-344 fused statistics, :347-422 overlaps by parts, :425-453 _prepare_data,
    :456-574 eval_class, :584-588 AP40, :600-643 do_eval, :666-812 kitti_eval)

PINNED (round 6) for everything but the rotated-overlap kernel: tests/golden/make_f4_golden.py imports the reference's
engine/kitti_eval/eval.py under identity ``numba.jit`` decorators (numba is absent from this image; the host loops then run
as the Python they are written in, no placeholder does any work -- meta_f4.json) and records what ITS functions return:
image_box_overlap, clean_data, compute_statistics_jit (both passes), get_thresholds, d3_box_overlap_kernel, eval_class for
the three metrics, get_mAP40, kitti_eval's dict.  tests/test_f4_reference_golden.py holds this file (and the product's
native matching) to those goldens.  PARITY UNPINNED, still: ``rotate_iou`` below -- the reference's version is a float32
numba.cuda kernel (rotate_iou.py:280-379) that cannot execute here; for the BEV / 3D goldens the reference's evaluator
was handed THIS function's overlaps.  What holds it (tests/test_kitti_eval.py): closed-form answers (identical / disjoint
/ axis-aligned / 45-degree boxes) and an independent float64 polygon-clipping implementation that shares no code with it.

Straight loops, numpy float32 scalars where the reference kernel computes in float32; meant for tens of boxes.
"""
import math

import numpy as np

F = np.float32


# ------------------------------------------------------------------------------------------------ rotated overlap
def rbox_corners(box):
    """(cx, cy, dx, dy, angle) float32 -> 8 float32 (x0, y0, ... x3, y3)   [rotate_iou.py:200-223]"""
    cx, cy, dx, dy, ang = (F(v) for v in box)
    c, s = F(np.cos(ang)), F(np.sin(ang))
    hx, hy = F(dx / F(2)), F(dy / F(2))
    lx = (-hx, -hx, hx, hx)
    ly = (-hy, hy, hy, -hy)
    out = np.zeros(8, dtype=F)
    for i in range(4):
        out[2 * i] = F(F(c * lx[i]) + F(s * ly[i])) + cx
        out[2 * i + 1] = F(F(-s * lx[i]) + F(c * ly[i])) + cy
    return out


def _inside(x, y, q):
    """[rotate_iou.py:157-173]"""
    ab0, ab1 = F(q[2] - q[0]), F(q[3] - q[1])
    ad0, ad1 = F(q[6] - q[0]), F(q[7] - q[1])
    ap0, ap1 = F(x - q[0]), F(y - q[1])
    abab = F(F(ab0 * ab0) + F(ab1 * ab1))
    abap = F(F(ab0 * ap0) + F(ab1 * ap1))
    adad = F(F(ad0 * ad0) + F(ad1 * ad1))
    adap = F(F(ad0 * ap0) + F(ad1 * ap1))
    return abab >= abap and abap >= 0 and adad >= adap and adap >= 0


def _cross(p, q, i, j):
    """edge i of p x edge j of q -> (x, y) or None   [rotate_iou.py:75-115]"""
    a = (p[2 * i], p[2 * i + 1]); b = (p[2 * ((i + 1) % 4)], p[2 * ((i + 1) % 4) + 1])
    c = (q[2 * j], q[2 * j + 1]); d = (q[2 * ((j + 1) % 4)], q[2 * ((j + 1) % 4) + 1])
    ba0, ba1 = F(b[0] - a[0]), F(b[1] - a[1])
    da0, ca0 = F(d[0] - a[0]), F(c[0] - a[0])
    da1, ca1 = F(d[1] - a[1]), F(c[1] - a[1])
    acd = F(da1 * ca0) > F(ca1 * da0)
    bcd = F(F(d[1] - b[1]) * F(c[0] - b[0])) > F(F(c[1] - b[1]) * F(d[0] - b[0]))
    if acd == bcd:
        return None
    abc = F(ca1 * ba0) > F(ba1 * ca0)
    abd = F(da1 * ba0) > F(ba1 * da0)
    if abc == abd:
        return None
    dc0, dc1 = F(d[0] - c[0]), F(d[1] - c[1])
    abba = F(F(a[0] * b[1]) - F(b[0] * a[1]))
    cddc = F(F(c[0] * d[1]) - F(d[0] * c[1]))
    dh = F(F(ba1 * dc0) - F(ba0 * dc1))
    with np.errstate(divide="ignore", invalid="ignore"):
        return (F(F(F(abba * dc0) - F(ba0 * cddc)) / dh), F(F(F(abba * dc1) - F(ba1 * cddc)) / dh))


def intersection_area(p, q):
    """area (python float, i.e. double) of the intersection of two quadrilaterals given as corner arrays
    [rotate_iou.py:176-197 vertices, :35-72 order, :19-32 area]"""
    pts = []
    for i in range(4):
        if _inside(p[2 * i], p[2 * i + 1], q):
            pts.append([p[2 * i], p[2 * i + 1]])
        if _inside(q[2 * i], q[2 * i + 1], p):
            pts.append([q[2 * i], q[2 * i + 1]])
    for i in range(4):
        for j in range(4):
            hit = _cross(p, q, i, j)
            if hit is not None:
                pts.append([hit[0], hit[1]])
    # (at most 8 corners + 8 crossings.  The reference's vertex buffer holds only 8 points: nearly coincident boxes, which
    #  produce more, overrun it there -- undefined behaviour in the reference; all candidates are kept here.)
    n = len(pts)
    if n == 0:
        return 0.0
    c0, c1 = F(0), F(0)
    for x, y in pts:
        c0 = F(c0 + x); c1 = F(c1 + y)
    c0 = F(float(c0) / n); c1 = F(float(c1) / n)
    keys = []
    with np.errstate(divide="ignore", invalid="ignore"):
        for x, y in pts:
            v0, v1 = F(x - c0), F(y - c1)
            dd = F(np.sqrt(F(F(v0 * v0) + F(v1 * v1))))
            v0, v1 = F(v0 / dd), F(v1 / dd)
            if v1 < 0:
                v0 = F(F(-2) - v0)
            keys.append(v0)
    for i in range(1, n):               # insertion sort, keys and points together
        if keys[i - 1] > keys[i]:
            key, pt = keys[i], pts[i]
            j = i
            while j > 0 and keys[j - 1] > key:
                keys[j] = keys[j - 1]; pts[j] = pts[j - 1]
                j -= 1
            keys[j] = key; pts[j] = pt
    area = 0.0
    a = pts[0]
    for i in range(n - 2):
        b, c = pts[i + 1], pts[i + 2]
        cr = F(F(F(a[0] - c[0]) * F(b[1] - c[1])) - F(F(a[1] - c[1]) * F(b[0] - c[0])))
        area += abs(float(cr) / 2.0)
    return area


def rotate_iou(boxes, query_boxes, criterion=-1):
    """(N,5), (K,5) -> float32 (N,K)   [rotate_iou.py:252-277 with rbox1 = the QUERY box, :280-378]"""
    boxes = np.asarray(boxes, dtype=F).reshape(-1, 5)
    query_boxes = np.asarray(query_boxes, dtype=F).reshape(-1, 5)
    out = np.zeros((len(boxes), len(query_boxes)), dtype=F)
    cb = [rbox_corners(b) for b in boxes]
    cq = [rbox_corners(b) for b in query_boxes]
    for n in range(len(boxes)):
        for k in range(len(query_boxes)):
            ai = intersection_area(cq[k], cb[n])
            a1 = F(query_boxes[k, 2] * query_boxes[k, 3])
            a2 = F(boxes[n, 2] * boxes[n, 3])
            with np.errstate(divide="ignore", invalid="ignore"):
                if criterion == -1:
                    r = np.float64(ai) / (np.float64(F(a1 + a2)) - ai)
                elif criterion == 0:
                    r = np.float64(ai) / np.float64(a1)
                elif criterion == 1:
                    r = np.float64(ai) / np.float64(a2)
                else:
                    r = ai
            out[n, k] = F(r)
    return out


def box3d_overlap(boxes, query_boxes, criterion=-1):
    """camera-frame (x, y, z, l, h, w, ry) float64 boxes -> float64 (N,K)   [eval.py:128-164]"""
    boxes = np.asarray(boxes, dtype=np.float64).reshape(-1, 7)
    query_boxes = np.asarray(query_boxes, dtype=np.float64).reshape(-1, 7)
    bev = rotate_iou(boxes[:, [0, 2, 3, 5, 6]], query_boxes[:, [0, 2, 3, 5, 6]], 2).astype(np.float64)
    out = np.zeros_like(bev)
    for i in range(len(boxes)):
        for j in range(len(query_boxes)):
            if bev[i, j] > 0:
                iw = min(boxes[i, 1], query_boxes[j, 1]) - max(boxes[i, 1] - boxes[i, 4], query_boxes[j, 1] - query_boxes[j, 4])
                if iw > 0:
                    v1 = boxes[i, 3] * boxes[i, 4] * boxes[i, 5]
                    v2 = query_boxes[j, 3] * query_boxes[j, 4] * query_boxes[j, 5]
                    inc = iw * bev[i, j]
                    ua = {-1: v1 + v2 - inc, 0: v1, 1: v2}.get(criterion, inc)
                    out[i, j] = inc / ua
    return out


def image_overlap(boxes, query_boxes, criterion=-1):
    """axis-aligned (x1, y1, x2, y2)   [eval.py:90-119]"""
    boxes = np.asarray(boxes, dtype=np.float64).reshape(-1, 4)
    query_boxes = np.asarray(query_boxes, dtype=np.float64).reshape(-1, 4)
    out = np.zeros((len(boxes), len(query_boxes)))
    for k, qb in enumerate(query_boxes):
        qa = (qb[2] - qb[0]) * (qb[3] - qb[1])
        for n, b in enumerate(boxes):
            iw = min(b[2], qb[2]) - max(b[0], qb[0])
            ih = min(b[3], qb[3]) - max(b[1], qb[1])
            if iw > 0 and ih > 0:
                ba = (b[2] - b[0]) * (b[3] - b[1])
                ua = {-1: ba + qa - iw * ih, 0: ba, 1: qa}.get(criterion, 1.0)
                out[n, k] = iw * ih / ua
    return out


# ------------------------------------------------------------------------------------------------ matching
CLASS_NAMES = ("car", "pedestrian", "cyclist")
MIN_HEIGHT = (40, 25, 25)
MAX_OCCLUSION = (0, 1, 2)
MAX_TRUNCATION = (0.15, 0.3, 0.5)


def clean(gt, dt, cls, difficulty):
    """ignore flags: 0 = counts, 1 = neighbouring class / too hard (neither tp nor fp), -1 = other class
    [eval.py:35-87]"""
    name = CLASS_NAMES[cls]
    ig_gt, ig_dt, dc = [], [], []
    nvalid = 0
    for i in range(len(gt["name"])):
        g = gt["name"][i].lower()
        h = gt["bbox"][i][3] - gt["bbox"][i][1]
        if g == name:
            kind = 1
        elif (name == "pedestrian" and g == "person_sitting") or (name == "car" and g == "van"):
            kind = 0
        else:
            kind = -1
        hard = (gt["occluded"][i] > MAX_OCCLUSION[difficulty] or gt["truncated"][i] > MAX_TRUNCATION[difficulty]
                or h <= MIN_HEIGHT[difficulty])
        if kind == 1 and not hard:
            ig_gt.append(0); nvalid += 1
        elif kind == 0 or (hard and kind == 1):
            ig_gt.append(1)
        else:
            ig_gt.append(-1)
        if gt["name"][i] == "DontCare":
            dc.append(gt["bbox"][i])
    for i in range(len(dt["name"])):
        h = abs(dt["bbox"][i][3] - dt["bbox"][i][1])
        if h < MIN_HEIGHT[difficulty]:
            ig_dt.append(1)
        elif dt["name"][i].lower() == name:
            ig_dt.append(0)
        else:
            ig_dt.append(-1)
    return nvalid, ig_gt, ig_dt, np.asarray(dc, dtype=np.float64).reshape(-1, 4)


def statistics(overlaps, gt_datas, dt_datas, ig_gt, ig_dt, dc, metric, min_overlap, thresh=0.0, compute_fp=False,
               compute_aos=False):
    """one frame -> tp, fp, fn, similarity, scores of the true positives   [eval.py:167-285]
    overlaps[det, gt]; gt_datas = bbox + alpha; dt_datas = bbox + alpha + score"""
    nd, ng = len(dt_datas), len(gt_datas)
    NONE = -10000000
    taken = [False] * nd
    low = [compute_fp and dt_datas[j][5] < thresh for j in range(nd)]
    tp = fp = fn = 0
    similarity = 0
    tp_scores, deltas = [], []
    for i in range(ng):
        if ig_gt[i] == -1:
            continue
        best, valid, top, via_ignored = -1, NONE, 0, False
        for j in range(nd):
            if ig_dt[j] == -1 or taken[j] or low[j]:
                continue
            ov, sc = overlaps[j][i], dt_datas[j][5]
            if not compute_fp and ov > min_overlap and sc > valid:
                best, valid = j, sc
            elif compute_fp and ov > min_overlap and (ov > top or via_ignored) and ig_dt[j] == 0:
                top, best, valid, via_ignored = ov, j, 1, False
            elif compute_fp and ov > min_overlap and valid == NONE and ig_dt[j] == 1:
                best, valid, via_ignored = j, 1, True
        if valid == NONE and ig_gt[i] == 0:
            fn += 1
        elif valid != NONE and (ig_gt[i] == 1 or ig_dt[best] == 1):
            taken[best] = True
        elif valid != NONE:
            tp += 1
            tp_scores.append(dt_datas[best][5])
            if compute_aos:
                deltas.append(gt_datas[i][4] - dt_datas[best][4])
            taken[best] = True
    if compute_fp:
        for j in range(nd):
            if not (taken[j] or ig_dt[j] == -1 or ig_dt[j] == 1 or low[j]):
                fp += 1
        stuff = 0
        if metric == 0 and len(dc) and nd:
            ov = image_overlap(np.asarray(dt_datas)[:, :4], dc, 0)
            for i in range(len(dc)):
                for j in range(nd):
                    if taken[j] or ig_dt[j] in (-1, 1) or low[j]:
                        continue
                    if ov[j, i] > min_overlap:
                        taken[j] = True
                        stuff += 1
        fp -= stuff
        if compute_aos:
            similarity = sum((1.0 + math.cos(d)) / 2.0 for d in deltas) if (tp > 0 or fp > 0) else -1
    return tp, fp, fn, similarity, tp_scores


def score_thresholds(scores, num_gt, num_sample_pts=41):
    """[eval.py:14-32]"""
    scores = sorted(scores, reverse=True)
    out, recall = [], 0
    for i, s in enumerate(scores):
        left = (i + 1) / num_gt
        right = (i + 2) / num_gt if i < len(scores) - 1 else left
        if (right - recall) < (recall - left) and i < len(scores) - 1:
            continue
        out.append(s)
        recall += 1 / (num_sample_pts - 1.0)
    return out


def _overlap_of_frame(gt, dt, metric):
    """(det, gt) overlap matrix of one frame   [eval.py:347-422 with the swapped call of :483]"""
    if metric == 0:
        return image_overlap(dt["bbox"], gt["bbox"])
    if metric == 1:
        def bev(a):
            return np.concatenate([np.asarray(a["location"]).reshape(-1, 3)[:, [0, 2]],
                                   np.asarray(a["dimensions"]).reshape(-1, 3)[:, [0, 2]],
                                   np.asarray(a["rotation_y"]).reshape(-1, 1)], axis=1)
        return rotate_iou(bev(dt), bev(gt)).astype(np.float64)

    def full(a):
        return np.concatenate([np.asarray(a["location"]).reshape(-1, 3), np.asarray(a["dimensions"]).reshape(-1, 3),
                               np.asarray(a["rotation_y"]).reshape(-1, 1)], axis=1)
    return box3d_overlap(full(dt), full(gt))


def eval_class(gt_annos, dt_annos, classes, difficulties, metric, min_overlaps, compute_aos=False):
    """-> precision, recall, orientation arrays [class, difficulty, overlap, 41]   [eval.py:456-574]
    (the reference batches frames into parts for speed; per frame the arithmetic is the same)"""
    n = len(gt_annos)
    overlaps = [_overlap_of_frame(gt_annos[i], dt_annos[i], metric) for i in range(n)]
    shape = (len(classes), len(difficulties), len(min_overlaps), 41)
    precision, recall, aos = np.zeros(shape), np.zeros(shape), np.zeros(shape)
    for m, cls in enumerate(classes):
        for l, diff in enumerate(difficulties):
            prep = []
            total_valid = 0
            for i in range(n):
                nvalid, ig_gt, ig_dt, dc = clean(gt_annos[i], dt_annos[i], cls, diff)
                total_valid += nvalid
                gd = np.concatenate([np.asarray(gt_annos[i]["bbox"], dtype=np.float64).reshape(-1, 4),
                                     np.asarray(gt_annos[i]["alpha"], dtype=np.float64).reshape(-1, 1)], axis=1)
                dd = np.concatenate([np.asarray(dt_annos[i]["bbox"], dtype=np.float64).reshape(-1, 4),
                                     np.asarray(dt_annos[i]["alpha"], dtype=np.float64).reshape(-1, 1),
                                     np.asarray(dt_annos[i]["score"], dtype=np.float64).reshape(-1, 1)], axis=1)
                prep.append((gd, dd, ig_gt, ig_dt, dc))
            for k, mo in enumerate(min_overlaps[:, metric, m]):
                scores = []
                for i in range(n):
                    gd, dd, ig_gt, ig_dt, dc = prep[i]
                    scores += statistics(overlaps[i], gd, dd, ig_gt, ig_dt, dc, metric, mo, 0.0, False)[4]
                thr = score_thresholds(scores, total_valid)
                pr = np.zeros((len(thr), 4))
                for i in range(n):
                    gd, dd, ig_gt, ig_dt, dc = prep[i]
                    for t, th in enumerate(thr):
                        tp, fp, fn, sim, _ = statistics(overlaps[i], gd, dd, ig_gt, ig_dt, dc, metric, mo, th, True, compute_aos)
                        pr[t, 0] += tp; pr[t, 1] += fp; pr[t, 2] += fn
                        if sim != -1:
                            pr[t, 3] += sim
                with np.errstate(divide="ignore", invalid="ignore"):
                    for t in range(len(thr)):
                        recall[m, l, k, t] = pr[t, 0] / (pr[t, 0] + pr[t, 2])
                        precision[m, l, k, t] = pr[t, 0] / (pr[t, 0] + pr[t, 1])
                        if compute_aos:
                            aos[m, l, k, t] = pr[t, 3] / (pr[t, 0] + pr[t, 1])
                    for t in range(len(thr)):
                        precision[m, l, k, t] = np.max(precision[m, l, k, t:])
                        recall[m, l, k, t] = np.max(recall[m, l, k, t:])
                        if compute_aos:
                            aos[m, l, k, t] = np.max(aos[m, l, k, t:])
    return precision, recall, aos


def ap40(prec):
    """[eval.py:584-588]"""
    return prec[..., 1:].sum(axis=-1) / 40 * 100


NAME_TO_CLASS = {"Car": 0, "Pedestrian": 1, "Cyclist": 2, "Van": 3, "Person_sitting": 4}
MIN_OVERLAPS = np.stack([np.array([[0.7, 0.5, 0.5, 0.7, 0.5]] * 3),
                         np.array([[0.7, 0.5, 0.5, 0.7, 0.5], [0.5, 0.25, 0.25, 0.5, 0.25], [0.5, 0.25, 0.25, 0.5, 0.25]])])


def kitti_eval(gt_annos, dt_annos, current_classes, eval_types=("bbox", "bev", "3d")):
    """-> dict 'KITTI/<Class>_<3D|BEV|2D>_AP40_<easy|moderate|hard>_<strict|loose>' (+ Overall_*)   [eval.py:666-812]
    (the printed table of the reference is formatted from the same numbers and is not restated)"""
    classes = [NAME_TO_CLASS[c] if isinstance(c, str) else c for c in current_classes]
    mo = MIN_OVERLAPS[:, :, classes]
    names = {v: k for k, v in NAME_TO_CLASS.items()}
    res = {}
    ap = {}
    for key, metric in (("bbox", 0), ("bev", 1), ("3d", 2)):
        if key in eval_types:
            ap[key] = ap40(eval_class(gt_annos, dt_annos, classes, (0, 1, 2), metric, mo)[0])
    tag = {"3d": "3D", "bev": "BEV", "bbox": "2D"}
    for j, c in enumerate(classes):
        for i in range(2):
            for d, dn in enumerate(("easy", "moderate", "hard")):
                for key in ("3d", "bev", "bbox"):
                    if key in ap:
                        res["KITTI/%s_%s_AP40_%s_%s" % (names[c], tag[key], dn, "strict" if i == 0 else "loose")] = ap[key][j, d, i]
    if len(classes) > 1:
        for d, dn in enumerate(("easy", "moderate", "hard")):
            for key in ("3d", "bev", "bbox"):
                if key in ap:
                    res["KITTI/Overall_%s_AP40_%s" % (tag[key], dn)] = ap[key].mean(axis=0)[d, 0]
    return res
