"""SURVEY 8f-4, last row: the KITTI AP evaluator (engine/kitti_eval) -- HIP rotated-overlap kernels + native matching
behind the C-ABI -- against oracle/kitti_eval_oracle.py.

The oracle's host logic (ignore rules, matching, recall sampling, AP) is pinned to the reference's own functions in
tests/test_f4_reference_golden.py (round 6).  Its rotated-overlap arithmetic stays PARITY UNPINNED (a float32 numba.cuda
kernel upstream, not executable here) and is held by what this file does: closed-form answers and an independent float64
polygon-clipping implementation (``clip_area`` below: Sutherland-Hodgman, shares nothing with the vertex-collection /
angle-sort formulation)."""
import math
import os

import numpy as np
import pytest

from oracle import kitti_eval_oracle as KO


# ------------------------------------------------------------------------------------------------ independent geometry
def corners64(b):
    cx, cy, dx, dy, a = (float(v) for v in b)
    c, s = math.cos(a), math.sin(a)
    pts = []
    for lx, ly in ((-dx / 2, -dy / 2), (-dx / 2, dy / 2), (dx / 2, dy / 2), (dx / 2, -dy / 2)):
        pts.append((c * lx + s * ly + cx, -s * lx + c * ly + cy))
    return pts


def signed_area(poly):
    return 0.5 * sum(poly[i][0] * poly[(i + 1) % len(poly)][1] - poly[(i + 1) % len(poly)][0] * poly[i][1] for i in range(len(poly)))


def clip_area(b1, b2):
    """area of the intersection of two rotated boxes, float64 Sutherland-Hodgman"""
    subj, clip = corners64(b1), corners64(b2)
    if signed_area(subj) < 0:
        subj = subj[::-1]
    if signed_area(clip) < 0:
        clip = clip[::-1]
    out = subj
    for i in range(4):
        a, b = clip[i], clip[(i + 1) % 4]
        inp, out = out, []
        if not inp:
            break

        def side(p):
            return (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
        for j in range(len(inp)):
            p, q = inp[j], inp[(j + 1) % len(inp)]
            sp, sq = side(p), side(q)
            if sp >= 0:
                out.append(p)
            if (sp > 0 and sq < 0) or (sp < 0 and sq > 0):
                t = sp / (sp - sq)
                out.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
    return abs(signed_area(out)) if len(out) >= 3 else 0.0


def random_rboxes(rng, n, spread=6.0):
    return np.stack([rng.uniform(-spread, spread, n), rng.uniform(10, 10 + 2 * spread, n), rng.uniform(0.5, 5.0, n),
                     rng.uniform(0.5, 2.5, n), rng.uniform(-math.pi, math.pi, n)], axis=1)


# ------------------------------------------------------------------------------------------------ oracle pins (CPU)
def test_oracle_rotated_overlap_closed_forms():
    sq = [0, 0, 2, 2, 0.0]
    assert KO.rotate_iou([sq], [sq])[0, 0] == pytest.approx(1.0, abs=1e-6)
    assert KO.rotate_iou([sq], [[5, 0, 2, 2, 0.3]])[0, 0] == 0.0
    # axis-aligned, shifted by (1, 0.5): intersection 1 x 1.5
    assert KO.rotate_iou([sq], [[1, 0.5, 2, 2, 0.0]], 2)[0, 0] == pytest.approx(1.5, abs=1e-6)
    assert KO.rotate_iou([sq], [[1, 0.5, 2, 2, 0.0]])[0, 0] == pytest.approx(1.5 / 6.5, abs=1e-6)
    # a square against itself turned by 45 degrees: a regular octagon, 2 (sqrt 2 - 1) s^2
    assert KO.rotate_iou([sq], [[0, 0, 2, 2, math.pi / 4]], 2)[0, 0] == pytest.approx(8 * (math.sqrt(2) - 1), abs=1e-5)
    # a quarter turn maps a 4 x 1 box onto a 1 x 4 box: the common part is the unit square
    assert KO.rotate_iou([[0, 0, 4, 1, 0.0]], [[0, 0, 4, 1, math.pi / 2]], 2)[0, 0] == pytest.approx(1.0, abs=1e-5)
    # criterion 0 divides by the QUERY box's area, 1 by the box's (the reference hands the query in as rbox1)
    big, small = [0, 0, 4, 4, 0.0], [0, 0, 2, 1, 0.2]
    assert KO.rotate_iou([big], [small], 0)[0, 0] == pytest.approx(1.0, abs=1e-5)
    assert KO.rotate_iou([big], [small], 1)[0, 0] == pytest.approx(2.0 / 16.0, abs=1e-5)
    # (NOT asserted: boxes that coincide up to float32 round-off, e.g. the same box turned by 2 pi.  The reference's
    #  formulation -- >= vertex tests plus strict edge-crossing tests -- loses vertices there and returns 0.63 instead of 1;
    #  the restatement and the kernel reproduce the formulation, not the ideal geometry.)


def test_oracle_rotated_overlap_vs_polygon_clipping():
    rng = np.random.default_rng(3)
    a, b = random_rboxes(rng, 40), random_rboxes(rng, 40)
    got = KO.rotate_iou(a, b, 2).astype(np.float64)
    ref = np.array([[clip_area(x, y) for y in b] for x in a])
    assert (ref > 0.05).sum() > 100                       # the sample does overlap
    assert np.abs(got - ref).max() < 2e-4, np.abs(got - ref).max()
    iou = KO.rotate_iou(a, b).astype(np.float64)
    areas = (a[:, 2] * a[:, 3])[:, None] + (b[:, 2] * b[:, 3])[None, :]
    assert np.abs(iou - ref / (areas - ref)).max() < 5e-5
    assert np.allclose(KO.rotate_iou(b, a), iou.T, atol=2e-5)           # symmetric up to round-off


def test_oracle_3d_overlap_closed_forms():
    # two 4 x 1.5 x 2 (l, h, w) boxes on the same ground spot, one lifted by half its height.  Axis-aligned on purpose:
    # for a TURNED box the float32 vertex tests of this formulation are not reliable against an exact copy of itself
    # (a perpendicular dot product rounds to -1e-7, the corner counts as outside, the polygon collapses: overlap 0 instead
    # of 1) -- a property of the reference's algorithm that restatement and kernel share; exact duplicates do not occur
    # between detections and labels
    a = [0.0, 1.5, 20.0, 4.0, 1.5, 2.0, 0.0]
    b = [0.0, 0.75, 20.0, 4.0, 1.5, 2.0, 0.0]
    assert KO.box3d_overlap([a], [a])[0, 0] == pytest.approx(1.0, abs=1e-5)
    assert KO.box3d_overlap([a], [b])[0, 0] == pytest.approx(6.0 / 18.0, abs=1e-5)           # half of one volume shared
    assert KO.box3d_overlap([a], [[0.0, -1.0, 20.0, 4.0, 1.5, 2.0, 0.0]])[0, 0] == 0.0       # no common height
    assert KO.box3d_overlap([a], [b], 0)[0, 0] == pytest.approx(0.5, abs=1e-5)


def test_oracle_matching_by_hand():
    """one frame, two cars: a confident hit, a weak hit below the overlap threshold, a false positive"""
    gt = np.array([[100, 100, 200, 200, 0.0], [300, 100, 400, 200, 0.0]])
    dt = np.array([[100, 100, 200, 200, 0.0, 0.9], [300, 100, 360, 200, 0.0, 0.8], [600, 100, 700, 200, 0.0, 0.7]])
    ov = KO.image_overlap(dt[:, :4], gt[:, :4])
    assert ov[0, 0] == 1.0 and ov[1, 1] == pytest.approx(0.6) and ov[2].max() == 0
    ig_gt, ig_dt, dc = [0, 0], [0, 0, 0], np.zeros((0, 4))
    tp, fp, fn, sim, sc = KO.statistics(ov, gt, dt, ig_gt, ig_dt, dc, 0, 0.7, 0.0, True, True)
    assert (tp, fp, fn) == (1, 2, 1) and sc == [0.9] and sim == pytest.approx(1.0)
    tp, fp, fn, _, sc = KO.statistics(ov, gt, dt, ig_gt, ig_dt, dc, 0, 0.5, 0.0, True)
    assert (tp, fp, fn) == (2, 1, 0) and sorted(sc) == [0.8, 0.9]
    assert KO.statistics(ov, gt, dt, ig_gt, ig_dt, dc, 0, 0.5, 0.85, True)[:3] == (1, 0, 1)   # score threshold drops two
    # a DontCare region swallows the false positive (2D metric only)
    dcb = np.array([[590.0, 90.0, 710.0, 210.0]])
    assert KO.statistics(ov, gt, dt, ig_gt, ig_dt, dcb, 0, 0.5, 0.0, True)[:3] == (2, 0, 0)
    assert KO.statistics(ov, gt, dt, ig_gt, ig_dt, dcb, 1, 0.5, 0.0, True)[:3] == (2, 1, 0)
    # an ignored ground truth (e.g. a van next to cars) absorbs its detection: neither tp nor fp
    assert KO.statistics(ov, gt, dt, [0, 1], ig_dt, dc, 0, 0.5, 0.0, True)[:3] == (1, 1, 0)
    # recall sampling: few hits -> every score is a threshold; many hits -> one per 1/40 of recall, the weakest always kept
    assert KO.score_thresholds([0.9, 0.5, 0.7], 4) == [0.9, 0.7, 0.5]
    many = list(np.linspace(0.99, 0.01, 197))
    thr = KO.score_thresholds(many, 197)
    assert len(thr) == 41 and thr[0] == many[0] and thr[-1] == many[-1] and thr == sorted(thr, reverse=True)
    assert len(KO.score_thresholds(many, 394)) == 21          # the detector found half of the objects: recall stops at 0.5


# ------------------------------------------------------------------------------------------------ synthetic annotations
from hipmonocon.synth import KITTI_NAMES as NAMES, random_kitti_annos as random_annos      # noqa: E402


def test_host_library_matches_the_oracle_on_random_frames():
    """the native host code (no device needed): 2D overlaps, ignore rules, matching in both modes, recall sampling"""
    from engine.kitti_eval import eval as E
    gts, dts = random_annos(11, frames=14)
    rng = np.random.default_rng(0)
    for crit in (-1, 0, 1, 2):
        a, b = gts[0]["bbox"], dts[0]["bbox"]
        assert np.array_equal(E.image_box_overlap(a, b, crit), KO.image_overlap(a, b, crit))
    assert E.image_box_overlap(np.zeros((0, 4)), gts[0]["bbox"]).shape == (0, len(gts[0]["bbox"]))
    n_checked = 0
    for cls in range(3):
        for diff in range(3):
            for g, d in zip(gts, dts):
                nv, ig, idt, dc = E.clean_data(g, d, cls, diff)
                onv, oig, oidt, odc = KO.clean(g, d, cls, diff)
                assert nv == onv and ig.tolist() == oig and idt.tolist() == oidt and np.array_equal(dc, odc)
                ov = KO.image_overlap(d["bbox"], g["bbox"])
                gd = np.concatenate([g["bbox"], g["alpha"][:, None]], 1).reshape(-1, 5)
                dd = np.concatenate([d["bbox"].reshape(-1, 4), d["alpha"].reshape(-1, 1), d["score"].reshape(-1, 1)], 1)
                for metric, mo in ((0, 0.5), (1, 0.7)):
                    want = KO.statistics(ov, gd, dd, oig, oidt, odc, metric, mo, 0.0, False)
                    got = E.compute_statistics_jit(ov, gd, dd, ig, idt, dc, metric, mo)
                    assert got[0] == want[0] and got[2] == want[2] and list(got[4]) == want[4]
                    th = float(rng.uniform(0, 0.6))
                    want = KO.statistics(ov, gd, dd, oig, oidt, odc, metric, mo, th, True, True)
                    got = E.compute_statistics_jit(ov, gd, dd, ig, idt, dc, metric, mo, thresh=th, compute_fp=True, compute_aos=True)
                    assert got[:3] == want[:3], (cls, diff, metric, got, want)
                    assert got[3] == pytest.approx(max(want[3], 0.0), abs=1e-12)       # (-1 = "no detections" adds nothing)
                    n_checked += 1
    assert n_checked == 3 * 3 * 14 * 2
    for n_gt in (1, 7, 40):
        s = rng.uniform(0, 1, int(rng.integers(1, 60)))
        assert E.get_thresholds(s, n_gt) == KO.score_thresholds(list(s), n_gt)
    assert E.get_split_parts(10, 3) == [3, 3, 3, 1] and E.get_split_parts(9, 3) == [3, 3, 3]


def test_2d_ap_matches_the_oracle_end_to_end_on_cpu():
    """the 2D metric needs no device: kitti_eval(eval_types=['bbox']) incl. orientation similarity vs the oracle"""
    from engine.kitti_eval import kitti_eval
    gts, dts = random_annos(5, frames=12)
    text, got = kitti_eval(gts, dts, ["Pedestrian", "Cyclist", "Car"], eval_types=["bbox"])
    want = KO.kitti_eval(gts, dts, ["Pedestrian", "Cyclist", "Car"], eval_types=("bbox",))
    assert set(got) == set(want) and len(got) == 3 * 3 * 2 + 3
    for k in want:
        assert got[k] == pytest.approx(want[k], abs=1e-9), k
    assert any(0.0 < v < 100.0 for v in got.values())                  # a non-trivial case
    assert "Car AP40@0.70, 0.70, 0.70:" in text and "aos  AP40:" in text and text.rstrip().endswith("-" * 37)
    # perfect detections.  AP40 averages the precision at recall 1/40 ... 40/40, and a recall level exists only where a
    # detection score sits: 50 easy cars, all found -> 100; the same with 11 cars fills 11 of the 41 positions -> 25
    def easy_car(i):
        return {"name": np.array(["Car"]), "truncated": np.zeros(1), "occluded": np.zeros(1), "alpha": np.array([0.1]),
                "bbox": np.array([[100.0 + i, 100.0, 200.0 + i, 180.0]]), "dimensions": np.array([[4.0, 1.5, 1.8]]),
                "location": np.array([[1.0, 1.6, 20.0 + i]]), "rotation_y": np.array([0.0]), "score": np.array([1.0 - 0.01 * i])}
    cars = [easy_car(i) for i in range(50)]
    _, res = kitti_eval(cars, cars, ["Car"], eval_types=["bbox"])
    assert all(v == pytest.approx(100.0) for v in res.values()) and len(res) == 6
    _, res = kitti_eval(cars[:11], cars[:11], ["Car"], eval_types=["bbox"])
    assert res["KITTI/Car_2D_AP40_moderate_strict"] == pytest.approx(25.0)
    # half of them missed: precision 1 up to recall 0.5, nothing beyond
    half = [c if i % 2 == 0 else {k: v[:0] for k, v in c.items()} for i, c in enumerate(cars)]
    _, res = kitti_eval(cars, half, ["Car"], eval_types=["bbox"])
    assert res["KITTI/Car_2D_AP40_easy_strict"] == pytest.approx(50.0)
    # no detections at all: AP 0
    empty = [{k: (np.zeros((0, 4)) if k == "bbox" else np.zeros((0, 3)) if k in ("dimensions", "location") else np.zeros(0))
              for k in dts[0] if k != "name"} for _ in gts]
    for e in empty:
        e["name"] = np.zeros(0, dtype=object)
    _, res = kitti_eval(gts, empty, ["Car"], eval_types=["bbox"])
    assert res["KITTI/Car_2D_AP40_easy_strict"] == 0.0


# ------------------------------------------------------------------------------------------------ device kernels
@pytest.mark.gpu
@pytest.mark.parametrize("criterion", [-1, 0, 1, 2])
def test_rotate_iou_kernel_vs_oracle(criterion):
    from engine.kitti_eval.rotate_iou import rotate_iou_gpu_eval
    rng = np.random.default_rng(17 + criterion)
    a, b = random_rboxes(rng, 70), random_rboxes(rng, 131)            # not multiples of the 64-box tile
    got = rotate_iou_gpu_eval(a, b, criterion)
    assert got.shape == (70, 131) and got.dtype == a.dtype
    want = KO.rotate_iou(a, b, criterion)
    assert (want > 0).sum() > 300
    # float32 throughout, and neither compiler's multiply-add contraction is reproducible: round-off, not bit-exact
    assert np.abs(got - want).max() < (5e-5 if criterion != 2 else 2e-4), np.abs(got - want).max()
    if criterion == 2:
        ref = np.array([[clip_area(x, y) for y in b] for x in a])
        assert np.abs(got - ref).max() < 3e-4
    assert rotate_iou_gpu_eval(a[:0], b).shape == (0, 131) and rotate_iou_gpu_eval(a, b[:0]).shape == (70, 0)


@pytest.mark.gpu
def test_rotate_iou_kernel_closed_forms_and_large_tiles():
    from engine.kitti_eval.rotate_iou import rotate_iou_gpu_eval
    sq = np.array([[0, 0, 2, 2, 0.0]])
    assert rotate_iou_gpu_eval(sq, sq)[0, 0] == pytest.approx(1.0, abs=1e-6)
    assert rotate_iou_gpu_eval(sq, np.array([[0, 0, 2, 2, math.pi / 4]]), 2)[0, 0] == pytest.approx(8 * (math.sqrt(2) - 1), abs=1e-5)
    assert rotate_iou_gpu_eval(sq, np.array([[9.0, 0, 2, 2, 1.0]]))[0, 0] == 0.0
    rng = np.random.default_rng(2)
    a = random_rboxes(rng, 700, spread=25.0)
    b = a + rng.normal(0, 2e-3, a.shape)                              # near-copies (exact copies: see the 3D closed forms)
    got = rotate_iou_gpu_eval(a, b)                                    # 11 x 11 tiles
    assert np.diag(got).min() > 0.97
    idx = rng.integers(0, 700, (80, 2))
    want = np.array([KO.rotate_iou(a[i:i + 1], b[j:j + 1])[0, 0] for i, j in idx])
    assert np.abs(got[idx[:, 0], idx[:, 1]] - want).max() < 5e-5
    # the near-copies themselves: almost parallel edges make the crossing points ill-conditioned, round-off (and the
    # compiler's multiply-add contraction) shows up at the 1e-3 level in float32 -- in the reference's kernel as well
    dg = rng.integers(0, 700, 25)
    want = np.array([KO.rotate_iou(a[i:i + 1], b[i:i + 1])[0, 0] for i in dg])
    assert np.abs(got[dg, dg] - want).max() < 5e-3


@pytest.mark.gpu
@pytest.mark.parametrize("criterion", [-1, 0, 1])
def test_box3d_overlap_kernel_vs_oracle(criterion):
    from engine.kitti_eval.rotate_iou import box3d_overlap_gpu
    rng = np.random.default_rng(23)

    def boxes(n):
        bev = random_rboxes(rng, n)
        return np.stack([bev[:, 0], rng.uniform(1.0, 2.5, n), bev[:, 1], bev[:, 2], rng.uniform(1.2, 2.2, n), bev[:, 3], bev[:, 4]], 1)
    a, b = boxes(50), boxes(77)
    got, want = box3d_overlap_gpu(a, b, criterion), KO.box3d_overlap(a, b, criterion)
    assert got.dtype == np.float64 and (want > 0).sum() > 100
    assert np.abs(got - want).max() < 5e-5
    near = box3d_overlap_gpu(a, a + rng.normal(0, 2e-3, a.shape))
    assert np.diag(near).min() > 0.95


@pytest.mark.gpu
def test_kitti_eval_all_metrics_vs_oracle():
    from engine.kitti_eval import kitti_eval
    gts, dts = random_annos(7, frames=12)
    classes = ["Pedestrian", "Cyclist", "Car"]
    text, got = kitti_eval(gts, dts, classes, eval_types=["bbox", "bev", "3d"])
    want = KO.kitti_eval(gts, dts, classes)
    assert set(got) == set(want) and len(got) == 3 * (3 * 3 * 2 + 3)
    for k in want:
        # overlaps agree to float32 round-off; an AP moves only if one of them crosses a threshold exactly
        assert got[k] == pytest.approx(want[k], abs=1e-6), k
    assert sum(0.0 < v < 100.0 for v in got.values()) > 10
    for line in ("bbox AP40:", "bev  AP40:", "3d   AP40:", "aos  AP40:", "Overall AP40@easy, moderate, hard:"):
        assert line in text


@pytest.mark.gpu
def test_dataset_evaluate_scores_the_labels_against_themselves():
    """MonoConDataset.evaluate on the two-frame mini tree: its own labels as detections -> AP 100 for the classes present"""
    from conftest import GOLDEN
    from dataset.monocon_dataset import MonoConDataset
    ds = MonoConDataset(os.path.join(GOLDEN, "kitti_mini"), "val")
    infos = ds.collect_gt_infos()
    assert len(infos) == 2 and set(infos[0]) == {"image", "calib", "annos"}
    dets = []
    for info in infos:
        a = {k: np.array(v, copy=True) for k, v in info["annos"].items()}
        keep = a["name"] != "DontCare"
        a = {k: v[keep] for k, v in a.items()}
        a["score"] = np.linspace(0.9, 0.5, keep.sum())
        a["location"] = a["location"] + 0.01          # near-copies (exact copies: see test_oracle_3d_overlap_closed_forms)
        a["rotation_y"] = a["rotation_y"] + 0.002
        dets.append(a)
    res = ds.evaluate({"img_bbox": dets, "img_bbox2d": dets}, verbose=False)
    # every labelled object is found with precision 1: AP40 = 2.5 x (number of recall positions reached - 1), and the
    # n valid objects of a class / difficulty reach n positions (see test_2d_ap_matches_the_oracle_end_to_end_on_cpu)
    from engine.kitti_eval.eval import clean_data
    gts = [i["annos"] for i in infos]
    for cls, name in enumerate(("Car", "Pedestrian", "Cyclist")):
        for diff, dn in enumerate(("easy", "moderate", "hard")):
            n = sum(clean_data(g, d, cls, diff)[0] for g, d in zip(gts, dets))
            want = 2.5 * max(n - 1, 0)
            for key in ("img_bbox/KITTI/%s_3D_AP40_%s_strict", "img_bbox/KITTI/%s_BEV_AP40_%s_loose", "img_bbox2d/KITTI/%s_2D_AP40_%s_strict"):
                assert res[key % (name, dn)] == pytest.approx(want), (key % (name, dn), n)
    assert res["img_bbox/KITTI/Car_3D_AP40_moderate_strict"] > 0
    assert not any(k.startswith("img_bbox2d") and "_3D_" in k for k in res)


def test_single_frame_helper_fn_with_an_ignored_detection():
    """a valid ground truth whose best match is a detection with ignored_det == 1 is neither a true positive
    nor a false negative (engine/kitti_eval/eval.py:228-247 of the reference); "valid ground truths minus tp" counted it
    as fn.  Two valid GTs: one matched by a scored, non-ignored detection, one only by an ignored detection."""
    from engine.kitti_eval import eval as E
    gd = np.array([[0, 0, 10, 10, 0.0], [20, 0, 30, 10, 0.0]])
    dd = np.array([[0, 0, 10, 10, 0.0, 0.9], [20, 0, 30, 10, 0.0, 0.8]])
    ov = KO.image_overlap(dd[:, :4], gd[:, :4])
    ig, idt = np.array([0, 0]), np.array([0, 1])
    tp, fp, fn, sim, scores = E.compute_statistics_jit(ov, gd, dd, ig, idt, np.zeros((0, 4)), 0, 0.5)
    want = KO.statistics(ov, gd, dd, [0, 0], [0, 1], np.zeros((0, 4)), 0, 0.5, 0.0, False)
    assert (tp, fn) == (1, 0) == (want[0], want[2]) and list(scores) == [0.9]
