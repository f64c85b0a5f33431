"""transforms/augmentations.py -- the reference's random training augmentations restated without cv2 (parity unpinned: the
reference modules need cv2 to import, and the transforms are random).  What is checked: the colour conversion against
python's colorsys and its own inverse, the invariants of every geometric transform (labels, calibration and pixels move
together; flipping twice is the identity; a transform that does not fire leaves the sample alone), the mask bookkeeping,
and the sample contract of the full training list on the mini KITTI tree."""
import colorsys
import copy
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

MINI = os.path.join(GOLDEN, "kitti_mini")


class Script:
    """an ``rng`` whose draws are scripted: coins / units / uniforms / integers are popped from lists"""

    def __init__(self, coins=(), units=(), uniforms=(), ints=(), perm=None):
        self.coins, self.units, self.uniforms, self.ints, self.perm = list(coins), list(units), list(uniforms), list(ints), perm

    def integers(self, lo, hi=None):
        if hi is None:                        # a coin: integers(2)
            return self.coins.pop(0)
        return self.ints.pop(0)

    def random(self):
        return self.units.pop(0)

    def uniform(self, lo, hi):
        v = self.uniforms.pop(0)
        assert lo <= v <= hi
        return v

    def permutation(self, n):
        return np.array(self.perm)


def raw_sample(idx=0):
    """an untransformed sample of the mini tree: uint8 HWC image, numpy labels, calibration"""
    from dataset.monocon_dataset import MonoConDataset
    ds = MonoConDataset(MINI, "val", transforms=[])
    return ds[idx]


# ------------------------------------------------------------------------------------------------ colour
def test_hsv_conversion_against_colorsys_and_its_inverse():
    from transforms.augmentations import bgr_to_hsv, hsv_to_bgr
    rng = np.random.default_rng(0)
    img = rng.uniform(0, 255, (16, 24, 3)).astype(np.float32)
    hsv = bgr_to_hsv(img)
    for y, x in ((0, 0), (3, 7), (15, 23), (8, 8)):
        b, g, r = (float(v) for v in img[y, x])
        h, s, v = colorsys.rgb_to_hsv(r / 255, g / 255, b / 255)
        assert hsv[y, x, 0] == pytest.approx(h * 360, abs=1e-2) and hsv[y, x, 1] == pytest.approx(s, abs=1e-5)
        assert hsv[y, x, 2] == pytest.approx(v * 255, abs=1e-3)
    assert np.abs(hsv_to_bgr(hsv) - img).max() < 2e-3
    pure = np.array([[[0, 0, 255], [0, 255, 0], [255, 0, 0], [90, 90, 90], [0, 0, 0]]], dtype=np.float32)     # BGR: red, green, blue, grey, black
    got = bgr_to_hsv(pure)[0]
    assert got[:3, 0].tolist() == [0.0, 120.0, 240.0] and np.allclose(got[:3, 1], 1.0, atol=1e-6) and got[3, 1] == 0 and got[4].tolist() == [0, 0, 0]
    # hue outside [0, 360) wraps; saturation above 1 is not clamped (float images, as in OpenCV)
    assert np.allclose(hsv_to_bgr(np.array([[[400.0, 1.0, 200.0]]], np.float32)), hsv_to_bgr(np.array([[[40.0, 1.0, 200.0]]], np.float32)))
    assert hsv_to_bgr(np.array([[[0.0, 1.5, 100.0]]], np.float32))[0, 0].min() == pytest.approx(-50.0)


def test_photometric_distortion_branches():
    from transforms import PhotometricDistortion
    s = raw_sample()
    img = s["img"]
    same = PhotometricDistortion(rng=Script(coins=[0, 0, 0, 0, 0, 0]))({"img": img.copy()})["img"]       # nothing fires (mode 0)
    assert same.dtype == np.float32 and same.shape == img.shape and np.abs(same - img).max() < 2e-3
    bright = PhotometricDistortion(rng=Script(coins=[1, 1, 0, 0, 0, 0], uniforms=[10.0]))({"img": img.copy()})["img"]
    assert np.abs(bright - (img.astype(np.float32) + 10)).max() < 3e-3
    contrast = PhotometricDistortion(rng=Script(coins=[0, 1, 1, 0, 0, 0], uniforms=[1.25]))({"img": img.copy()})["img"]
    assert np.abs(contrast - img.astype(np.float32) * 1.25).max() < 3e-3
    late = PhotometricDistortion(rng=Script(coins=[0, 0, 0, 0, 1, 0], uniforms=[0.5]))({"img": img.copy()})["img"]
    assert np.abs(late - img.astype(np.float32) * 0.5).max() < 3e-3
    swapped = PhotometricDistortion(rng=Script(coins=[0, 0, 0, 0, 0, 1], perm=[2, 1, 0]))({"img": img.copy()})["img"]
    assert np.abs(swapped - img[:, :, ::-1]).max() < 2e-3                    # BGR permuted by (2,1,0) = channels reversed
    grey = PhotometricDistortion(rng=Script(coins=[0, 0, 1, 0, 0, 0], uniforms=[0.5]))({"img": img.copy()})["img"]
    f = img.astype(np.float32)
    assert np.abs(grey.max(-1) - f.max(-1)).max() < 2e-3                     # value is kept, saturation halves
    assert np.abs((grey.max(-1) - grey.min(-1)) - 0.5 * (f.max(-1) - f.min(-1))).max() < 2e-2
    hue = PhotometricDistortion(rng=Script(coins=[0, 0, 0, 1, 0, 0], uniforms=[18.0]))({"img": img.copy()})["img"]
    assert np.abs(hue.max(-1) - f.max(-1)).max() < 2e-3 and np.abs(hue - f).max() > 1.0
    # with a real generator: reproducible per seed
    a = PhotometricDistortion(rng=np.random.default_rng(4))({"img": img.copy()})["img"]
    b = PhotometricDistortion(rng=np.random.default_rng(4))({"img": img.copy()})["img"]
    assert np.array_equal(a, b)


# ------------------------------------------------------------------------------------------------ geometry
def test_random_shift_moves_pixels_labels_and_principal_point_together():
    from transforms import RandomShift
    s = raw_sample()
    before = copy.deepcopy(s)
    out = RandomShift(prob=0.5, rng=Script(units=[0.9]))(copy.deepcopy(s))          # does not fire
    assert out["img_metas"]["is_shifted"] is False and out["img_metas"]["shift_params"] == (0, 0)
    assert np.array_equal(out["img"], before["img"]) and np.array_equal(out["calib"].P2, before["calib"].P2)
    sx, sy = 17, -9
    out = RandomShift(prob=0.5, rng=Script(units=[0.1], uniforms=[17.9, -9.9]))(copy.deepcopy(s))   # int() truncates towards 0
    assert out["img_metas"]["is_shifted"] is True and out["img_metas"]["shift_params"] == (sx, sy)
    H, W = before["img_metas"]["ori_shape"]
    assert np.array_equal(out["img"][0:H + sy, sx:W], before["img"][-sy:H, 0:W - sx])
    assert not out["img"][H + sy:, :].any() and not out["img"][:, :sx].any()
    assert out["calib"].P2[0, 2] == before["calib"].P2[0, 2] + sx and out["calib"].P2[1, 2] == before["calib"].P2[1, 2] + sy
    assert out["calib"].cu == out["calib"].P2[0, 2]
    keep = out["label"]["mask"].astype(bool)
    was = before["label"]["mask"].astype(bool)
    assert keep.any() and not (keep & ~was).any()
    assert np.allclose(out["label"]["centers2d"][keep], before["label"]["centers2d"][keep] + [sx, sy])
    assert np.allclose(out["label"]["gt_kpts_2d"][keep].reshape(-1, 9, 2), before["label"]["gt_kpts_2d"][keep].reshape(-1, 9, 2) + [sx, sy])
    exp = before["label"]["gt_bboxes"][keep] + [sx, sy, sx, sy]
    exp[:, 0::2] = exp[:, 0::2].clip(0, W); exp[:, 1::2] = exp[:, 1::2].clip(0, H)
    assert np.allclose(out["label"]["gt_bboxes"][keep], exp)
    assert np.array_equal(out["label"]["gt_bboxes_3d"][keep], before["label"]["gt_bboxes_3d"][keep])     # the scene did not move
    for k, v in out["label"].items():
        if k != "mask":
            assert not np.asarray(v)[~keep].any(), k                        # dropped rows are zero everywhere
    # a shift that pushes every box out of the frame leaves the sample untouched
    far = RandomShift(prob=1.0, shift_range=(-5000, 5000), rng=Script(units=[0.0], uniforms=[4000.0, 0.0]))(copy.deepcopy(s))
    assert far["img_metas"]["is_shifted"] is False and np.array_equal(far["img"], before["img"])


def test_horizontal_flip_is_an_involution_and_mirrors_consistently():
    from transforms import RandomHorizontalFlip
    s = raw_sample()
    before = copy.deepcopy(s)
    off = RandomHorizontalFlip(prob=0.5, rng=Script(units=[0.7]))(copy.deepcopy(s))
    assert off["img_metas"]["is_flipped"] is False and np.array_equal(off["img"], before["img"])
    once = RandomHorizontalFlip(prob=1.0, rng=Script(units=[0.0]))(copy.deepcopy(s))
    W = before["img"].shape[1]
    keep = before["label"]["mask"].astype(bool)
    assert once["img_metas"]["is_flipped"] is True and np.array_equal(once["img"], before["img"][:, ::-1])
    assert once["calib"].P2[0, 2] == pytest.approx(W - before["calib"].P2[0, 2] - 1) and once["calib"].P2[0, 3] == -before["calib"].P2[0, 3]
    assert np.allclose(once["label"]["centers2d"][keep][:, 0], W - before["label"]["centers2d"][keep][:, 0] - 1)
    assert np.allclose(once["label"]["gt_bboxes"][keep][:, [0, 2]], W - before["label"]["gt_bboxes"][keep][:, [2, 0]])
    assert np.allclose(once["label"]["gt_bboxes_3d"][keep][:, 0], -before["label"]["gt_bboxes_3d"][keep][:, 0])
    assert np.allclose(once["label"]["gt_bboxes_3d"][keep][:, 6], np.pi - before["label"]["gt_bboxes_3d"][keep][:, 6])
    kb, ka = before["label"]["gt_kpts_2d"][keep].reshape(-1, 9, 2), once["label"]["gt_kpts_2d"][keep].reshape(-1, 9, 2)
    assert np.allclose(ka[:, [1, 0, 3, 2, 5, 4, 7, 6, 8], 0], W - kb[:, :, 0] - 1) and np.allclose(ka[:, [1, 0, 3, 2, 5, 4, 7, 6, 8], 1], kb[:, :, 1])
    # a mirrored scene projects to mirrored pixels: the same 3D point, x negated, through the flipped P2 (up to the
    # (w - 1) * tz / z of KITTI's small P2[2, 3], which the reference's flip ignores as well: < 0.3 px beyond 10 m)
    P, Pf = before["calib"].P2.astype(np.float64), once["calib"].P2.astype(np.float64)
    for pt in ([2.0, 1.5, 20.0], [-7.0, 1.0, 35.0]):
        u = (P @ np.array([*pt, 1.0]))
        uf = (Pf @ np.array([-pt[0], pt[1], pt[2], 1.0]))
        assert uf[0] / uf[2] == pytest.approx(W - u[0] / u[2] - 1, abs=0.3) and uf[1] / uf[2] == pytest.approx(u[1] / u[2], abs=1e-3)
    twice = RandomHorizontalFlip(prob=1.0, rng=Script(units=[0.0]))(copy.deepcopy(once))
    assert np.array_equal(twice["img"], before["img"]) and np.allclose(twice["calib"].P2, before["calib"].P2, atol=1e-4)
    for k in ("gt_bboxes", "centers2d", "gt_kpts_2d", "gt_kpts_valid_mask"):
        assert np.allclose(twice["label"][k], before["label"][k], atol=1e-3), k
    assert np.allclose(twice["label"]["gt_bboxes_3d"][:, :6], before["label"]["gt_bboxes_3d"][:, :6])


def test_crops_blank_the_outside_and_drop_what_left_the_window():
    from transforms import RandomCrop3D, RandomRangeCrop3D
    s = raw_sample()
    before = copy.deepcopy(s)
    off = RandomCrop3D(prob=0.5, rng=Script(units=[0.6]))(copy.deepcopy(s))
    assert off["img_metas"]["is_cropped"] is False and off["img_metas"]["crop_coord"] == (0, 0, 0, 0)
    assert np.array_equal(off["img"], before["img"])
    out = RandomCrop3D(prob=1.0, crop_size=(320, 960), hide_kpts_in_crop_area=True, rng=Script(units=[0.0], ints=[40, 200]))(copy.deepcopy(s))
    x0, y0, x1, y1 = out["img_metas"]["crop_coord"]
    assert (x0, y0, x1, y1) == (200, 40, 1160, 360) and out["img_metas"]["is_cropped"] is True
    assert np.array_equal(out["img"][y0:y1, x0:x1], before["img"][y0:y1, x0:x1])
    blank = out["img"].copy(); blank[y0:y1, x0:x1] = 0
    assert not blank.any() and out["img"].shape == before["img"].shape
    assert np.array_equal(out["calib"].P2, before["calib"].P2)                        # geometry untouched
    was, now = before["label"]["mask"].astype(bool), out["label"]["mask"].astype(bool)
    assert not (now & ~was).any()
    for i in np.nonzero(was)[0]:
        bx = before["label"]["gt_bboxes"][i]
        ix = max(0.0, min(bx[2], x1) - max(bx[0], x0)); iy = max(0.0, min(bx[3], y1) - max(bx[1], y0))
        frac = ix * iy / ((bx[2] - bx[0]) * (bx[3] - bx[1]))
        assert now[i] == (frac >= 0.2), (i, frac)
        if now[i]:
            nb = out["label"]["gt_bboxes"][i]
            assert nb[0] >= x0 - 1e-3 and nb[1] >= y0 - 1e-3 and nb[2] <= x1 + 1e-3 and nb[3] <= y1 + 1e-3
            kp = out["label"]["gt_kpts_2d"][i].reshape(9, 2)
            outside = ~((kp[:, 0] >= x0) & (kp[:, 0] <= x1) & (kp[:, 1] >= y0) & (kp[:, 1] <= y1))
            assert (out["label"]["gt_kpts_valid_mask"][i][outside] == 1).all()
    # a window that keeps no object: RandomCrop3D hands the frame on unchanged, RandomRangeCrop3D really crops
    empty_win = dict(units=[0.0], ints=[0, 0])
    keepit = RandomCrop3D(prob=1.0, crop_size=(8, 8), rng=Script(**empty_win))(copy.deepcopy(s))
    assert np.array_equal(keepit["img"], before["img"]) and np.array_equal(keepit["label"]["mask"], before["label"]["mask"])
    rr = RandomRangeCrop3D(prob=1.0, height_range=(256, 320), aspect_ratio=3.0, rng=Script(units=[0.0], ints=[300, 900, 10, 100]))(copy.deepcopy(s))
    assert rr["img_metas"]["crop_coord"] == (100, 10, 1000, 310) and not rr["img"][:10].any() and rr["img"][10:310, 100:1000].any()
    with pytest.raises(AssertionError):
        RandomCrop3D(prob=1.0, crop_size=(400, 960))(copy.deepcopy(s))


def test_training_list_keeps_the_sample_contract_and_is_seedable():
    from dataset.monocon_dataset import MonoConDataset, default_train_transforms
    from transforms import Compose, Resize3D, Convert_3D_to_4D
    names = [t.__class__.__name__ for t in default_train_transforms()]
    assert names == ["PhotometricDistortion", "RandomShift", "RandomHorizontalFlip", "RandomCrop3D", "Normalize", "Pad", "ToTensor"]
    ds_val = MonoConDataset(MINI, "val")
    assert [t.__class__.__name__ for t in ds_val.transforms.transforms] == names[-3:]     # no augmentation outside 'train'

    def run(seed, idx):
        ds = MonoConDataset(MINI, "val", transforms=default_train_transforms(np.random.default_rng(seed)))
        return ds[idx]
    a, b, c = run(3, 0), run(3, 0), run(4, 0)
    assert torch.equal(a["img"], b["img"]) and all(torch.equal(a["label"][k], b["label"][k]) for k in a["label"])
    assert not torch.equal(a["img"], c["img"])
    ref = ds_val[0]
    assert a["img"].shape == ref["img"].shape and a["img"].dtype == torch.float32
    assert {k: (tuple(v.shape), v.dtype) for k, v in a["label"].items()} == {k: (tuple(v.shape), v.dtype) for k, v in ref["label"].items()}
    for key in ("is_shifted", "shift_params", "is_flipped", "is_cropped", "crop_coord", "pad_shape"):
        assert key in a["img_metas"]
    for seed in range(6):                                    # any draw collates with any other
        batch = MonoConDataset.collate_fn([run(seed, 0), run(seed + 10, 1)])
        assert tuple(batch["img"].shape) == (2, 3, 384, 1248) and bool(torch.isfinite(batch["img"]).all())
        assert batch["label"]["mask"].sum() >= 1
    # Resize3D rescales image, calibration and 2D labels together; Convert_3D_to_4D makes a batch of one
    s = raw_sample()
    fu, cu = float(s["calib"].P2[0, 0]), float(s["calib"].P2[0, 2])
    c0 = s["label"]["centers2d"].copy()
    r = Resize3D((188, 621))(s)
    assert r["img"].shape == (188, 621, 3) and r["img"].dtype == np.uint8 and r["img_metas"]["ori_shape"] == (188, 621)
    assert float(r["calib"].P2[0, 0]) == pytest.approx(fu * 0.5, rel=1e-6) and float(r["calib"].P2[0, 2]) == pytest.approx(cu * 0.5, rel=1e-6)
    assert np.allclose(r["label"]["centers2d"][:, 0], c0[:, 0] * 0.5) and np.allclose(r["label"]["centers2d"][:, 1], c0[:, 1] * (188 / 375))
    one = Compose(ds_val.transforms.transforms + [Convert_3D_to_4D()])(raw_sample())
    assert tuple(one["img"].shape) == (1, 3, 384, 1248) and isinstance(one["calib"], list) and one["img_metas"]["sample_idx"] == [7]
