"""GPU-side Normalize + Pad + ToTensor (bit-exact vs the oracle's restatement of the reference
transforms) and the data-parallel sampler (host logic, CPU)."""
import numpy as np
import pytest
import torch

from hipmonocon import synth


def test_sharded_sampler_partitions_every_epoch():
    from dataset.sharded_sampler import ShardedSampler
    n, world = 103, 8
    for epoch in (0, 1, 7):
        shards = []
        for r in range(world):
            s = ShardedSampler(n, r, world, shuffle=True, seed=5)
            s.set_epoch(epoch)
            shards.append(list(s))
            assert len(shards[-1]) == len(s) == 13            # ceil(103 / 8): equal steps on every rank
        flat = [i for sh in shards for i in sh]
        assert set(flat) == set(range(n))                     # every sample is seen ...
        assert len(flat) - len(set(flat)) == 13 * 8 - n       # ... and only the wrap-around padding repeats
    a = ShardedSampler(n, 3, world, seed=5); a.set_epoch(2)
    b = ShardedSampler(n, 3, world, seed=5); b.set_epoch(2)
    c = ShardedSampler(n, 3, world, seed=5); c.set_epoch(3)
    assert list(a) == list(b) and list(a) != list(c)          # deterministic per (seed, epoch), reshuffled per epoch
    d = [list(ShardedSampler(n, r, world, shuffle=False, drop_last=True)) for r in range(world)]
    assert all(len(x) == 12 for x in d) and sorted(i for x in d for i in x) == list(range(96))
    with pytest.raises(ValueError):
        ShardedSampler(10, 8, 8)


def test_oracle_preprocess_shapes():
    from oracle import monocon_oracle as O
    img = (np.arange(375 * 1242 * 3) % 251).astype(np.uint8).reshape(375, 1242, 3)
    t, pad = O.preprocess(img)
    assert pad == (384, 1248) and tuple(t.shape) == (3, 384, 1248) and t.dtype == torch.float32
    assert float(t[:, 375:, :].abs().max()) == 0.0 and float(t[:, :, 1242:].abs().max()) == 0.0
    assert t[0, 0, 0].item() == np.float32((np.float64(np.float32(img[0, 0, 0])) - 123.675) / 58.395)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["uint8", "float32"])
def test_gpu_preprocess_bit_exact(dtype):
    from hipmonocon.engine import Engine
    from oracle import monocon_oracle as O
    eng = Engine()
    sizes = [(375, 1242), (370, 1224), (384, 1280), (33, 65)]
    imgs = []
    for i, (h, w) in enumerate(sizes):
        a = (synth.uniform(31 + i, "raw", (h, w, 3), 0.0, 255.0)).astype(np.float32)
        imgs.append(np.floor(a).astype(np.uint8) if dtype == "uint8" else a)
    batch, pads = eng.preprocess([torch.from_numpy(a).cuda() for a in imgs])
    Hp, Wp = pads[0]
    assert (Hp, Wp) == (384, 1280) and tuple(batch.shape) == (4, 3, 384, 1280)
    for b, a in enumerate(imgs):
        ref, (hp, wp) = O.preprocess(a)
        got = batch[b].cpu()
        assert torch.equal(got[:, :hp, :wp], ref)              # bit-exact inside the image's own pad box
        assert float(got[:, hp:, :].abs().max() if hp < Hp else 0.0) == 0.0
        assert float(got[:, :, wp:].abs().max() if wp < Wp else 0.0) == 0.0


# ------------------------------------------------------------------------------------- round 3: the KITTI file dataset
import os                                                                     # noqa: E402
from conftest import GOLDEN, load_golden                                      # noqa: E402

MINI = os.path.join(GOLDEN, "kitti_mini")
FRAMES = ("000007", "000011")


def test_calibration_and_objects_match_the_reference_classes():
    """utils.data_classes on the two-frame mini tree vs the reference's own classes (tests/golden/kitti_objects.npz,
    make_golden.py dataset_pins): calibration matrices / intrinsics exactly, and per object -- in the camera-2 /
    local-yaw frame the dataset builds labels in -- box, 3D box, projected centre and the nine projected keypoints.
    The product keeps an object's label-file state and VIEWS it in a frame (the reference converts its state back and
    forth in float32), so agreement is to float32 round-off, not bit-exact: 2e-6 relative on the 3D box, 1e-3 px on
    projections."""
    from utils.data_classes import KITTICalibration, KITTIMultiObjects
    g = load_golden("kitti_objects.npz")
    for pid in FRAMES:
        calib = KITTICalibration(os.path.join(MINI, "training", "calib", pid + ".txt"))
        for k in ("P0", "P1", "P2", "P3", "R0", "V2C", "C2V", "I2V", "V2I"):
            assert np.array_equal(getattr(calib, k), g["%s.calib.%s" % (pid, k)]), (pid, k)
        intr = np.array([calib.cu, calib.cv, calib.fu, calib.fv, calib.tx, calib.ty], dtype=np.float64)
        assert np.array_equal(intr, g["%s.calib.intr" % pid])
        objs = KITTIMultiObjects.get_objects_from_label(os.path.join(MINI, "training", "label_2", pid + ".txt"), calib)
        assert len(objs) == int(g["%s.n" % pid])
        objs.convert_cam(src_cam=0, dst_cam=2)
        objs.convert_yaw(src_type="global", dst_type="local")
        for i, o in enumerate(objs):
            tag = "%s.obj%d." % (pid, i)
            assert o.cls_num == int(g[tag + "cls"])
            assert np.array_equal(np.array([o.occlusion, o.truncation, o.level], dtype=np.float64), g[tag + "occ_trunc_level"])
            assert np.array_equal(o.box2d, g[tag + "box2d"])
            box3d = np.concatenate([o.loc, o.dim, [o.ry]]).astype(np.float64)
            assert np.allclose(box3d, g[tag + "box3d"], rtol=2e-6, atol=2e-6), (tag, box3d, g[tag + "box3d"])
            assert np.allclose(o.projected_center, g[tag + "center"], rtol=1e-6, atol=1e-3), tag
            kp = o.projected_kpts
            if g[tag + "kpts"].shape[0] == 0:
                assert kp is None
            else:
                assert np.array_equal(kp[:, 2], g[tag + "kpts"][:, 2]), tag         # in-front flags
                assert np.allclose(kp[:, :2], g[tag + "kpts"][:, :2], rtol=1e-5, atol=2e-3), tag
        info = objs.original_objects.info_dict
        assert list(info["name"]) == [ln.split(" ")[0] for ln in open(os.path.join(MINI, "training", "label_2", pid + ".txt")) if ln.strip()]
        for k in ("truncated", "occluded", "alpha", "bbox", "dimensions", "score"):
            assert np.array_equal(np.asarray(info[k], dtype=np.float64), g["%s.info.%s" % (pid, k)]), k
        for k in ("location", "rotation_y"):            # reported in the current (camera 2 / local yaw) frame, as the reference does
            assert np.allclose(np.asarray(info[k], dtype=np.float64), g["%s.info.%s" % (pid, k)], rtol=2e-6, atol=2e-6), k


def test_dataset_samples_and_collate():
    """MonoConDataset on the mini tree: the sample / collate contract of SURVEY 8b, the filter rules (frame 000007 is built
    so that each rule rejects exactly one object and rows 0 and 5 survive), labels = the pinned per-object quantities,
    image = Normalize + Pad + ToTensor of the decoded PNG (bit-equal to the oracle's restatement)."""
    from PIL import Image
    from dataset.monocon_dataset import MonoConDataset
    from oracle import monocon_oracle as O
    g = load_golden("kitti_objects.npz")
    ds = MonoConDataset(MINI, "val")
    assert len(ds) == 2 and ds.file_prefix == list(FRAMES)
    s0 = ds[0]
    assert set(s0) == {"img", "img_metas", "calib", "label"}
    assert tuple(s0["img"].shape) == (3, 384, 1248) and s0["img"].dtype == torch.float32
    assert s0["img_metas"]["ori_shape"] == (375, 1242) and s0["img_metas"]["pad_shape"] == (384, 1248)
    assert s0["img_metas"]["sample_idx"] == 7 and s0["img_metas"]["split"] == "val"
    raw = np.asarray(Image.open(os.path.join(MINI, "training", "image_2", "000007.png")).convert("RGB"))
    ref_img, _ = O.preprocess(raw)
    assert torch.equal(s0["img"], ref_img)
    lab = s0["label"]
    assert all(v.dtype == torch.float32 and v.shape[0] == 1 and v.shape[1] == 30 for v in lab.values())
    assert lab["mask"][0].nonzero().flatten().tolist() == [0, 5]
    for row in (0, 5):
        tag = "000007.obj%d." % row
        assert np.array_equal(lab["gt_bboxes"][0, row].numpy(), g[tag + "box2d"])
        assert float(lab["gt_labels"][0, row]) == float(g[tag + "cls"]) == float(lab["gt_labels_3d"][0, row])
        assert np.allclose(lab["gt_bboxes_3d"][0, row].numpy(), g[tag + "box3d"], rtol=2e-6, atol=2e-6)
        assert np.allclose(lab["centers2d"][0, row].numpy(), g[tag + "center"][:2], atol=1e-3)
        assert np.allclose(float(lab["depths"][0, row]), g[tag + "center"][2], rtol=1e-6)
        assert np.allclose(lab["gt_kpts_2d"][0, row].numpy().reshape(9, 2), g[tag + "kpts"][:, :2], rtol=1e-5, atol=2e-3)
        kp = g[tag + "kpts"]
        inside = (kp[:, 0] >= 0) & (kp[:, 0] <= 1242) & (kp[:, 1] >= 0) & (kp[:, 1] <= 375)
        assert np.array_equal(lab["gt_kpts_valid_mask"][0, row].numpy(), np.where(inside, 2.0, kp[:, 2]))
    assert set(lab["gt_kpts_valid_mask"][0, 5].tolist()) == {1.0, 2.0}       # the truncated car: some corners off-frame
    for row in (1, 2, 3, 4, 6):                                              # rejected objects leave all-zero rows
        assert float(lab["gt_bboxes_3d"][0, row].abs().sum()) == 0.0 and float(lab["depths"][0, row]) == 0.0
    # frames of different size collate only after padding to a common shape: per-frame batches here
    b = MonoConDataset.collate_fn([s0, ds[0]])
    assert tuple(b["img"].shape) == (2, 3, 384, 1248) and b["label"]["gt_bboxes"].shape == (2, 30, 4)
    assert b["img_metas"]["sample_idx"] == [7, 7] and len(b["calib"]) == 2 and b["calib"][0].P2.shape == (3, 4)
    s1 = ds[1]
    assert tuple(s1["img"].shape) == (3, 384, 1248) and s1["label"]["mask"][0].nonzero().flatten().tolist() == [0, 1]
    with pytest.raises(ValueError):
        MonoConDataset(MINI, "val", filter_configs={"min_width": 3})


@pytest.mark.gpu
def test_detector_trains_and_evaluates_on_the_file_dataset(golden_sd, tmp_path):
    """the file dataset feeds the detector end to end on the GPU: a train step on a collated batch, batch_eval in KITTI
    format on the same frames, AP evaluation (engine/kitti_eval) and result files through the dataset"""
    import json
    from dataset.monocon_dataset import MonoConDataset
    from model import MonoConDetector
    from utils.engine_utils import move_data_device
    ds = MonoConDataset(MINI, "val")
    batch = move_data_device(MonoConDataset.collate_fn([ds[0], ds[1]]), torch.device("cuda"))
    m = MonoConDetector(34, pretrained_backbone=False)
    m.load_state_dict(golden_sd, strict=True)
    m = m.cuda().train()
    _, loss = m(batch)
    total = sum(loss.values())
    total.backward()
    assert bool(torch.isfinite(total))
    m.eval()
    with torch.no_grad():
        res = m.batch_eval(batch)
    assert set(res) == {"img_bbox", "img_bbox2d"} and len(res["img_bbox"]) == 2
    out_json = os.path.join(str(tmp_path), "ap.json")
    ap = ds.evaluate(res, save_path=out_json, verbose=False)
    # 3 classes x 3 difficulties x (strict, loose) + 3 overall, for 3D / BEV / 2D of the 3D set and 2D of the 2D set
    assert len(ap) == 4 * 21 and all(k.startswith(("img_bbox/KITTI/", "img_bbox2d/KITTI/")) for k in ap)
    assert all(0.0 <= v <= 100.0 for v in ap.values())                  # random weights: the values are ~0, but defined
    assert json.load(open(out_json)) == ap
    ds.write_kitti_results(res, str(tmp_path))
    assert os.path.isfile(os.path.join(str(tmp_path), "img_bbox", "000007.txt"))
