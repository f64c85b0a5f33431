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


def test_host_side_label_check_agrees_with_the_device_rule():
    """hipmonocon.train.labels_ok_on_host: the test _require_objects runs on the device (utils/target_generator.py:70-75,
    losses/l1_loss.py:15 in the reference: an index error / an assertion), on tensors that have not left the host"""
    from hipmonocon.train import labels_ok_on_host
    H, W = 128, 224
    lab = synth.make_batch(5, 3, H, W)["label"]
    assert labels_ok_on_host(lab, (H, W))
    empty = {k: v.clone() for k, v in lab.items()}
    empty["mask"].zero_()
    assert not labels_ok_on_host(empty, (H, W))                         # no valid object in the batch
    b, o = [int(i[0]) for i in torch.nonzero(lab["mask"], as_tuple=True)]
    outside = {k: v.clone() for k, v in lab.items()}
    outside["gt_bboxes"][b, o] = torch.tensor([W + 8.0, 10.0, W + 40.0, 40.0])     # centre right of the map
    assert not labels_ok_on_host(outside, (H, W))
    edge = {k: v.clone() for k, v in lab.items()}
    edge["gt_bboxes"][b, o] = torch.tensor([W - 8.0, H - 8.0, W - 0.5, H - 0.5])   # centre in the last cell: inside
    assert labels_ok_on_host(edge, (H, W))
    cls = {k: v.clone() for k, v in lab.items()}
    cls["gt_labels"][b, o] = 3
    assert not labels_ok_on_host(cls, (H, W))
    cls["gt_labels"][b, o] = -1
    assert not labels_ok_on_host(cls, (H, W))
    ignored = {k: v.clone() for k, v in lab.items()}
    slot = [int(i[0]) for i in torch.nonzero(lab["mask"] == 0, as_tuple=True)]
    ignored["gt_labels"][slot[0], slot[1]] = 7                          # an unused slot may hold anything
    assert labels_ok_on_host(ignored, (H, W))


def test_feed_helpers_without_a_device():
    """DevicePrefetcher passes batches through when there is no HIP device; DeferredScalars keeps order and `keep`"""
    from hipmonocon.feed import DeferredScalars, DevicePrefetcher
    batches = [{"img": torch.full((1, 3, 32, 32), float(i)), "label": {"mask": torch.ones(1, 2)}, "img_metas": {"idx": [i]}}
               for i in range(4)]
    got = list(DevicePrefetcher(batches, "cpu"))
    assert [int(b["img"][0, 0, 0, 0]) for b in got] == [0, 1, 2, 3] and all(a is b for a, b in zip(got, batches))
    assert len(DevicePrefetcher(batches, None)) == 4
    d = DeferredScalars()
    out = []
    for i in range(5):
        d.push(torch.tensor(float(i)))
        out += d.ready(keep=1)
        assert out == [float(j) for j in range(i)]
    assert d.ready(0) == [4.0] and d.ready(0) == []


@pytest.mark.gpu
def test_prefetcher_hands_over_the_loader_batches_validated():
    """the batches of a pinned DataLoader arrive on the device in order and equal to the host copies; a batch that passes the
    host-side label check is marked on the detector (its forward reads no verdict back), one that fails is not -- and the
    forward then raises as the reference does (target_generator.py:70-75)"""
    from torch.utils.data import DataLoader
    from dataset.synthetic_dataset import SyntheticMonoConDataset
    from hipmonocon.feed import DeferredScalars, DevicePrefetcher
    from model import MonoConDetector
    ds = SyntheticMonoConDataset(length=6, height=96, width=224, seed=3)
    host = list(DataLoader(ds, batch_size=2, collate_fn=ds.collate_fn))
    m = MonoConDetector(34, pretrained_backbone=False).cuda().train()
    loader = DataLoader(ds, batch_size=2, collate_fn=ds.collate_fn, pin_memory=True)
    n = 0
    for dev, ref in zip(DevicePrefetcher(loader, "cuda:0", m), host):
        assert dev["img"].is_cuda and torch.equal(dev["img"].cpu(), ref["img"])
        for k, v in ref["label"].items():
            assert dev["label"][k].is_cuda and torch.equal(dev["label"][k].cpu(), v), k
        seen = m._mask_validated
        assert seen[0]() is dev["label"]["mask"]
        _, loss = m(dev)
        assert all(torch.isfinite(v) for v in loss.values())
        n += 1
    assert n == 3

    class Bad(torch.utils.data.Dataset):
        def __len__(self):
            return 2

        def __getitem__(self, i):
            d = ds[i]
            d["label"]["gt_labels"] = torch.full_like(d["label"]["gt_labels"], 5.0)
            return d
    object.__setattr__(m, "_mask_validated", None)
    it = iter(DevicePrefetcher(DataLoader(Bad(), batch_size=2, collate_fn=ds.collate_fn), "cuda:0", m))
    dev = next(it)
    assert m._mask_validated is None
    with pytest.raises(IndexError):
        m(dev)
    d = DeferredScalars()
    vals = [torch.tensor(float(i), device="cuda") * 2 for i in range(4)]
    got = []
    for v in vals:
        d.push(v)
        got += d.ready(1)
    assert got + d.ready(0) == [0.0, 2.0, 4.0, 6.0]


def _same_batch(a, b):
    assert torch.equal(a["img"], b["img"])
    assert a["label"].keys() == b["label"].keys() and all(torch.equal(a["label"][k], b["label"][k]) for k in a["label"])
    assert a["img_metas"] == b["img_metas"]


def test_ring_loader_yields_the_batches_of_a_dataloader():
    """hipmonocon.feed.RingLoader: the workers write the frames into a shared ring of batch slots, only the labels travel
    through their queues -- same batches, same order, same collated dict as the DataLoader the reference builds
    (engine/monocon_engine.py:60-72), over two epochs, a ragged last batch and several trips around the ring"""
    from torch.utils.data import DataLoader
    from dataset.synthetic_dataset import SyntheticMonoConDataset
    from hipmonocon.feed import RingLoader
    ds = SyntheticMonoConDataset(length=23, height=32, width=64, seed=4)
    ref = list(DataLoader(ds, batch_size=3, collate_fn=ds.collate_fn))
    rl = RingLoader(ds, batch_size=3, num_workers=2, prefetch_factor=1, pin=False)
    assert rl.nslots == 5 and len(rl) == len(ref) == 8 and not rl.pinned
    for epoch in range(2):
        got = list(rl)
        assert len(got) == 8 and tuple(got[-1]["img"].shape) == (2, 3, 32, 64)
        for a, b in zip(got, ref):
            _same_batch(a, b)
    g = torch.Generator().manual_seed(11)
    order = [b["img_metas"]["sample_idx"] for b in RingLoader(ds, 4, 2, shuffle=True, drop_last=True, pin=False, generator=g)]
    assert len(order) == 5 and sorted(i for b in order for i in b) != [i for b in order for i in b]
    assert len({i for b in order for i in b}) == 20
    # a consumer that uploads out of the ring itself sees views of the slots, and reports its uploads
    class Ev:
        def __init__(self, log, k):
            self.log, self.k = log, k

        def synchronize(self):
            self.log.append(self.k)
    log = []
    for k, (a, b) in enumerate(zip(rl.host_batches(), ref)):
        assert a["img"].untyped_storage().data_ptr() == rl.ring.untyped_storage().data_ptr()
        _same_batch(a, b)
        rl.note_upload(Ev(log, k))
    assert log == [0, 1, 2, 3, 4]          # batch k is asked for only after the upload of batch k - 3 has completed
    with pytest.raises(ValueError):
        RingLoader(ds, 3, 0)
    # frames of a shape the ring was not built for travel with their samples, as under a DataLoader
    small = RingLoader(ds, batch_size=3, num_workers=1, pin=False, image_shape=(3, 16, 16))
    for a, b in zip(small, ref[:2]):
        _same_batch(a, b)


@pytest.mark.gpu
def test_ring_loader_uploads_out_of_its_pinned_ring():
    """under a DevicePrefetcher the frames go from the page-locked ring to the device in one asynchronous copy; the batches
    are those of the DataLoader, over more batches than the ring has slots"""
    from torch.utils.data import DataLoader
    from dataset.synthetic_dataset import SyntheticMonoConDataset
    from hipmonocon.feed import DevicePrefetcher, RingLoader
    ds = SyntheticMonoConDataset(length=30, height=64, width=96, seed=9)
    ref = list(DataLoader(ds, batch_size=2, collate_fn=ds.collate_fn))
    rl = RingLoader(ds, batch_size=2, num_workers=2, prefetch_factor=1)
    assert rl.pinned and rl.ring.is_pinned() and rl.nslots == 5
    for epoch in range(2):
        n = 0
        for dev, host in zip(DevicePrefetcher(rl, "cuda:0"), ref):
            assert dev["img"].is_cuda and torch.equal(dev["img"].cpu(), host["img"])
            assert all(torch.equal(dev["label"][k].cpu(), host["label"][k]) for k in host["label"])
            n += 1
        assert n == 15
    rl.close()
    assert not rl.pinned


def test_ring_loader_over_the_file_dataset():
    """the KITTI file dataset through RingLoader's fork-server workers (the dataset, its transforms and its collate are
    pickled for them): the batch of the DataLoader the reference builds (monocon_engine.py:60-72), and the 'train' split's
    augmented frames keep the padded shape the ring is built for (transforms/augmentations.py: crops are pasted back)"""
    from torch.utils.data import DataLoader
    from dataset.monocon_dataset import MonoConDataset
    from hipmonocon.feed import RingLoader
    ds = MonoConDataset(MINI, "val")
    ref = list(DataLoader(ds, batch_size=2, collate_fn=ds.collate_fn))
    rl = RingLoader(ds, batch_size=2, num_workers=1, pin=False)
    assert tuple(rl.ring.shape) == (5, 2, 3, 384, 1248)
    got = list(rl)
    assert len(got) == len(ref) == 1
    _same_batch(got[0], ref[0])
    assert [c.P2.tolist() for c in got[0]["calib"]] == [c.P2.tolist() for c in ref[0]["calib"]]
    tr = MonoConDataset(MINI, "train")
    for b in RingLoader(tr, batch_size=2, num_workers=1, pin=False):
        assert tuple(b["img"].shape) == (2, 3, 384, 1248) and torch.isfinite(b["img"]).all()
        assert b["label"]["mask"].shape == (2, 30)


class _TaggedFrames(torch.utils.data.Dataset):
    """frames filled with their own sample index (module level: pickled for the fork-server workers)"""

    def __init__(self, n, shape):
        self.n, self.shape = n, shape

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return {"img": torch.full(self.shape, float(i)), "label": {"mask": torch.ones(1)}, "img_metas": {"sample_idx": i}}

    @staticmethod
    def collate_fn(b):
        return {"img": torch.stack([d["img"] for d in b]), "label": {"mask": torch.stack([d["label"]["mask"] for d in b])},
                "img_metas": {"sample_idx": [d["img_metas"]["sample_idx"] for d in b]}}


@pytest.mark.gpu
def test_ring_slots_are_not_refilled_before_their_upload():
    """600 batches through an 11-slot ring with workers that are much faster than the uploads, which queue behind other
    traffic on the copy stream while the consumer never waits for the device: every frame that arrives on the device still holds its own sample index (a slot refilled before its
    upload had completed would carry the index of a later batch)"""
    from hipmonocon.feed import DevicePrefetcher, RingLoader
    ds = _TaggedFrames(600 * 8, (3, 96, 512))
    rl = RingLoader(ds, batch_size=8, num_workers=4, shuffle=True, collate_fn=ds.collate_fn)
    assert rl.nslots == 11 and rl.pinned
    pf = DevicePrefetcher(rl, "cuda:0")
    ballast_host = torch.empty(256 << 20, dtype=torch.uint8).pin_memory()
    ballast_dev = torch.empty_like(ballast_host, device="cuda")
    seen, bad = 0, torch.zeros((), dtype=torch.int64, device="cuda")
    for k, b in enumerate(pf):
        with torch.cuda.stream(pf.copy_stream):            # the next upload queues behind 256 MB of other traffic ...
            ballast_dev.copy_(ballast_host, non_blocking=True)
        want = torch.tensor(b["img_metas"]["sample_idx"], dtype=torch.float32).cuda(non_blocking=True)
        bad += (b["img"][:, 0, 0, 0] != want).sum() + (b["img"][:, -1, -1, -1] != want).sum()     # ... and nothing here waits
        seen += len(want)
    assert seen == 4800 and int(bad) == 0          # (scratch/ring_stress_broken.py: with the events ignored this loop does see bad frames)
    rl.close()


def test_ring_loader_epochs_follow_the_distributed_sampler():
    """one iterator serves all epochs (the workers fill slots for epoch e + 1 while e is consumed), so RingLoader moves a
    sampler with set_epoch on by itself: the batches of epoch e are those of DataLoader + DistributedSampler.set_epoch(e), the
    engine's call at the start of each epoch (monocon_engine.py, train_one_epoch) is a no-op, a different epoch (a resume, a
    replay) or an epoch abandoned half-way starts the loader over"""
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler
    from dataset.synthetic_dataset import SyntheticMonoConDataset
    from hipmonocon.feed import RingLoader
    ds = SyntheticMonoConDataset(length=13, height=32, width=64, seed=6)

    def reference(epoch):
        sp = DistributedSampler(ds, num_replicas=2, rank=1, shuffle=True, drop_last=True, seed=3)
        sp.set_epoch(epoch)
        return [b["img_metas"]["sample_idx"] for b in DataLoader(ds, batch_size=2, sampler=sp, collate_fn=ds.collate_fn)]

    sp = DistributedSampler(ds, num_replicas=2, rank=1, shuffle=True, drop_last=True, seed=3)
    rl = RingLoader(ds, batch_size=2, num_workers=2, sampler=sp, pin=False)
    ids = lambda: [b["img_metas"]["sample_idx"] for b in rl]
    for epoch in (4, 5, 6):                      # the engine's sequence
        rl.sampler.set_epoch(epoch)
        assert ids() == reference(epoch) and len(reference(epoch)) == 3
    assert ids() == reference(7)                 # ... and without the call
    rl.sampler.set_epoch(2)                      # a replay
    assert ids() == reference(2)
    it = iter(rl)
    next(it)                                     # an epoch left after one batch
    del it
    rl.sampler.set_epoch(3)
    assert ids() == reference(3)
    assert reference(4) != reference(5)
