"""SURVEY 8f-4, the train split's image work on the device: the loader's workers keep the random decisions, the labels and the
calibration of the reference's augmentations (transforms/default_transforms.py:52-373, geo_aware_transforms.py:14-418; the
train list of dataset/monocon_dataset.py:22-35) and ship the raw uint8 frame with 24 parameters; `mc_preprocess_augmented`
produces the float32 CHW frame that PhotometricDistortion -> RandomShift -> RandomHorizontalFlip -> RandomCrop3D -> Normalize
-> Pad -> ToTensor produce on the host, bit for bit."""
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MINI = os.path.join(GOLDEN, "kitti_mini")
SEEDS = list(range(40))


def _pair(seed, idx, split="train"):
    from dataset.monocon_dataset import MonoConDataset
    host = MonoConDataset(MINI, split, aug_rng=np.random.default_rng(seed))[idx]
    dev = MonoConDataset(MINI, split, aug_rng=np.random.default_rng(seed), device_image=True)[idx]
    return host, dev


def _interpret(frame, prm):
    """what the 24 parameters say, carried out with the host transforms' own numpy code"""
    from dataset.monocon_dataset import IMG_MEAN, IMG_STD
    from transforms import default_transforms as T
    from transforms.augmentations import PhotometricDistortion
    H, W, flags = int(prm[0]), int(prm[1]), int(prm[2])
    img = frame[:H, :W].numpy()
    if flags & T.AUG_COLOUR:
        p = {"brightness": prm[3] if flags & T.AUG_BRIGHTNESS else None, "contrast_before": prm[4] if flags & T.AUG_CONTRAST_BEFORE else None,
             "saturation": prm[5] if flags & T.AUG_SATURATION else None, "hue": prm[6] if flags & T.AUG_HUE else None,
             "contrast_after": prm[7] if flags & T.AUG_CONTRAST_AFTER else None,
             "permutation": prm[8:11].astype(np.int64) if flags & T.AUG_PERMUTATION else None}
        img = PhotometricDistortion.apply(img, p)
    if flags & T.AUG_SHIFT:
        sx, sy = int(prm[11]), int(prm[12])
        canvas = np.zeros_like(img)
        h, w = H - abs(sy), W - abs(sx)
        canvas[max(0, sy):max(0, sy) + h, max(0, sx):max(0, sx) + w] = img[max(0, -sy):max(0, -sy) + h, max(0, -sx):max(0, -sx) + w]
        img = canvas
    if flags & T.AUG_FLIP:
        img = img[:, ::-1, :]
    if flags & T.AUG_WINDOW:
        x0, y0, x1, y1 = (int(v) for v in prm[13:17])
        canvas = np.zeros_like(img)
        canvas[y0:y1, x0:x1] = img[y0:y1, x0:x1]
        img = canvas
    d = {"img": img, "img_metas": {}}
    for t in (T.Normalize(mean=IMG_MEAN, std=IMG_STD), T.Pad(32), T.ToTensor()):
        d = t(d)
    return d["img"]


def test_deferred_samples_carry_the_host_pipelines_decisions():
    """same Generator seed: the deferred sample has the host sample's labels, calibration and metas; its frame is the decoded
    PNG, zero-padded; its 24 parameters, carried out with the host transforms' numpy code, give the host sample's image bit
    for bit.  Over 40 seeds every operation and most of their combinations occur."""
    from PIL import Image
    from transforms import default_transforms as T
    seen = 0
    for seed in SEEDS:
        for idx in (0, 1):
            host, dev = _pair(seed, idx)
            assert dev["img"].dtype == torch.uint8 and tuple(dev["img"].shape) == (384, 1248, 3)
            assert dev["img_aug"].dtype == torch.float32 and tuple(dev["img_aug"].shape) == (T.AUG_PARAMS,)
            assert host["label"].keys() == dev["label"].keys()
            for k in host["label"]:
                assert torch.equal(host["label"][k], dev["label"][k]), (seed, idx, k)
            assert np.array_equal(host["calib"].P2, dev["calib"].P2)
            assert host["img_metas"] == dev["img_metas"]
            prm = dev["img_aug"].numpy()
            raw = np.asarray(Image.open(os.path.join(MINI, "training", "image_2", "%06d.png" % host["img_metas"]["sample_idx"])).convert("RGB"))
            H, W = raw.shape[:2]
            assert (int(prm[0]), int(prm[1])) == (H, W) and np.array_equal(dev["img"][:H, :W].numpy(), raw)
            assert int(dev["img"][H:].sum()) == 0 and int(dev["img"][:, W:].sum()) == 0
            assert torch.equal(_interpret(dev["img"], prm), host["img"]), (seed, idx, int(prm[2]))
            seen |= int(prm[2])
    assert seen == 1023                              # every flag occurred
    host, dev = _pair(0, 0, "val")
    assert int(dev["img_aug"][2]) == 0 and torch.equal(_interpret(dev["img"], dev["img_aug"].numpy()), host["img"])


def test_deferred_image_rejects_what_the_kernel_cannot_compose():
    from dataset.monocon_dataset import MonoConDataset
    from transforms import DeferImage, DeferredImage, RandomHorizontalFlip, RandomShift, Normalize
    rng = np.random.default_rng(1)
    wrong_order = [DeferImage(), RandomHorizontalFlip(prob=1.0, rng=rng), RandomShift(prob=1.0, rng=rng), DeferredImage()]
    with pytest.raises(NotImplementedError):
        MonoConDataset(MINI, "train", transforms=wrong_order)[0]
    with pytest.raises(TypeError):                   # a host transform has already turned the frame into floats
        MonoConDataset(MINI, "train", transforms=[DeferImage(), Normalize([0, 0, 0], [1, 1, 1]), DeferredImage()])[0]
    b = MonoConDataset(MINI, "val", device_image=True)
    batch = b.collate_fn([b[0], b[1]])
    assert tuple(batch["img"].shape) == (2, 384, 1248, 3) and tuple(batch["img_aug"].shape) == (2, 24)


@pytest.mark.gpu
def test_device_augmentation_is_bit_identical_to_the_host_pipeline():
    """mc_preprocess_augmented on the raw frames + parameters of the deferred samples == the image of the host pipeline's
    sample, every float32 bit, for 40 seeds x 2 frames (all ten flags occur) and for the validation list (no flags: the
    plain Normalize + Pad + ToTensor of mc_preprocess)"""
    from hipmonocon.engine import Engine
    eng = Engine()
    frames, params, want = [], [], []
    for seed in SEEDS:
        for idx in (0, 1):
            host, dev = _pair(seed, idx)
            frames.append(dev["img"]); params.append(dev["img_aug"]); want.append(host["img"])
    host, dev = _pair(0, 1, "val")
    frames.append(dev["img"]); params.append(dev["img_aug"]); want.append(host["img"])
    for lo in range(0, len(frames), 27):             # three launches of up to 27 frames
        f = torch.stack(frames[lo:lo + 27]).cuda()
        p = torch.stack(params[lo:lo + 27]).cuda()
        got = eng.preprocess_augmented(f, p).cpu()
        for k in range(got.shape[0]):
            w = want[lo + k]
            same = torch.equal(got[k].view(torch.int32), w.view(torch.int32))
            assert same, (lo + k, int(params[lo + k][2]), float((got[k] - w).abs().max()), int((got[k] != w).sum()))


@pytest.mark.gpu
def test_detector_steps_on_deferred_batches_as_on_host_batches(golden_sd):
    """MonoConDetector.forward finishes a collated batch of deferred samples itself (finish_batch -> mc_preprocess_augmented):
    the train step on it gives the losses and gradients of the step on the host pipeline's batch, bit for bit; the batches
    come through RingLoader's uint8 ring + DevicePrefetcher on one side, a plain DataLoader on the other"""
    from torch.utils.data import DataLoader
    from dataset.monocon_dataset import MonoConDataset
    from hipmonocon.feed import DevicePrefetcher, RingLoader
    from model import MonoConDetector
    res = []
    for device_image in (False, True):
        ds = MonoConDataset(MINI, "train", aug_rng=np.random.default_rng(17), device_image=device_image)
        m = MonoConDetector(34, pretrained_backbone=False)
        m.load_state_dict(golden_sd, strict=True)
        m = m.cuda().train()
        if device_image:
            ds._aug_rng_worker = 0               # (the test wants the parent's random stream in the worker: no per-worker re-seed)
            rl = RingLoader(ds, batch_size=2, num_workers=1, collate_fn=ds.collate_fn, image_shape=(384, 1248, 3),
                            image_dtype=torch.uint8)       # (given: probing a sample would draw from the Generator)
            assert rl.ring.dtype == torch.uint8 and tuple(rl.ring.shape[1:]) == (2, 384, 1248, 3)
            batch = next(iter(DevicePrefetcher(rl, "cuda:0", m)))
            assert batch["img"].dtype == torch.uint8 and tuple(batch["img_aug"].shape) == (2, 24)
        else:
            host = next(iter(DataLoader(ds, batch_size=2, collate_fn=ds.collate_fn)))
            batch = dict(host, img=host["img"].cuda(), label={k: v.cuda() for k, v in host["label"].items()})
        _, loss = m(batch)
        assert batch["img"].dtype == torch.float32 and tuple(batch["img"].shape) == (2, 3, 384, 1248) and "img_aug" not in batch
        sum(loss.values()).backward()
        res.append(({k: float(v.detach()) for k, v in loss.items()}, {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None},
                    batch["img"].clone()))
    assert torch.equal(res[0][2], res[1][2])
    assert res[0][0] == res[1][0]
    assert all(torch.equal(res[0][1][n], res[1][1][n]) for n in res[0][1])
