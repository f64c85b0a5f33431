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
