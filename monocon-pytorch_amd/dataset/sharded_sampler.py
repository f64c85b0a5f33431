"""Data-parallel sampler for the input pipeline.

The reference is single-GPU and uses ``DataLoader(shuffle=True)`` (``engine/monocon_engine.py:45-56``).  With one
process per GPU every rank must draw a disjoint shard of the same per-epoch permutation; this mirrors the
contract of ``torch.utils.data.DistributedSampler`` (same ``set_epoch`` / ``__len__`` behaviour) without needing an
initialised process group, so it can be unit-tested on the CPU and built before ``torch.distributed`` comes up.
"""
import math

import torch
from torch.utils.data import Sampler


class ShardedSampler(Sampler):
    """rank ``r`` of ``world`` gets indices ``perm[r::world]`` of the epoch's permutation (seed + epoch), padded by
    wrap-around (or truncated with ``drop_last``) so that every rank sees the same number of samples -- a
    requirement of the per-step gradient all-reduce."""

    def __init__(self, dataset_len, rank=0, world=1, shuffle=True, seed=0, drop_last=False):
        if world < 1 or not 0 <= rank < world:
            raise ValueError("bad rank/world %r/%r" % (rank, world))
        self.n = int(dataset_len)
        self.rank, self.world = int(rank), int(world)
        self.shuffle, self.seed, self.drop_last = bool(shuffle), int(seed), bool(drop_last)
        self.epoch = 0
        self.num_samples = self.n // self.world if drop_last else math.ceil(self.n / self.world)
        self.total = self.num_samples * self.world

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def __len__(self):
        return self.num_samples

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            idx = torch.randperm(self.n, generator=g).tolist()
        else:
            idx = list(range(self.n))
        if self.drop_last:
            idx = idx[:self.total]
        elif self.total > len(idx):
            pad = self.total - len(idx)
            idx += (idx * math.ceil(pad / max(len(idx), 1)))[:pad]
        return iter(idx[self.rank:self.total:self.world])
