"""Data-parallel host logic on CPU with the gloo backend, world_size 2: the flat
gradient buffer, the one-collective mean all-reduce and the batch sharding.  Expectation:
N-rank result == mean of the per-shard gradients."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hipmonocon import netspec, synth
from hipmonocon.dist import FlatGrads, shard_batch, shard_indices


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shapes = [("a", (64, 32, 3, 3)), ("b", (7,)), ("c", (10, 64)), ("d", (1,))]
    named = [(n, torch.empty(s)) for n, s in shapes]
    fg = FlatGrads(named, torch.device("cpu"))
    for i, (n, _) in enumerate(named):
        fg.views[n].copy_(torch.from_numpy(synth.normalish(100 + rank, n, named[i][1].shape).astype(np.float32)))
    fg.allreduce_mean()
    out[rank] = {n: fg.views[n].clone() for n, _ in named}
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_mean_equals_mean_of_shard_gradients():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    for n, shape in [("a", (64, 32, 3, 3)), ("b", (7,)), ("c", (10, 64)), ("d", (1,))]:
        expect = sum(torch.from_numpy(synth.normalish(100 + r, n, shape).astype(np.float32)) for r in range(world)) / world
        for r in range(world):
            assert torch.allclose(out[r][n], expect, atol=1e-7), (n, r)
        assert torch.equal(out[0][n], out[1][n])            # every rank ends with identical gradients


def test_flat_views_alias_one_buffer_and_are_16B_aligned():
    spec = [(k, torch.empty(shape)) for k, shape, dt, role in netspec.state_fields()
            if role == "param" and k not in netspec.DEAD_PARAMS]
    fg = FlatGrads(spec, torch.device("cpu"))
    assert fg.numel == 19578533                                # 19 620 261 parameters minus the 41 728 dead ones
    base = fg.flat.data_ptr()
    for n, p in spec:
        v = fg.views[n]
        assert v.shape == p.shape and (v.data_ptr() - base) % 16 == 0
    fg.views[spec[3][0]].fill_(2.0)
    assert float(fg.flat.sum()) == 2.0 * spec[3][1].numel()


def test_shard_batch_partitions_the_global_batch():
    b = synth.make_batch(3, 6, 64, 64)
    seen = []
    for r in range(4):
        sb = shard_batch(b, r, 4)
        idx = shard_indices(6, r, 4)
        assert sb["img"].shape[0] == len(idx) == len(sb["calib"]) == len(sb["img_metas"]["pad_shape"])
        assert torch.equal(sb["img"], b["img"][idx[0]:idx[-1] + 1])
        assert torch.equal(sb["label"]["mask"], b["label"]["mask"][idx[0]:idx[-1] + 1])
        seen += idx
    assert seen == list(range(6))
