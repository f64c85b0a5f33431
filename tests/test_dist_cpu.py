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


def _sync_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hipmonocon import dist as hdist
    torch.manual_seed(1000 + rank)                       # every rank draws its own weights ...
    m = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.BatchNorm2d(8), torch.nn.Linear(4, 4))
    m[1].num_batches_tracked += rank                     # ... and has its own (int64) buffer state
    before = hdist.state_checksum(m)
    n = hdist.sync_module_state(m)                       # ... until rank 0's state is broadcast
    seed = hdist.broadcast_seed(4242 + rank)
    ok_all = hdist.all_ranks_ok(rank != 1)               # rank 1 reports a bad batch: every rank must see False
    ok_none = hdist.all_ranks_ok(True)
    out[rank] = (before, hdist.state_checksum(m), n, seed, ok_all, ok_none, int(m[1].num_batches_tracked))
    dist.barrier()
    dist.destroy_process_group()


def test_replicas_start_identical_and_fail_together():
    """ADVICE (round 1, high): with a per-rank random seed every replica used to initialise differently and nothing
    synchronised them.  ``sync_module_state`` makes every rank adopt rank 0's parameters and buffers, ``broadcast_seed``
    its seed, and ``all_ranks_ok`` turns one rank's invalid batch into an exception on all ranks."""
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_sync_worker, args=(world, port, out), nprocs=world, join=True)
    (b0, a0, n0, s0, f0, t0, nbt0), (b1, a1, n1, s1, f1, t1, nbt1) = out[0], out[1]
    assert b0 != b1                                      # different before
    assert a0 == a1 == b0                                # rank 0's state everywhere afterwards
    assert n0 == n1 == 9 and nbt0 == nbt1 == 0
    assert s0 == s1 == 4242
    assert f0 is False and f1 is False and t0 is True and t1 is True
