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


# ----------------------------------------------------------------------------- world 8 (the node the driver benches on)
class _FakeEngine:
    """stands in for hipmonocon.engine.Engine in the start-up protocol tests (no GPU here): records what the protocol
    asks of it"""

    def __init__(self, rank, fail_uid=False, fail_init_on=()):
        self.rank, self.device = rank, torch.device("cpu")
        self.comm_world = 0
        self.fail_uid, self.fail_init_on = fail_uid, set(fail_init_on)
        self.built, self.imported, self.inited = [], None, None
        self._table = [10, 32, 96, 320, 3, 1, 64, 64, 1, 6, 64, 4,  11, 32, 48, 160, 1, 1, 128, 128, 2, 6, 128, 128, 1]

    def comm_unique_id(self):
        if self.fail_uid:
            raise RuntimeError("librccl.so not found")
        return bytes((7 * i + 3) % 251 for i in range(128))

    def comm_init(self, rank, world, uid):
        if rank in self.fail_init_on:
            raise RuntimeError("mc_comm_init: rank %d of %d still waits in ncclCommInitRank" % (rank, world))
        self.inited = (rank, world, uid)
        self.comm_world = world

    def comm_destroy(self):
        self.comm_world = 0

    def build_train_plan(self, B, H, W):
        self.built.append((B, H, W))

    def tune_export(self):
        return list(self._table)

    def tune_import(self, table):
        self.imported = list(table)
        return 2


def _world8_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), MONOCON_HIP_DP="rccl")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hipmonocon import dist as hdist
    res = {}
    # (1) rank-0 state broadcast over 8 ranks
    torch.manual_seed(2000 + rank)
    m = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.BatchNorm2d(8))
    m[1].num_batches_tracked += rank
    hdist.sync_module_state(m)
    res["checksum"] = hdist.state_checksum(m)
    # (2) the bucket schedule: four dense buckets of the flat gradient buffer, exchanged in backward order, each
    #     averaged over the ranks -- the same partition csrc/mc_comm.hip makes (mc_grad_bucket_of), on gloo
    spec = [("head.a", (64, 9)), ("neck.b", (33,)), ("backbone.level5.c", (128, 5)), ("backbone.level4.d", (77,)),
            ("backbone.level2.e", (16, 3)), ("backbone.base_layer.f", (5,))]
    fg = FlatGrads([(n, torch.empty(s)) for n, s in spec], torch.device("cpu"))
    for n, s in spec:
        fg.views[n].copy_(torch.from_numpy(synth.normalish(300 + rank, n, s).astype(np.float32)))

    def bucket_of(name):
        if name.startswith(("head.", "neck.")):
            return 0
        if name.startswith("backbone.level5"):
            return 1
        if name.startswith("backbone.level4"):
            return 2
        return 3
    order = []
    for b in range(4):
        names = [n for n, _ in spec if bucket_of(n) == b]
        lo = min(fg.views[n].data_ptr() for n in names)
        hi = max(fg.views[n].data_ptr() + fg.views[n].numel() * 4 for n in names)
        i0, i1 = (lo - fg.flat.data_ptr()) // 4, (hi - fg.flat.data_ptr()) // 4
        seg = fg.flat[i0:i1]                               # a dense range: ONE collective per bucket
        dist.all_reduce(seg, op=dist.ReduceOp.SUM)
        seg.mul_(1.0 / world)
        order.append((b, int(i0), int(i1)))
    res["buckets"] = order
    res["grads"] = {n: fg.views[n].clone() for n, _ in spec}
    # (3) tune table: rank 0 builds + exports, everyone else imports the same ints
    eng = _FakeEngine(rank)
    res["n_tune"] = hdist.share_tune_table(eng, 32, 384, 1280)
    res["built"], res["imported"] = eng.built, eng.imported
    # (4) communicator start-up: every rank adopts rank 0's id ...
    assert hdist.ensure_engine_comm(eng) is True
    res["inited"] = eng.inited
    # (5) ... and when rank 0 cannot create the id (it used to skip the broadcast and hang the others), or
    #     one rank's init times out, ALL ranks fall back together
    os.environ["MONOCON_HIP_DP_FALLBACK"] = "1"
    e2 = _FakeEngine(rank, fail_uid=(rank == 0))
    res["fallback_uid"] = hdist.ensure_engine_comm(e2)
    os.environ["MONOCON_HIP_DP"] = "rccl"
    e3 = _FakeEngine(rank, fail_init_on={5})
    res["fallback_init"] = (hdist.ensure_engine_comm(e3), e3.comm_world)
    res["fallback_reason"] = getattr(e3, "comm_fallback_reason", None)       # bench.py prints it as comm_path
    os.environ["MONOCON_HIP_DP"] = "rccl"
    os.environ["MONOCON_HIP_DP_FALLBACK"] = "0"
    e4 = _FakeEngine(rank, fail_init_on={3})
    try:
        hdist.ensure_engine_comm(e4)
        res["hard"] = "no error"
    except RuntimeError as e:
        res["hard"] = str(e)[:40]
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_world8_start_up_protocol_and_bucket_schedule():
    """the first real 8-GPU run is a single shot.  Eight ranks over gloo: rank-0 state broadcast, the
    four-bucket exchange == the mean of the per-rank gradients, one tune table on every rank, the communicator id on
    every rank, and a failure on ONE rank (id creation on rank 0, init time-out on rank 5 / 3) turning into the SAME
    outcome on all eight -- fallback when allowed, an exception when not -- instead of a hang."""
    world, port = 8, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_world8_worker, args=(world, port, out), nprocs=world, join=True)
    r0 = out[0]
    spec = [("head.a", (64, 9)), ("neck.b", (33,)), ("backbone.level5.c", (128, 5)), ("backbone.level4.d", (77,)),
            ("backbone.level2.e", (16, 3)), ("backbone.base_layer.f", (5,))]
    for r in range(world):
        o = out[r]
        assert o["checksum"] == r0["checksum"]
        assert o["buckets"] == r0["buckets"] and [b for b, _, _ in o["buckets"]] == [0, 1, 2, 3]
        for n, s in spec:
            expect = sum(torch.from_numpy(synth.normalish(300 + q, n, s).astype(np.float32)) for q in range(world)) / world
            assert torch.allclose(o["grads"][n], expect, atol=1e-6), (r, n)
            assert torch.equal(o["grads"][n], r0["grads"][n])
        assert o["n_tune"] == 25
        assert (o["built"] == [(32, 384, 1280)]) == (r == 0) and (o["imported"] is None) == (r == 0)
        if r:
            assert o["imported"] == _FakeEngine(0).tune_export()
        assert o["inited"] == (r, world, _FakeEngine(0).comm_unique_id())
        assert o["fallback_uid"] is False
        assert o["fallback_init"] == (False, 0)
        assert o["fallback_reason"]             # every rank can name why (its own error, or "another rank ...")
        assert o["hard"].startswith("could not build the handle's RCCL")
    # bucket ranges are disjoint and cover the buffer in order
    rng = r0["buckets"]
    assert all(rng[i][2] <= rng[i + 1][1] for i in range(3))


# ----------------------------------------------------------------------------- rank-local state must not gate collectives
class _FailingEngine(_FakeEngine):
    def __init__(self, rank, fail_build=False, fail_import_on=()):
        super().__init__(rank)
        self.fail_build, self.fail_import_on = fail_build, set(fail_import_on)

    def build_train_plan(self, B, H, W):
        if self.fail_build:
            raise MemoryError("train plan: out of device memory (injected)")
        super().build_train_plan(B, H, W)

    def tune_import(self, table):
        if self.rank in self.fail_import_on:
            raise ValueError("mc_tune_import: unknown shape id 99 (injected)")
        return super().tune_import(table)


class _Det:
    """stands in for the detector in _require_objects (it only carries the validated-mask note)"""


def _asym_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MONOCON_HIP_DP="rccl")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hipmonocon import dist as hdist, train as htrain
    res = {}
    # (1) the source rank cannot build its plan: every rank raises, with the SOURCE's message
    try:
        hdist.share_tune_table(_FailingEngine(rank, fail_build=(rank == 0)), 32, 384, 1280)
        res["build"] = "no error"
    except RuntimeError as e:
        res["build"] = str(e)
    # (2) ONE rank cannot import the table: every rank raises (the others are not left in the next collective)
    try:
        hdist.share_tune_table(_FailingEngine(rank, fail_import_on={2}), 32, 384, 1280)
        res["import"] = "no error"
    except RuntimeError as e:
        res["import"] = str(e)[:60]
    # (3) set-up runs once per engine, whatever shapes come later; a later shape issues NO collective
    eng = _FakeEngine(rank)
    hdist.setup_engine_dp(eng, 32, 384, 1280)
    hdist.setup_engine_dp(eng, 7 + rank, 384, 1280)          # rank-dependent "uneven last batch": nothing to mis-pair
    res["setup"] = (eng.built, eng.comm_world, getattr(eng, "_dp_ready", False))
    # (4) label validation with DIFFERENT cache states per rank: rank 0 re-uses its (validated) label tensors, the others
    #     feed fresh ones -- all return, nobody waits in a collective the others skipped; then a bad label on ONE rank
    #     raises on ALL of them
    det = _Det()
    lab = {"mask": torch.ones(2, 30), "gt_bboxes": torch.tensor([10.0, 10.0, 40.0, 40.0]).repeat(2, 30, 1),
           "gt_labels": torch.zeros(2, 30)}
    htrain._require_objects(det, lab, (64, 128))             # step 1: everybody validates
    for step in range(3):
        if rank != 0:                                         # fresh tensors on the other ranks, the same objects on rank 0
            lab = {k: v.clone() for k, v in lab.items()}
        htrain._require_objects(det, lab, (64, 128))
    bad = {k: v.clone() for k, v in lab.items()}
    if rank == 1:
        bad["gt_labels"][0, 0] = 9.0
    try:
        htrain._require_objects(det, bad if rank == 1 else lab, (64, 128))
        res["labels"] = "no error"
    except IndexError:
        res["labels"] = "IndexError"
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_collectives_of_the_dp_start_up_and_label_vote_do_not_depend_on_rank_local_state():
    world, port = 4, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_asym_worker, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        o = out[r]
        assert "rank 0 could not build" in o["build"] and "out of device memory (injected)" in o["build"], o["build"]
        assert o["import"].startswith("tune table of rank 0 could not be imported"), o["import"]
        assert o["setup"] == ([(32, 384, 1280)] if r == 0 else [], world, True), o["setup"]
        assert o["labels"] == "IndexError"


def _host_group_worker(rank, world, port, out):
    """the host-side vote group is cached per DEFAULT process group; after destroy + re-init the cache entry of the
    old group must not be served, and every rank must take the same decision about using it.  (On gloo there is no second
    group to make -- the decision is "none" -- but the cache bookkeeping and the agreement are the same code.)"""
    from hipmonocon import dist as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    res = []
    for k in range(2):
        os.environ["MASTER_PORT"] = str(port + k)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        g = D.host_group()
        tok = D._HOST_GROUP[0]
        res.append((g is None, tok is D._default_group_token(), D.host_votes_are_cheap(),
                    D.all_ranks_ok_many((True, rank != 1), host=True)))
        dist.destroy_process_group()
    assert D._HOST_GROUP[0] is not None
    out[rank] = res


def test_host_group_cache_follows_the_default_process_group():
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_host_group_worker, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        assert out[r] == [(True, True, True, [True, False])] * 2, out[r]
