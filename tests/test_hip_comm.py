"""The communicator behind the C-ABI (csrc/mc_comm.hip: mc_comm_init / mc_allreduce_grads, bucketed exchange overlapped with
mc_backward) on a ONE-GPU box: a world-1 RCCL communicator is the most a single device allows (RCCL refuses two ranks on
one device), so these tests make RCCL itself execute -- id creation, ncclCommInitRank, ncclAllReduce(ncclAvg) on the real
gradient buffer and streams, the event choreography inside mc_backward -- and check that an exchange over one rank is
the identity, bit for bit.  The N-rank arithmetic (mean of per-shard gradients vs the reference's fp64 goldens) is covered
by the gloo tests in tests/test_hip_round2.py / tests/test_dist_cpu.py through the torch.distributed path that the RCCL
path replaces."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN_SEED
from hipmonocon import dist as hdist
from hipmonocon import synth

pytestmark = pytest.mark.gpu


def build(sd):
    from model import MonoConDetector
    m = MonoConDetector(34, pretrained_backbone=False)
    m.load_state_dict(sd, strict=True)
    return m.cuda().train()


def to_cuda(batch):
    d = dict(batch)
    d["img"] = batch["img"].cuda()
    d["label"] = {k: v.cuda() for k, v in batch["label"].items()}
    return d


def step(m, batch):
    for p in m.parameters():
        p.grad = None
    _, loss = m(batch)
    sum(loss.values()).backward()
    torch.cuda.synchronize()
    return (torch.stack([v.detach() for v in loss.values()]).clone(),
            torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None]).clone())


def test_world1_rccl_exchange_of_the_gradient_buffer_is_the_identity(golden_sd):
    m = build(golden_sd)
    batch = to_cuda(synth.make_batch(GOLDEN_SEED + 31, 2, 64, 128))
    step(m, batch)                                   # binds the parameters and the flat gradient buffer
    eng = m._rt.engine
    assert eng.comm_world == 0
    assert hdist.ensure_engine_comm(eng, force=True)
    info = eng.comm_info()
    assert (info["rank"], info["world"], info["overlap"]) == (0, 1, True)
    assert "rccl" in info["library"]
    # the Python binding keeps all gradients in one flat buffer in parameter order: every bucket is ONE dense range
    assert info["collectives_per_exchange"] == 4, info
    flat = m._train_binding.flat.flat
    gen = torch.Generator(device="cuda").manual_seed(5)
    flat.copy_(torch.randn(flat.shape, generator=gen, device="cuda"))
    before = flat.clone()
    n0 = info["launches"]
    eng.allreduce_grads()
    torch.cuda.synchronize()
    assert torch.equal(flat, before)                 # average over one rank
    assert eng.comm_info()["launches"] == n0 + 4
    eng.comm_destroy()
    assert eng.comm_world == 0 and eng.comm_info()["world"] == 0


@pytest.mark.parametrize("overlap", [True, False], ids=["overlapped", "after_backward"])
def test_train_step_with_a_handle_owned_communicator_equals_the_plain_step(golden_sd, overlap):
    """the bucketed exchange inside mc_backward (events from both compute streams, RCCL on the communicator's stream, join
    at the end) must not change a single bit of the losses or gradients at world 1, and must have issued one collective
    per bucket"""
    batch = to_cuda(synth.make_batch(GOLDEN_SEED + 32, 2, 96, 160))
    mr = build(golden_sd)
    ref = [step(mr, batch) for _ in range(4)]        # (the running statistics move between steps: compare step k with step k)
    m = build(golden_sd)
    got = step(m, batch)
    assert torch.equal(got[0], ref[0][0]) and torch.equal(got[1], ref[0][1])
    eng = m._rt.engine
    hdist.ensure_engine_comm(eng, force=True)
    eng.comm_set_overlap(overlap)
    n0 = eng.comm_info()["launches"]
    for k in (1, 2):
        got = step(m, batch)
        assert torch.equal(got[0], ref[k][0]) and torch.equal(got[1], ref[k][1]), k
    assert eng.comm_info()["launches"] == n0 + 2 * 4
    if overlap:
        ms = eng.comm_exposed_ms()
        assert 0.0 <= ms < 50.0, ms
    eng.comm_destroy()
    got = step(m, batch)                             # and back to the path without a communicator
    assert torch.equal(got[1], ref[3][1])


def test_rccl_and_torch_paths_agree_on_one_rank(golden_sd):
    """the same step through the torch.distributed path (gloo process group of one rank) and through the handle's RCCL
    communicator: identical gradients (the torch path is what the N-rank goldens are checked through)"""
    import os
    import torch.distributed as dist
    batch = to_cuda(synth.make_batch(GOLDEN_SEED + 33, 2, 64, 128))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("gloo", rank=0, world_size=1)
        created = True
    try:
        ma = build(golden_sd)                        # world 1: is_distributed() is False -> plain path
        step(ma, batch)
        a = step(ma, batch)
        m = build(golden_sd)
        step(m, batch)
        hdist.ensure_engine_comm(m._rt.engine, force=True)
        b = step(m, batch)
        assert torch.equal(a[1], b[1])
        m._rt.engine.comm_destroy()
    finally:
        if created:
            dist.destroy_process_group()
