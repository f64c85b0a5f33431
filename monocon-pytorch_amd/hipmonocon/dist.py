"""Image-batch data parallelism: one process per GPU, one gradient all-reduce per step.

The reference is single-GPU only (README.MD:11,15); this is the new MI355X-side design of
SURVEY §8(e): every rank holds a full replica, runs forward / targets / losses / backward on its own
shard of the global batch (BatchNorm statistics stay per replica, as torch DDP's default), then the
gradients of the 236 live parameter tensors -- kept in ONE contiguous fp32 buffer (78.3 MB) whose
views are the ``.grad`` tensors -- are summed across ranks with a single RCCL all-reduce over xGMI
(``torch.distributed`` backend ``nccl`` == RCCL on ROCm; ``gloo`` on CPU for the host-logic tests)
and divided by the world size.  Gradient clipping and AdamW then run identically on every rank.
"""
import os

import torch


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init_from_env(backend=None):
    """Initialise torch.distributed from the torchrun environment (no-op for a single process).
    Returns (world, rank, local_rank)."""
    import torch.distributed as dist
    world, rank, local = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, **kw)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return world, rank, local


def is_distributed():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def broadcast_seed(seed, src=0):
    """Every rank adopts rank ``src``'s seed (a per-rank random seed would initialise each replica differently)."""
    if not is_distributed():
        return int(seed)
    import torch.distributed as dist
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([int(seed)], dtype=torch.int64, device=dev)
    dist.broadcast(t, src=src)
    return int(t.item())


@torch.no_grad()
def sync_module_state(module, src=0):
    """Broadcast every parameter and buffer of ``module`` from rank ``src`` (after construction and after a
    checkpoint load): data parallelism only averages gradients, so replicas must START identical.  Floating-point
    tensors travel in one flat buffer per dtype; returns the number of tensors synchronised."""
    if not is_distributed():
        return 0
    import torch.distributed as dist
    tensors = [t for t in list(module.parameters()) + list(module.buffers())]
    groups = {}
    for t in tensors:
        groups.setdefault((t.dtype, t.device), []).append(t)
    for (_, _), ts in groups.items():
        flat = torch.cat([t.detach().reshape(-1) for t in ts])
        dist.broadcast(flat, src=src)
        off = 0
        for t in ts:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n
    return len(tensors)


def state_checksum(module):
    """Order-independent fp64 checksum of all parameters and buffers (test / assertion aid: equal across ranks)."""
    tot = 0.0
    for t in list(module.parameters()) + list(module.buffers()):
        tot += float(t.detach().double().sum()) + 1e-3 * float(t.detach().double().abs().sum())
    return tot


_HOST_GROUP = None          # (default process group it belongs to, gloo group or False)
_HOST_GROUP_WARNED = False


def _default_group_token():
    """identity of the CURRENT default process group: a cached subgroup of a destroyed / re-created default group is stale"""
    import torch.distributed as dist
    try:
        return dist.distributed_c10d._get_default_group()
    except Exception:           # noqa: BLE001
        return None


def host_group():
    """Process group for the small host-side votes of the step path.  On the 'nccl' backend a collective on a device tensor
    is ordered behind everything already queued on the stream, and reading its result is a host sync that stops the host
    from running ahead of the GPU; a gloo group carries the same few bytes between the hosts in ~0.1 ms without touching
    the device.  Created once per default process group, COLLECTIVELY (dist.new_group): call it first from a point every
    rank reaches (the engine's distributed set-up at its first step).  Whether the group is used is itself AGREED on by all
    ranks (a rank whose new_group failed alone would have voted on the nccl group while the others voted on
    gloo -- mismatched collectives, a hang): one MIN all-reduce of "my creation succeeded" over the default group, and a
    rank that succeeded where another failed destroys its group again.  The cache is tied to the default group's identity:
    after destroy_process_group() + init_process_group() in the same process a new group is made.
    None = use the default group (gloo runs, or gloo unavailable on some rank)."""
    global _HOST_GROUP
    if not is_distributed():
        return None
    import torch.distributed as dist
    token = _default_group_token()
    if _HOST_GROUP is None or _HOST_GROUP[0] is not token:
        grp = False
        if dist.get_backend() == "nccl":
            try:
                grp = dist.new_group(backend="gloo")
            except Exception:           # noqa: BLE001  (no gloo in this build: the votes fall back to device tensors)
                grp = False
            dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
            t = torch.tensor([1 if grp is not False else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)          # (once, at set-up: the one device-side vote)
            if not bool(t.item()):
                if grp is not False:
                    try:
                        dist.destroy_process_group(grp)
                    except Exception:   # noqa: BLE001
                        pass
                grp = False
        _HOST_GROUP = (token, grp)
    return _HOST_GROUP[1] or None


def host_votes_are_cheap():
    """False when the per-step votes would have to run on device tensors of the nccl backend (no host-side group): each one
    is then a stream sync in front of the overlapped gradient exchange -- the callers fall back to their cached verdicts
    (train._require_objects) and this warns once."""
    global _HOST_GROUP_WARNED
    if not is_distributed():
        return True
    import torch.distributed as dist
    if dist.get_backend() != "nccl" or host_group() is not None:
        return True
    if not _HOST_GROUP_WARNED:
        _HOST_GROUP_WARNED = True
        import warnings
        warnings.warn("hipmonocon: no gloo group beside the nccl backend -- the per-step label vote is skipped for label tensors "
                      "that were already validated on this rank (ranks must then re-use / refresh their label tensors in step)")
    return False


def all_ranks_ok(ok, device=None):
    """Logical AND of a per-rank flag, so that an assertion fires on every rank together instead of leaving the
    others hanging in the next collective."""
    if not is_distributed():
        return bool(ok)
    import torch.distributed as dist
    dev = device if (device is not None and dist.get_backend() == "nccl") else torch.device("cpu")
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


def dp_backend():
    """who averages the gradients across ranks.  'rccl': the communicator owned by the C handle (csrc/mc_comm.hip) --
    bucketed, overlapped with the backward, no PyTorch in the data path; the default whenever torch.distributed runs
    on the 'nccl' (= RCCL) backend.  'torch': one torch.distributed all_reduce of the flat buffer after the backward --
    the default on 'gloo' (CPU / several ranks on one device, which RCCL refuses) and the yard-stick the tests hold the
    former to.
    MONOCON_HIP_DP=rccl|torch overrides."""
    forced = os.environ.get("MONOCON_HIP_DP", "").strip().lower()
    if forced in ("rccl", "torch"):
        return forced
    if not is_distributed():
        return "torch"
    import torch.distributed as dist
    return "rccl" if dist.get_backend() == "nccl" else "torch"


def ensure_engine_comm(engine, force=False):
    """Give the engine's handle its RCCL communicator (once): rank 0 creates the id, torch.distributed (any backend)
    is only the host-side channel that carries its 128 bytes.  With ``force`` a world-1 communicator is created even
    without torch.distributed (self-test on a one-GPU box).  Returns True when the handle exchanges the gradients
    itself.  A communicator that cannot be built is a hard error on the 'rccl' backend unless MONOCON_HIP_DP_FALLBACK=1."""
    if engine.comm_world:
        return True
    world, rank = 1, 0
    if is_distributed():
        import torch.distributed as dist
        world, rank = dist.get_world_size(), dist.get_rank()
    elif not force:
        return False
    if not force and dp_backend() != "rccl":
        return False
    # every rank issues the SAME collective sequence whether or not something failed locally: broadcast of a status byte
    # + the 128-byte id, then one vote.  (A rank 0 that cannot create the id -- librccl not loadable, the case
    # MONOCON_HIP_DP_FALLBACK exists for -- still broadcasts, with status 0, instead of skipping to the vote while the
    # other ranks sit in the broadcast.)
    ok, err = True, None
    uid = None
    if world > 1:
        import torch.distributed as dist
        dev = engine.device if dist.get_backend() == "nccl" else torch.device("cpu")
        t = torch.zeros(129, dtype=torch.uint8, device=dev)
        if rank == 0:
            try:
                raw = engine.comm_unique_id()
                t[0] = 1
                t[1:].copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
            except Exception as e:      # noqa: BLE001
                ok, err = False, e
        dist.broadcast(t, src=0)
        host = t.cpu()
        if int(host[0]) != 1:
            ok, err = False, err or RuntimeError("rank 0 could not create the RCCL unique id")
        else:
            uid = bytes(host[1:].numpy().tobytes())
    else:
        try:
            uid = engine.comm_unique_id()
        except Exception as e:          # noqa: BLE001
            ok, err = False, e
    if ok:
        try:
            engine.comm_init(rank, world, uid)      # collective inside RCCL, under the library's watchdog (names the rank on expiry)
        except Exception as e:          # noqa: BLE001
            ok, err = False, e
    if world > 1 and not all_ranks_ok(ok, engine.device):
        ok = False
    if not ok:
        if engine.comm_world:
            engine.comm_destroy()
        if os.environ.get("MONOCON_HIP_DP_FALLBACK", "0") == "1" and not force:
            import sys
            if rank == 0:
                print("[hipmonocon] RCCL communicator unavailable (%s): falling back to torch.distributed all_reduce" % (err,),
                      file=sys.stderr, flush=True)
            # (err is None on a rank whose own init succeeded and which lost the vote)
            engine.comm_fallback_reason = str(err) if err is not None else "another rank could not join the communicator"
            os.environ["MONOCON_HIP_DP"] = "torch"
            return False
        raise RuntimeError("could not build the handle's RCCL communicator: %s" % (err,))
    return True


def share_tune_table(engine, B, H, W, src=0):
    """Identical convolution tilings on every rank: rank ``src`` builds (and autotunes) its train plan for this shape
    -- no kernel of the step, no collective -- and broadcasts the table of chosen workgroup shapes; the other ranks adopt
    it before they build their own plans.  (Concurrent timing-based tuning can pick different shapes per rank: results
    would stay bit-identical, step times would not.)  Returns the number of table entries; 0 outside a distributed run.
    COLLECTIVE: call it from a point every rank reaches (setup_engine_dp: the engine's first distributed step).  Every rank
    runs the same sequence whatever happens locally -- header broadcast (entry count, or the length of rank ``src``'s error
    text), body broadcast, one vote on the import -- and all of them raise together, with the source's error text."""
    if not is_distributed():
        return 0
    import torch.distributed as dist
    rank = dist.get_rank()
    dev = engine.device if dist.get_backend() == "nccl" else torch.device("cpu")
    table, err = [], None
    if rank == src:
        try:
            engine.build_train_plan(B, H, W)
            table = engine.tune_export()
        except Exception as e:          # noqa: BLE001  (the other ranks wait in the broadcast: report through the header)
            err = ("%s: %s" % (type(e).__name__, e)).encode("utf-8", "replace")[:2000]
    head = torch.tensor([len(table), len(err) if err is not None else -1], dtype=torch.int64, device=dev)
    dist.broadcast(head, src=src)
    n, nerr = int(head[0].item()), int(head[1].item())
    if nerr >= 0:       # the source failed: its message travels instead of the table
        msg = torch.frombuffer(bytearray(err), dtype=torch.uint8).to(dev) if rank == src else torch.zeros(nerr, dtype=torch.uint8, device=dev)
        if nerr:
            dist.broadcast(msg, src=src)
        raise RuntimeError("rank %d could not build / autotune its train plan: %s"
                           % (src, bytes(msg.cpu().numpy().tobytes()).decode("utf-8", "replace")))
    body = torch.tensor(table, dtype=torch.int32, device=dev) if rank == src else torch.zeros(n, dtype=torch.int32, device=dev)
    if n:
        dist.broadcast(body, src=src)
    ok, ierr = True, None
    if rank != src and n:
        try:
            engine.tune_import(body.cpu().tolist())
        except Exception as e:          # noqa: BLE001  (voted on below: nobody is left alone in the next collective)
            ok, ierr = False, e
    if not all_ranks_ok(ok, engine.device):
        raise RuntimeError("tune table of rank %d could not be imported on every rank%s" % (src, (": %s" % (ierr,)) if ierr is not None else ""))
    return n


def setup_engine_dp(engine, B, H, W):
    """The engine's data-parallel set-up, at its FIRST distributed step -- a point every rank of an SPMD loop reaches
    together, whatever its later batches look like (the tune-table broadcast used to be gated on the rank's
    own last input shape, so an uneven last batch on one rank mis-paired the collectives): the host-side vote group, one
    tuning table for all ranks, the handle's communicator.  Shapes that appear LATER are tuned locally by each rank, without
    a collective (results do not depend on the tilings; only step times can differ)."""
    if not is_distributed() or getattr(engine, "_dp_ready", False):
        return
    host_group()
    share_tune_table(engine, B, H, W)
    if not engine.comm_world:
        ensure_engine_comm(engine)                     # 'rccl' backend: the handle exchanges the gradients itself
    engine._dp_ready = True


def all_ranks_ok_many(flags, device=None, host=False):
    """element-wise logical AND over ranks of several per-rank flags, in ONE collective.  ``host``: over the host-side
    group (host_group(): no device work, no stream sync) when there is one."""
    flags = [bool(f) for f in flags]
    if not is_distributed():
        return flags
    import torch.distributed as dist
    grp = host_group() if host else None
    dev = device if (grp is None and device is not None and dist.get_backend() == "nccl") else torch.device("cpu")
    t = torch.tensor([1 if f else 0 for f in flags], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=grp)
    return [bool(v) for v in t.tolist()]


class FlatGrads:
    """One contiguous buffer holding the gradients of ``named`` (name, tensor-like with .shape/.numel)
    in order; ``views[name]`` aliases the slice of each tensor.  Slices start at multiples of 4
    elements so every view is 16-byte aligned for the float4 kernels."""

    def __init__(self, named, device, dtype=torch.float32):
        offs, total = {}, 0
        for n, p in named:
            offs[n] = total
            total += (p.numel() + 3) // 4 * 4
        self.flat = torch.zeros(total, dtype=dtype, device=device)
        self.views = {n: self.flat[offs[n]:offs[n] + p.numel()].view(p.shape) for n, p in named}
        self.numel = sum(p.numel() for _, p in named)

    def allreduce_mean(self, group=None):
        """sum over ranks / world, in place on the flat buffer (one collective per step)."""
        if not is_distributed():
            return self.flat
        import torch.distributed as dist
        if dist.get_backend(group) == "nccl":
            dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=group)        # sum / world inside the collective
        else:                                                                    # gloo has no AVG
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.mul_(1.0 / dist.get_world_size(group))
        return self.flat


def shard_indices(n_items, rank, world):
    """Contiguous, balanced partition of a global batch by image index: rank r gets
    [r*n/world, (r+1)*n/world)."""
    lo, hi = n_items * rank // world, n_items * (rank + 1) // world
    return list(range(lo, hi))


def shard_batch(data_dict, rank, world):
    """Slice a collated global batch (dataset/monocon_dataset.py:173-200 layout) for one rank."""
    idx = shard_indices(data_dict["img"].shape[0], rank, world)
    out = {"img": data_dict["img"][idx[0]:idx[-1] + 1]}
    if "label" in data_dict:
        out["label"] = {k: v[idx[0]:idx[-1] + 1] for k, v in data_dict["label"].items()}
    if "img_metas" in data_dict:
        out["img_metas"] = {k: [v[i] for i in idx] for k, v in data_dict["img_metas"].items()}
    if "calib" in data_dict:
        out["calib"] = [data_dict["calib"][i] for i in idx]
    return out
