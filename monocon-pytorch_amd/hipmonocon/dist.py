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
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        self.flat.div_(dist.get_world_size(group))
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
