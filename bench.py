#!/usr/bin/env python
"""Benchmark of the MonoCon hot path on MI355X (contract in the task brief).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--no-cpu-baseline]

One "step" = one pass of the hot path over one synthetic batch already resident in HBM:
the DLA-34 + DLAUp + dense-head forward at B=32, 3x384x1280, fp32 (BASELINE.json configs[1]).
For N>1 the driver launches this file under torch.distributed.run; every rank runs its own
B=32 shard (image-batch data parallelism, no data-path collective in the forward), timing is
barrier + synchronize bracketed and the MAX over ranks is reported (weak scaling).

Rank 0 prints ONE JSON line with the throughput, the roofline of the dominant kernel family
(fused conv on the fp32 MFMA pipe, measured live with HIP events on the launch stream) and
the CPU baseline (the oracle's CPU restatement of the same forward, timed on the host cores
on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(REPO, "monocon-pytorch_amd"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: fp32-in MFMA == fp32 vector peak
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step")
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train-steps", type=int, default=3,
                    help="also time this many full train steps (fwd+targets+losses+bwd+all-reduce+clip+AdamW); 0 = skip")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline time budget")
    return ap.parse_args()


def cpu_baseline(sd, height, width, budget_s):
    """The oracle (CPU restatement of the reference forward, equality with the reference pinned
    by tests/golden) timed on this box's host cores: eval forward at B=2, 384x1280."""
    from oracle import monocon_oracle as O
    from hipmonocon import synth
    img = synth.make_batch(3, 2, height, width, with_labels=False)["img"]
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    with torch.no_grad():
        # pick the fastest intra-op thread count the box actually sustains (a container may
        # expose far more logical CPUs than it is allowed to run; oversubscribing MKLDNN is
        # orders of magnitude slower), then time at that setting
        best = None
        for nt in sorted({n for n in (8, 16, 32, 64, avail) if n <= avail}):
            torch.set_num_threads(nt)
            t0 = time.perf_counter()
            O.forward(sd, img)                   # doubles as warm-up
            dt = time.perf_counter() - t0
            if best is None or dt < best[1]:
                best = (nt, dt)
            if dt > budget_s:                    # already hopeless at this count; larger is worse
                break
        torch.set_num_threads(best[0])
        times = []
        t_end = time.perf_counter() + budget_s
        while len(times) < 3 or (time.perf_counter() < t_end and len(times) < 50):
            t0 = time.perf_counter()
            O.forward(sd, img)
            times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return {"value": round(2 / med, 3), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle eval forward, B=2 x 3x%dx%d fp32, median of %d runs (%.3f s/run)"
                      % (height, width, len(times), med)}


def time_train_steps(args, sd, rank, world, dist_on, sync_all):
    """Full train step through the drop-in API (model.detector + solver): train-mode forward with
    batch-statistics BN, target generation, the ten losses, backward, gradient all-reduce (N > 1),
    fused clip + AdamW, cyclic schedule.  fp32 throughout (BASELINE configs[2] asks for bf16
    activations; not built yet -- this is the fp32 reference-precision step)."""
    from hipmonocon import synth
    from model import MonoConDetector
    from solver import AdamW, CyclicScheduler
    B, H, W = args.batch, args.height, args.width
    m = MonoConDetector(34, pretrained_backbone=False)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    opt = AdamW(m.parameters(), lr=2.25e-4, weight_decay=1e-5, betas=(0.95, 0.99), max_grad_norm=35.0)
    sch = CyclicScheduler(opt, total_steps=1000)
    nb = min(B, 8)
    small = synth.make_batch(500 + rank, nb, H, W)
    rep = (B + nb - 1) // nb
    batch = {"img": small["img"].repeat(rep, 1, 1, 1)[:B].cuda().contiguous(),
             "label": {k: v.repeat(rep, *([1] * (v.dim() - 1)))[:B].cuda().contiguous() for k, v in small["label"].items()},
             "img_metas": {"pad_shape": [(H, W)] * B}}

    def step():
        opt.zero_grad()
        _, loss = m(batch)
        total = sum(v for v in loss.values())
        total.backward()
        opt.step()
        sch.step()
        return total

    step()                       # warm-up (builds the plan)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.train_steps):
        total = step()
    sync_all()
    dt = time.perf_counter() - t0
    if dist_on:
        import torch.distributed as dist
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert bool(torch.isfinite(total))
    del m, opt
    return {"images_per_sec": round(world * B * args.train_steps / dt, 2), "ms_per_step": round(dt / args.train_steps * 1e3, 2),
            "steps": args.train_steps, "batch_per_gpu": B, "dtype": "f32",
            "what": "fwd(train BN)+targets+losses+bwd+%sclip+AdamW+cyclic schedule" % ("grad all-reduce+" if world > 1 else "")}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1
    if args.gpus != world and dist_on:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and not dist_on:
        raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`"
                         % (args.gpus, args.gpus))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from hipmonocon import synth
    from hipmonocon.engine import Engine

    stats = np.load(os.path.join(REPO, "tests", "golden", "bn_calib_seed7.npz"))
    sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
    eng = Engine(local)
    dsd = {k: v.cuda() for k, v in sd.items()}
    eng.bind_state(dsd)

    B, H, W = args.batch, args.height, args.width
    gen = torch.Generator(device="cuda").manual_seed(1234 + rank)
    img = torch.randn((B, 3, H, W), generator=gen, device="cuda", dtype=torch.float32)

    def sync_all():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.forward_infer(img)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        preds = eng.forward_infer(img)
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert all(torch.isfinite(v).all() for v in preds.values())

    train = None
    if args.train_steps > 0:
        train = time_train_steps(args, sd, rank, world, dist_on, sync_all)

    if rank == 0:
        cost = eng.forward_cost(B, H, W)
        flops = cost["conv_flops"] + cost["other_flops"]
        fbytes = cost["conv_bytes"] + cost["other_bytes"]
        prof = eng.profile_forward(iters=3)                     # HIP events on the launch stream
        # conv-MFMA family: algorithmic FLOPs of the conv launches / their summed duration
        conv_tflops = cost["conv_flops"] / (prof["conv_ms"] * 1e-3) / 1e12
        ms_step = elapsed / args.steps * 1e3
        out = {
            "metric": "images/sec (384x1280) fwd, DLA-34+DLAUp+MonoCon heads",
            "value": round(world * B * args.steps / elapsed, 2),
            "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: DLA-34 + MonoCon dense heads forward-only, "
                                   "batch=%d/GPU, 3x%dx%d synthetic, fp32, eval-mode BN" % (B, H, W),
                       "global_batch": world * B, "parallelism": "dp%d" % world},
            "roofline": {
                "bound": "mfma", "kernel": "conv_mfma_kernel (all %d fused-conv launches of one forward)" % prof["n_conv"],
                "achieved": round(conv_tflops, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(conv_tflops / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": None,
                "avg_launch_ms": round(prof["conv_ms"] / max(prof["n_conv"], 1), 4),
                "conv_ms_per_forward": round(prof["conv_ms"], 3), "other_ms_per_forward": round(prof["other_ms"], 3),
                "forward_gflop_per_image": round(flops / B / 1e9, 2),
                "forward_model_hbm_mb_per_image": round(fbytes / B / 1e6, 1),
                "whole_forward_tflops": round(flops / (ms_step * 1e-3) / 1e12, 2),
                "whole_forward_frac_of_fp32_peak": round(flops / (ms_step * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                "whole_forward_model_gbs": round(fbytes / (ms_step * 1e-3) / 1e9, 1),
                "whole_forward_frac_of_hbm_peak": round(fbytes / (ms_step * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
            },
            "workspace_gb": round(eng.workspace_bytes() / 1e9, 2),
        }
        if train is not None:
            out["train_step"] = train
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd, H, W, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
