#!/usr/bin/env python
"""Benchmark of the MonoCon hot path on MI355X (contract in the task brief).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--no-cpu-baseline]

BASELINE.json's metric is "images/sec (384x1280) fwd+bwd at 1/2/4/8 GPUs": one "step" is one FULL
train step through the drop-in API (model.detector + solver) on a synthetic KITTI-shaped batch
already resident in HBM -- train-mode forward (batch-statistics BN, AttnBN heads), target
generation, the ten losses, backward (dgrad + wgrad), gradient all-reduce over RCCL when N > 1,
fused clip + AdamW and the cyclic schedule -- at B=32 images per GPU, 3x384x1280.

Precision of the headline: fp32 VALUES everywhere (activations, gradients, weights, BN statistics,
losses -- the reference trains in fp32), with the convolution arithmetic EMULATED on the fp16 matrix
pipe ("f16x2": every operand tensor scaled by the power of two its max |x| dictates and split into
two fp16 pieces, three partial products, fp32 accumulation).  It meets the same
fp32 tolerances as the native fp32 MFMA path: every golden test of `pytest -m gpu` runs in all three
fp32 modes, incl. the reference-fp64 train-step golden at the headline shape
(tests/test_hip_train_full.py).  `native_fp32` repeats the measurement on v_mfma_f32_32x32x2_f32,
`fp32_emulated_bf16x3` on the 3-way bf16 split (round 2's headline); `mixed_precision` (plain bf16
operands, BASELINE configs[2]) is a side figure -- it does NOT track the fp32 reference within the
parity budget (numbers in DESIGN.md 3a).  `realistic_loop` repeats the headline with what the
reference's loop does around the step (engine/monocon_engine.py:84-102): fresh label tensors every
step (label validation = a host sync), uint8 frames copied host -> device on a second stream and
normalised / padded on the GPU, `total_loss.item()` every step.

For N>1 the driver launches this file under torch.distributed.run (plain `python bench.py --gpus N`
re-launches itself that way): one process per GPU, every rank steps its own B=32 shard (weak
scaling), the 78 MB of gradients averaged over the ranks once per step -- four buckets on the
handle's own RCCL communicator, launched from inside mc_backward as the backward completes them
(csrc/mc_comm.hip) -- timing is barrier + synchronize bracketed and the MAX over ranks is reported;
`multi_gpu` carries every rank's own step time and the exposed part of the exchange.

Rank 0 prints ONE JSON line: the throughput, `roofline` of the dominant kernel family of the step
(forward convolutions + data gradients, timed live with HIP events around every launch on the
launch stream), `native_fp32`, `forward_only` (BASELINE configs[1]), `mixed_precision`,
`decode_only` (configs[4]: B=64, top-k 100) and `cpu_baseline` (the oracle's CPU restatement timed
on the host cores: train step B=2 plus the eval-forward and decode legs of SURVEY 8d).
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

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC for RCCL's peer mappings (read at runtime init)

import numpy as np
import torch

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: fp32-in MFMA == fp32 vector peak
PEAK_BF16_MFMA_TFLOPS = 2516.6     # same table: bf16 MFMA dense (~2.5 PF) = 16 x the fp32 MFMA rate
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
    ap.add_argument("--forward-steps", type=int, default=10,
                    help="also time this many eval forwards (BASELINE configs[1]); 0 = skip")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline time budget")
    ap.add_argument("--precision", default=os.environ.get("MONOCON_BENCH_PRECISION", "f16x2"), choices=("f16x2", "bf16x3", "fp32"),
                    help="precision mode of the headline value: all three keep fp32 values and meet the fp32 parity tolerances")
    ap.add_argument("--feed-steps", type=int, default=32,
                    help="steps per epoch of the engine_feed leg (N=1 only: RingLoader workers -> pinned ring -> copy stream -> "
                         "step, the engine's own loop); 0 = skip")
    ap.add_argument("--feed-workers", type=int, default=8)
    ap.add_argument("--realistic-steps", type=int, default=6,
                    help="steps of the realistic_loop leg (fresh labels + H2D of uint8 frames + loss.item() per step); 0 = skip")
    ap.add_argument("--strict-comm", action="store_true",
                    help="N > 1: exit non-zero if the handle cannot build its own RCCL communicator.  Default: measure with "
                         "torch.distributed's all_reduce (also RCCL, not overlapped) instead and SAY SO -- comm_path = 'torch "
                         "(FALLBACK: <reason>)' in the line, the reason on stderr; never silent")
    ap.add_argument("--allow-fallback", action="store_true", help="(the default since round 5; kept for old command lines)")
    ap.add_argument("--n1-json", default=None,
                    help="file holding the compact line of an N = 1 run of the same command: the line then carries scaling_vs_n1 = "
                         "value / (n_gpus x that run's value) (the driver computes its own from the per-N values; this is for reading "
                         "a run by hand)")
    ap.add_argument("--full-json", default=None,
                    help="where the verbose report (every leg with its prose notes) goes; default gpurun_out/bench_full.json when "
                         "that directory exists, else bench_full.json beside this file.  stdout carries ONE compact line")
    ap.add_argument("--no-extra-modes", action="store_true",
                    help="skip the fp32_emulated / mixed_precision legs (profiles/collect.sh: keeps the kernel trace on the headline path)")
    return ap.parse_args()


def _pick_cpu_threads():
    """intra-op thread count the box actually sustains (a container may expose far more logical CPUs than its quota
    lets it run, and oversubscribed MKLDNN is orders of magnitude slower), found on a CHEAP probe -- one 64->64 3x3
    convolution of the level-2 shape -- so that a bad candidate costs seconds, not minutes"""
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        pass
    cands = sorted({n for n in (4, 8, 16, 32, 64, 128, avail) if n <= avail})
    if quota:
        cands = [n for n in cands if n <= int(quota + 0.5)] or [max(1, int(quota))]
    px, pw = torch.randn(2, 64, 96, 320), torch.randn(64, 64, 3, 3)
    best = None
    for nt in cands:
        torch.set_num_threads(nt)
        torch.nn.functional.conv2d(px, pw, padding=1)
        dt = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            torch.nn.functional.conv2d(px, pw, padding=1)
            dt = min(dt, time.perf_counter() - t0)
        if best is None or dt < best[1]:
            best = (nt, dt)
        elif dt > 2.0 * best[1]:             # past the knee: larger counts only get worse
            break
    torch.set_num_threads(best[0])
    return best[0]


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(sd, height, width, budget_s):
    """The oracle (CPU restatement of the reference, equality with the reference pinned by tests/golden) timed on this
    box's host cores on bounded samples of the SAME workloads: full train steps at B=2 (the headline leg:
    value / unit / cores / kind / sample), the eval forward at B=2 and B=32, and the decode at B=64 / K=100."""
    from oracle import monocon_oracle as O
    from hipmonocon import synth
    threads = _pick_cpu_threads()
    batch = synth.make_batch(3, 2, height, width)
    names = [k for k, v in sd.items() if v.dtype == torch.float32 and not k.endswith(("running_mean", "running_var"))]

    def one_step(state):
        live = {k: (v.clone().requires_grad_(True) if k in names else v.clone()) for k, v in state.items()}
        _, _, L, _ = O.train_forward(live, batch)
        sum(L.values()).backward()
        ps = [live[k] for k in names if live[k].grad is not None]
        gs = [p.grad for p in ps]
        ms = [torch.zeros_like(p) for p in ps]
        vs = [torch.zeros_like(p) for p in ps]
        with torch.no_grad():
            O.clip_and_adamw([p.detach() for p in ps], gs, ms, vs, 1, 2.25e-4, 0.95)

    def timed(fn, budget, min_runs=3, max_runs=30, warm=True):
        if warm:
            fn()                                 # warm-up
        times = []
        t_end = time.perf_counter() + budget
        while len(times) < min_runs or (time.perf_counter() < t_end and len(times) < max_runs):
            t0 = time.perf_counter()
            fn()
            times.append(time.perf_counter() - t0)
        return float(np.median(times)), len(times)

    med, n = timed(lambda: one_step(sd), budget_s)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    out = {"value": round(2 / med, 3), "unit": "images/sec", "cores": threads, "kind": "port", "cpu": _cpu_model(),
           "sample": "oracle full train step (fwd + targets + losses + autograd bwd + clip + AdamW), B=2 x 3x%dx%d fp32, "
                     "median of %d runs (%.3f s/run), %d threads (%s; %d logical CPUs visible)"
                     % (height, width, n, med, threads, "the cgroup quota / the knee of a conv probe", avail)}
    legs = {}
    with torch.no_grad():
        img2 = batch["img"]
        m2, n2 = timed(lambda: O.forward(sd, img2), budget_s / 4)
        legs["eval_forward_b2"] = {"images_per_sec": round(2 / m2, 3), "s_per_run": round(m2, 3), "runs": n2}
        img32 = img2.repeat(16, 1, 1, 1)
        m32, n32 = timed(lambda: O.forward(sd, img32), 0.0, min_runs=3, max_runs=3, warm=True)   # warm-up 1, median of 3
        legs["eval_forward_b32"] = {"images_per_sec": round(32 / m32, 3), "s_per_run": round(m32, 3), "runs": n32,
                                    "note": "1 warm-up + median of 3"}
        K, DB = 100, 64
        d = {k: torch.from_numpy(v) for k, v in synth.make_decode_inputs(77, DB, height // 4, width // 4, topk=K).items()}
        P2 = np.stack([synth.KITTI_P2] * DB)
        md, nd = timed(lambda: O.decode(d, P2, (height, width), topk=K, thres=0.4), budget_s / 4)
        legs["decode_b64_k100"] = {"images_per_sec": round(DB / md, 1), "ms_per_batch": round(md * 1e3, 2), "runs": nd}
    out["legs"] = legs
    return out


def device_image_leg(engine, B=32, iters=20):
    """mc_preprocess_augmented on B deferred train samples of the mini tree (every operation drawn), HIP events on the launch
    stream: the device's share of the train list's image work"""
    from dataset.monocon_dataset import MonoConDataset
    mini = os.path.join(REPO, "tests", "golden", "kitti_mini")
    ds = MonoConDataset(mini, "train", aug_rng=np.random.default_rng(9), device_image=True)
    samples = [ds[i % len(ds)] for i in range(B)]
    frames = torch.stack([d["img"] for d in samples]).cuda()
    params = torch.stack([d["img_aug"] for d in samples]).cuda()
    for _ in range(3):
        out = engine.preprocess_augmented(frames, params)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        out = engine.preprocess_augmented(frames, params)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / iters
    nbytes = frames.numel() + out.numel() * 4
    return {"ms_per_batch": round(ms, 4), "batch": B, "frame": list(frames.shape[1:3]), "gb_per_s": round(nbytes / ms / 1e6, 1),
            "bytes": "3 B in (uint8 HWC, zero-padded) + 12 B out (float32 CHW) per pixel; includes the allocation of the output batch"}


def input_feed_capacity(train_img_per_s_per_gpu, seconds=2.0):
    """how fast ONE loader worker produces samples on this host, on the mini KITTI tree of tests/golden (real
    375x1242 frames: PNG decode with PIL, label file, calibration, filter rules, transforms), against what the GPUs consume.
    Four pipelines: the validation list and the train list (the reference's random augmentations, dataset/monocon_dataset.py:
    22-35), each with the image work on the host (Normalize / Pad / ToTensor; + the float32 colour round trip, shift, flip,
    crop for 'train') and with it deferred to the device (transforms.DeferredImage: the worker ships the decoded uint8 frame +
    24 parameters, mc_preprocess_augmented forms the float32 frame, bit-identical).  The headline fields are the TRAIN list
    as the engine runs it beside a device (deferred)."""
    mini = os.path.join(REPO, "tests", "golden", "kitti_mini")
    if not os.path.isdir(mini):
        return None
    from dataset.monocon_dataset import MonoConDataset

    def rate(split, device_image):
        ds = MonoConDataset(mini, split, aug_rng=np.random.default_rng(5), device_image=device_image)
        ds[0]
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            ds[n % len(ds)]
            n += 1
        return n / (time.perf_counter() - t0), n
    legs = {}
    for split in ("val", "train"):
        for dev in (False, True):
            r, n = rate(split, dev)
            legs["%s_%s" % (split, "device_image" if dev else "host_image")] = {"per_worker_img_per_s": round(r, 1), "samples": n}
    per_worker = legs["train_device_image"]["per_worker_img_per_s"]
    need = lambda g, r=per_worker: int(np.ceil(g * train_img_per_s_per_gpu / r))       # noqa: E731
    return {"per_worker_img_per_s": per_worker, "samples": legs["train_device_image"]["samples"],
            "workers_needed_1gpu": need(1), "workers_needed_8gpu": need(8),
            "workers_needed_1gpu_host_image": need(1, legs["train_host_image"]["per_worker_img_per_s"]),
            "legs": legs,
            "host_logical_cpus": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count(),
            "note": "one process, samples per second of MonoConDataset.__getitem__ on 375x1242 PNG frames; per_worker_img_per_s / workers_needed_* are the train list with the image work deferred to the device (what the engine runs beside a GPU), workers_needed_1gpu_host_image the same list with the reference's host-side image work"}


_T0 = time.perf_counter()


def _phase(msg):
    """progress to stderr (stdout carries only the JSON line)"""
    if os.environ.get("RANK", "0") == "0":
        print("[bench %7.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


MODES = {
    "bf16x3": {"dtype": "f32 (bf16x3-emulated: fp32 values, conv arithmetic on the bf16 matrix pipe by a 3-way operand split, "
                        "fp32 accumulation)",
               "family": "mc::conv_bf16_kernel", "peak": PEAK_BF16_MFMA_TFLOPS, "mfma_per_mac": 6.0,
               "unit": "TFLOP/s (bf16 MFMA executed: 6 partial products per fp32 multiply-add)"},
    "f16x2": {"dtype": "f32 (f16x2-emulated: fp32 values, conv arithmetic on the fp16 matrix pipe by a 2-way split of the "
                       "power-of-two-scaled operands, fp32 accumulation)",
              "family": "mc::conv_bf16_kernel", "peak": PEAK_BF16_MFMA_TFLOPS, "mfma_per_mac": 3.0,
              "unit": "TFLOP/s (fp16 MFMA executed: 3 partial products per fp32 multiply-add; dense fp16 peak = dense bf16 peak)"},
    "fp32": {"dtype": "f32 (native v_mfma_f32_32x32x2_f32)", "family": "mc::conv_mfma_kernel", "peak": PEAK_FP32_MFMA_TFLOPS,
             "mfma_per_mac": 1.0, "unit": "TFLOP/s"},
    "bf16": {"dtype": "bf16 operands / f32 accumulate", "family": "mc::conv_bf16_kernel", "peak": PEAK_BF16_MFMA_TFLOPS,
             "mfma_per_mac": 1.0, "unit": "TFLOP/s"},
}


# the kernels behind launch_conv() in the split modes: the event-timed conv bucket is their sum (conv_bf16_kernel dominates:
# 1 860 of the 1 976 sampled launches in profiles/r5m_pmc.txt)
CONV_FAMILY = ("mc::conv_bf16_kernel", "mc::conv_wres_kernel", "mc::conv_thin16_kernel", "mc::conv_small_kernel")


def _traffic(family):
    """HBM bytes per launch of a kernel family from the committed rocprofv3 PMC passes (cannot be collected in-process);
    for the conv family of the split modes: launch-weighted over the kernels launch_conv() dispatches to"""
    try:
        tj = json.load(open(os.path.join(REPO, "profiles", "latest_traffic.json")))
        members = [m for m in (CONV_FAMILY if family == CONV_FAMILY[0] else (family,)) if m in tj["families"]]
        fams = [tj["families"][m] for m in members]
        n = sum(f["launches_sampled"] for f in fams)
        mb = sum((f["hbm_read_bytes_per_launch"] + f["hbm_write_bytes_per_launch"]) * f["launches_sampled"] for f in fams) / n / 1e6
        t = sum(f["avg_us_under_pmc"] * f["launches_sampled"] for f in fams)
        pmc = {k: round(sum(f[k] * f["avg_us_under_pmc"] * f["launches_sampled"] for f in fams) / t, 3)
               for k in ("mfma_busy", "clock_ghz") if all(f.get(k) is not None for f in fams)}
        return round(mb, 1), tj["source"] + " [" + ", ".join(members) + "]", pmc
    except Exception:
        return None, None, {}


def _num(d, *path, nd=3):
    for k in path:
        if not isinstance(d, dict) or k not in d or d[k] is None:
            return None
        d = d[k]
    return round(d, nd) if isinstance(d, float) else d


def compact_line(out, full_path):
    """the contract keys + one short numeric block per BASELINE config / precision leg (no prose)"""
    c = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "world_size", "collective_backend", "steps", "warmup", "ms_per_step",
                             "higher_is_better", "scaling", "vs_baseline", "data") if k in out}
    c["dtype"] = "f32 (%s)" % out["config"]["precision_mode"]
    c["config"] = {"workload": "train step B=%d/GPU 3x384x1280 (configs[2] shape, fp32 values, mode %s)"
                               % (out["config"]["global_batch"] // max(out["n_gpus"], 1), out["config"]["precision_mode"]),
                   "global_batch": out["config"]["global_batch"], "parallelism": out["config"]["parallelism"],
                   "precision_mode": out["config"]["precision_mode"]}
    r = out.get("roofline") or {}
    c["roofline"] = {"bound": r.get("bound"), "kernel": "mc::conv_bf16_kernel" if "bf16" in str(r.get("kernel")) else str(r.get("kernel"))[:40],
                     "achieved": r.get("achieved"), "peak": r.get("peak"), "unit": "TFLOP/s", "frac": r.get("frac"),
                     "traffic": r.get("traffic"), "traffic_unit": "MB/launch (PMC)",
                     "alg_mb_per_launch": r.get("algorithmic_mb_per_launch"), "avg_launch_ms": r.get("avg_launch_ms"),
                     "mfma_per_mac": 3 if out["config"]["precision_mode"] == "f16x2" else (6 if out["config"]["precision_mode"] == "bf16x3" else 1),
                     "alg_tflops_fp32eq": r.get("algorithmic_tflops_fp32_equivalent"),
                     "mfma_busy": _num(r, "pmc", "mfma_busy"), "clock_ghz": _num(r, "pmc", "clock_ghz"),
                     "conv_ms": r.get("conv_ms_per_step"), "wgrad_ms": _num(r, "wgrad", "ms_per_step"), "wgrad_frac": _num(r, "wgrad", "frac", nd=4),
                     "other_ms": r.get("other_ms_per_step")}
    # the WHOLE step against the matrix pipe: (conv + data-gradient + weight-gradient FLOPs) x MFMAs per MAC / step time / peak
    # (`frac` above is the conv family timed alone; the buckets overlap in the step)
    if r.get("whole_step_tflops_fp32_equivalent") and r.get("peak"):
        c["roofline"]["whole_step_frac"] = round(r["whole_step_tflops_fp32_equivalent"] * c["roofline"]["mfma_per_mac"] / r["peak"], 4)
    c["img_s_per_gpu"] = round(out["value"] / max(out.get("n_gpus", 1), 1), 2) if isinstance(out.get("value"), (int, float)) else None
    if out.get("scaling_vs_n1") is not None:
        c["scaling_vs_n1"] = out["scaling_vs_n1"]
    c["workspace_gb"] = out.get("workspace_gb")
    f = out.get("forward_only")
    if f:      # BASELINE configs[1]
        c["forward_only"] = {"img_s": f["images_per_sec"], "ms": f["ms_per_step"], "conv_ms": _num(f, "roofline", "conv_ms"),
                             "other_ms": _num(f, "roofline", "other_ms"), "mfma_frac": _num(f, "roofline", "frac", nd=4),
                             "hbm_frac": f.get("whole_forward_frac_of_hbm_peak"), "hbm_frac_max_attainable": f.get("max_attainable_hbm_frac")}
    modes = {}
    for key, tag in (("native_fp32", "fp32"), ("fp32_emulated_bf16x3", "bf16x3"), ("fp32_emulated_f16x2", "f16x2"), ("mixed_precision", "bf16")):
        b = out.get(key)
        if b:
            modes[tag] = {"train_img_s": _num(b, "train", "images_per_sec"), "train_ms": _num(b, "train", "ms_per_step"),
                          "fwd_img_s": _num(b, "forward", "images_per_sec"), "fwd_ms": _num(b, "forward", "ms_per_step"),
                          "conv_frac": _num(b, "train", "roofline", "frac", nd=4), "fwd_hbm_frac": _num(b, "forward", "whole_forward_frac_of_hbm_peak", nd=4)}
    if modes:
        c["modes"] = modes
    if out.get("realistic_loop"):
        c["realistic_loop"] = {"img_s": out["realistic_loop"]["images_per_sec"], "ms": out["realistic_loop"]["ms_per_step"]}
    if out.get("engine_feed_kitti") and "ms_per_step" in out["engine_feed_kitti"]:
        k = out["engine_feed_kitti"]
        c["engine_feed_kitti"] = {"img_s": k["images_per_sec"], "ms": k["ms_per_step"], "resident_ms": k["resident_ms_per_step"],
                                  "workers": k["workers"]}
    if out.get("engine_feed") and "ms_per_step" in out["engine_feed"]:
        c["engine_feed"] = {"img_s": out["engine_feed"]["images_per_sec"], "ms": out["engine_feed"]["ms_per_step"],
                            "workers": out["engine_feed"]["workers"]}
    if out.get("input_feed"):
        c["input_feed"] = {k: v for k, v in out["input_feed"].items() if k != "note"}
    if out.get("decode_only"):     # BASELINE configs[4]
        c["decode_only"] = {"img_s": out["decode_only"]["images_per_sec"], "ms_per_batch": out["decode_only"]["ms_per_batch"], "batch": 64, "topk": 100}
    if out.get("kitti_eval_overlaps"):
        c["kitti_eval_overlaps"] = {k: out["kitti_eval_overlaps"][k] for k in ("bev_ms", "box3d_ms")}
    if out.get("power_ceiling"):
        c["power_ceiling"] = out["power_ceiling"]
    for k in ("comm_world", "n_collectives", "comm_path", "comm_fallback", "per_rank_ms_per_step", "per_rank_exposed_allreduce_ms"):
        if k in out:
            c[k] = out[k]
    cb = out.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                             "sample": "oracle train step B=2 3x384x1280 fp32, median of runs in %.0f s" % 12,
                             "cpu": cb.get("cpu"),
                             "eval_fwd_b2_img_s": _num(cb, "legs", "eval_forward_b2", "images_per_sec"),
                             "eval_fwd_b32_img_s": _num(cb, "legs", "eval_forward_b32", "images_per_sec"),
                             "decode_b64_img_s": _num(cb, "legs", "decode_b64_k100", "images_per_sec")}
    c["full_report"] = os.path.relpath(full_path, REPO) if full_path else None
    return c


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1
    if args.gpus != world and dist_on:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and not dist_on:
        # plain `python bench.py --gpus N`: re-launch as N ranks (one process per GPU) under torch.distributed.run
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        _phase("--gpus %d without a torchrun environment: re-launching as %s" % (args.gpus, " ".join(cmd[1:9])))
        raise SystemExit(subprocess.call(cmd))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # test hook (1-GPU box): MONOCON_BENCH_BACKEND=gloo runs all ranks on device 0 over gloo, to exercise the
    # N > 1 control flow (barriers, max over ranks, gradient all-reduce) where RCCL refuses a shared device
    backend = os.environ.get("MONOCON_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = 0
    torch.cuda.set_device(local)
    dist = None
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the exchange path is part of what is measured and is NAMED in the line (comm_path): a communicator the handle cannot
        # build switches every rank (voted) to torch.distributed's all_reduce with the reason in the line and on stderr;
        # --strict-comm makes it a hard error on every rank (non-zero exit) instead
        os.environ["MONOCON_HIP_DP_FALLBACK"] = "0" if args.strict_comm else "1"
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    from hipmonocon import synth
    from hipmonocon import dist as hdist
    from model import MonoConDetector
    from solver import AdamW, CyclicScheduler

    stats = np.load(os.path.join(REPO, "tests", "golden", "bn_calib_seed7.npz"))
    sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
    B, H, W = args.batch, args.height, args.width
    headline_mode = args.precision

    def sync_all():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(dt):
        if not dist_on:
            return dt
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    m = MonoConDetector(34, pretrained_backbone=False)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    hdist.sync_module_state(m)                      # (identical already; this is what a training run does)
    opt = AdamW(m.parameters(), lr=2.25e-4, weight_decay=1e-5, betas=(0.95, 0.99), max_grad_norm=35.0)
    sch = CyclicScheduler(opt, total_steps=1000)
    nb = min(B, 8)
    small = synth.make_batch(500 + rank, nb, H, W)
    rep = (B + nb - 1) // nb
    batch = {"img": small["img"].repeat(rep, 1, 1, 1)[:B].cuda().contiguous(),
             "label": {k: v.repeat(rep, *([1] * (v.dim() - 1)))[:B].cuda().contiguous() for k, v in small["label"].items()},
             "img_metas": {"pad_shape": [(H, W)] * B}}
    gen = torch.Generator(device="cuda").manual_seed(1234 + rank)
    eval_img = torch.randn((B, 3, H, W), generator=gen, device="cuda", dtype=torch.float32)

    def step():
        opt.zero_grad()
        _, loss = m(batch)
        total = sum(v for v in loss.values())
        total.backward()            # includes the gradient all-reduce when N > 1
        opt.step()                  # fused clip_grad_norm_(35) + AdamW
        sch.step()
        return total

    def train_leg(mode, steps, warmup, profile):
        """W untimed + K timed full train steps in `mode`; barrier + synchronize on both sides, max over ranks"""
        m.train().set_precision(mode)
        for _ in range(max(warmup, 1)):           # the first step in a mode builds (and autotunes) its plan
            total = step()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            total = step()
        sync_all()
        mine = time.perf_counter() - t0
        elapsed = max_over_ranks(mine)
        assert bool(torch.isfinite(total)), "non-finite loss in the timed region (%s)" % mode
        out = {"ms_per_step": elapsed / steps * 1e3, "images_per_sec": world * B * steps / elapsed}
        if dist_on:       # every rank's own clock + how long the last exchange kept its stream waiting
            eng_ = m._rt.engine
            exposed = eng_.comm_exposed_ms() if eng_.comm_world else -1.0
            t = torch.tensor([mine / steps * 1e3, exposed], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
            allt = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(allt, t)
            out["per_rank_ms_per_step"] = [round(float(x[0]), 3) for x in allt]
            out["per_rank_exposed_allreduce_ms"] = [round(float(x[1]), 3) for x in allt]
            out["comm"] = eng_.comm_info() if eng_.comm_world else {"path": "torch.distributed all_reduce after backward",
                                                                     "fallback_reason": getattr(eng_, "comm_fallback_reason", None)}
        if profile and rank == 0:
            out["profile"] = m._rt.engine.profile_train(iters=2)   # HIP events around every launch on the launch stream
            out["workspace_gb"] = m._rt.engine.workspace_bytes() / 1e9
        return out

    def realistic_leg(mode, steps):
        """The headline step inside the loop the reference runs around it (engine/monocon_engine.py:84-102): every step
        gets NEW label tensors (so the label validation -- a host sync -- happens, as it would with a DataLoader), its frames
        arrive as uint8 HWC in pinned host memory, are copied to the device on a second stream (overlapping the previous
        step) and normalised / padded by mc_preprocess, and `total_loss.item()` is read every step."""
        m.train().set_precision(mode)
        eng_ = m._rt.engine
        nb_ = min(B, 8)
        frames_host = [torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(77 + i)).pin_memory()
                       for i in range(2)]
        labels_host = [synth.make_batch(600 + i + 10 * rank, nb_, H, W)["label"] for i in range(steps + 2)]
        copy_stream = torch.cuda.Stream()
        slots = [torch.empty((B, H, W, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
        ev = [torch.cuda.Event() for _ in range(2)]

        def feed(i):        # H2D of step i's frames + labels on the copy stream
            with torch.cuda.stream(copy_stream):
                slots[i % 2].copy_(frames_host[i % 2], non_blocking=True)
                lab = {k: v.repeat(rep, *([1] * (v.dim() - 1)))[:B].pin_memory().cuda(non_blocking=True)
                       for k, v in labels_host[i].items()}
                ev[i % 2].record(copy_stream)
            return lab

        def one(i, lab):
            torch.cuda.current_stream().wait_event(ev[i % 2])
            img, _ = eng_.preprocess(list(slots[i % 2].unbind(0)))
            opt.zero_grad()
            _, loss = m({"img": img, "label": lab, "img_metas": {"pad_shape": [(H, W)] * B}})
            total = sum(v for v in loss.values())
            total.backward()
            opt.step()
            sch.step()
            return total

        lab = feed(0)
        nxt = feed(1)
        one(0, lab).item()                      # warm-up (same plan as the headline leg)
        sync_all()
        t0 = time.perf_counter()
        for i in range(1, steps + 1):
            lab, nxt = nxt, feed(i + 1)
            v = one(i, lab).item()              # the reference logs the loss every step (monocon_engine.py:89)
        sync_all()
        dt = max_over_ranks(time.perf_counter() - t0)
        assert np.isfinite(v)
        return {"ms_per_step": round(dt / steps * 1e3, 3), "images_per_sec": round(world * B * steps / dt, 2), "steps": steps,
                "workload": "headline train step inside the reference's loop: fresh label tensors per step (label validation "
                            "host sync), uint8 HWC frames host -> device (pinned, second stream, %.0f MB/step) + mc_preprocess, "
                            "total_loss.item() every step" % (B * H * W * 3 / 1e6)}

    def engine_feed_leg(mode, steps, workers):
        """The headline step fed the way MonoconEngine.train_one_epoch feeds it (engine/monocon_engine.py, hipmonocon/feed.py):
        worker processes write float32 CHW frames (the reference's collate contract, monocon_dataset.py:173-200) into the
        shared page-locked ring, the batch is uploaded one step ahead on a copy stream, the labels are checked on the host,
        the loss of each step is read back one step late.  One epoch to start the workers, one timed (the loader serves all epochs from one
        iterator: the first batches of an epoch are in the ring before the previous one ends)."""
        from dataset.synthetic_dataset import PooledSyntheticDataset
        from hipmonocon.feed import DeferredScalars, DevicePrefetcher, RingLoader
        m.train().set_precision(mode)
        ds = PooledSyntheticDataset(B * steps, H, W, pool=8, seed=900)
        ring = RingLoader(ds, B, workers, shuffle=True, collate_fn=ds.collate_fn, timeout=180)
        try:
            def epoch():
                losses, got, t_wait = DeferredScalars(), [], 0.0
                it = iter(DevicePrefetcher(ring, torch.device("cuda", torch.cuda.current_device()), m))
                while True:
                    ta = time.perf_counter()
                    batch = next(it, None)
                    t_wait += time.perf_counter() - ta
                    if batch is None:
                        break
                    opt.zero_grad()
                    _, loss = m(batch)
                    total = sum(v for v in loss.values())
                    total.backward()
                    losses.push(total)
                    opt.step()
                    sch.step()
                    got += losses.ready(1)
                return got + losses.ready(0), t_wait
            epoch()
            sync_all()
            t0 = time.perf_counter()
            got, t_wait = epoch()
            sync_all()
            dt = time.perf_counter() - t0
            pinned = bool(ring.pinned)
        finally:
            ring.close()
        assert len(got) == steps and all(np.isfinite(v) for v in got)
        return {"ms_per_step": round(dt / steps * 1e3, 3), "images_per_sec": round(B * steps / dt, 2), "steps": steps,
                "workers": workers, "host_ms_per_step_waiting_for_the_batch": round(t_wait / steps * 1e3, 3),
                "ring_gb": round(ring.ring.numel() * 4 / 1e9, 2), "pinned": pinned,
                "workload": "headline train step fed by the engine's loop: %d fork-server workers write float32 CHW frames "
                            "(%.0f MB per batch, pre-drawn samples) into a shared page-locked ring of %d batch slots, upload "
                            "one step ahead on a copy stream, labels checked on the host, loss read back one step late"
                            % (workers, B * 3 * H * W * 4 / 1e6, ring.nslots)}

    def kitti_feed_leg(mode, steps, workers):
        """engine_feed on REAL frames: the workers decode 375x1242 KITTI PNGs (the two frames of tests/golden/kitti_mini, over and
        over), parse labels and calibration, draw the train list's random augmentations and move the labels; the frames travel
        as uint8 through the ring and `mc_preprocess_augmented` forms the float32 batch (3x384x1248: what Pad(32) makes of KITTI
        frames) in front of the step.  The same loop on ONE resident batch of that shape is timed beside it."""
        from dataset.monocon_dataset import MonoConDataset, RepeatedDataset
        from hipmonocon.feed import DeferredScalars, DevicePrefetcher, RingLoader
        mini = os.path.join(REPO, "tests", "golden", "kitti_mini")
        if not os.path.isdir(mini):
            return None
        m.train().set_precision(mode)
        dev = torch.device("cuda", torch.cuda.current_device())
        ds = RepeatedDataset(MonoConDataset(mini, "train", aug_rng=np.random.default_rng(31), device_image=True), B * steps)

        def run(feed):
            losses, got, n = DeferredScalars(), [], 0
            for batch in feed:
                opt.zero_grad()
                _, loss = m(batch)
                total = sum(v for v in loss.values())
                total.backward()
                losses.push(total)
                opt.step()
                sch.step()
                got += losses.ready(1)
                n += 1
            return got + losses.ready(0), n

        def timed(make_feed):
            sync_all()
            t0 = time.perf_counter()
            got, n = run(make_feed())
            sync_all()
            dt = time.perf_counter() - t0
            assert n and all(np.isfinite(v) for v in got)
            return dt / n * 1e3

        host = ds.collate_fn([ds[i] for i in range(B)])
        first = next(iter(DevicePrefetcher([host], dev, m)))
        run([first])                                    # builds (and tunes) the train plan of this shape; `first` now holds float frames
        res_ms = timed(lambda: [dict(first) for _ in range(steps)])
        ring = RingLoader(ds, B, workers, shuffle=True, collate_fn=ds.collate_fn, timeout=180)
        try:
            run(DevicePrefetcher(ring, dev, m))         # starts the workers
            ms = timed(lambda: DevicePrefetcher(ring, dev, m))
            pinned, slots, gb = bool(ring.pinned), ring.nslots, ring.ring.numel() / 1e9
        finally:
            ring.close()
        return {"ms_per_step": round(ms, 3), "images_per_sec": round(B * 1e3 / ms, 2), "resident_ms_per_step": round(res_ms, 3),
                "steps": steps, "workers": workers, "ring_gb": round(gb, 2), "pinned": pinned,
                "workload": "train step at 3x384x1248 (KITTI frames after Pad(32)), B=%d, fed by the engine's loop from REAL 375x1242 PNG "
                            "frames: %d fork-server workers decode, parse labels, draw the reference's train augmentations; uint8 "
                            "frames through a page-locked ring of %d slots, image work + Normalize/Pad/ToTensor on the device "
                            "(mc_preprocess_augmented); resident_ms_per_step = the same loop on one resident batch of that shape"
                            % (B, workers, slots)}

    def forward_leg(mode, steps):
        m.eval().set_precision(mode)
        eng = m._engine()           # re-binds the (updated) parameters for the eval plan
        for _ in range(3):
            preds = eng.forward_infer(eval_img)
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            preds = eng.forward_infer(eval_img)
        sync_all()
        fdt = max_over_ranks(time.perf_counter() - t0)
        assert all(torch.isfinite(v).all() for v in preds.values())
        out = {"ms_per_step": fdt / steps * 1e3, "images_per_sec": world * B * steps / fdt}
        if rank == 0:
            out["cost"] = eng.forward_cost(B, H, W)
            out["profile"] = eng.profile_forward(iters=3)
        return out

    def roofline_of(mode, conv, n_launch_label):
        mi = MODES[mode]
        alg_tf = conv["flops"] / (conv["ms"] * 1e-3) / 1e12            # algorithmic fp32 FLOPs of the family / its time
        exe_tf = alg_tf * mi["mfma_per_mac"]                            # MFMA FLOPs the kernel executes for them
        traffic, src, pmc = _traffic(mi["family"])
        r = {"bound": "mfma", "kernel": "%s: %s" % (mi["family"], n_launch_label),
             "achieved": round(exe_tf, 2), "peak": mi["peak"], "unit": mi["unit"], "frac": round(exe_tf / mi["peak"], 4),
             "traffic": traffic, "traffic_unit": "MB of HBM traffic per launch (PMC)", "traffic_source": src,
             "algorithmic_tflops_fp32_equivalent": round(alg_tf, 2),
             "frac_of_fp32_mfma_peak": round(alg_tf / PEAK_FP32_MFMA_TFLOPS, 4),
             "algorithmic_mb_per_launch": round(conv.get("bytes", 0.0) / max(conv["launches"], 1) / 1e6, 1),
             "avg_launch_ms": round(conv["ms"] / max(conv["launches"], 1), 4)}
        if pmc:   # committed PMC pass of the same command: how busy the matrix pipe is and at which clock the chip sustains it
            r["pmc"] = dict(pmc, note="SQ_VALU_MFMA_BUSY_CYCLES / SIMD cycles and GRBM_GUI_ACTIVE / time, time-weighted over the "
                                      "family's launches; frac ~= mfma_busy * clock_ghz / 2.4 -- the matrix kernels of this mode "
                                      "are power-limited (MI355X_MICROARCH.md, DVFS give-back)")
        return r

    def train_report(mode, leg):
        out = {"images_per_sec": round(leg["images_per_sec"], 2), "ms_per_step": round(leg["ms_per_step"], 3),
               "dtype": MODES[mode]["dtype"]}
        if "profile" in leg:
            conv, wg, oth = leg["profile"]["conv"], leg["profile"]["wgrad"], leg["profile"]["other"]
            r = roofline_of(mode, conv, "%d launches per train step (forward convs + data gradients)" % conv["launches"])
            wg_tf = wg["flops"] / (wg["ms"] * 1e-3) / 1e12 if wg["ms"] > 0 else 0.0
            r.update({"conv_ms_per_step": round(conv["ms"], 2),
                      "wgrad": {"kernel": "weight gradients (+ split-K reduce): %d launches" % wg["launches"],
                                "algorithmic_tflops_fp32_equivalent": round(wg_tf, 2),
                                "achieved": round(wg_tf * MODES[mode]["mfma_per_mac"], 2),
                                "frac": round(wg_tf * MODES[mode]["mfma_per_mac"] / MODES[mode]["peak"], 4),
                                "ms_per_step": round(wg["ms"], 2)},
                      "other_ms_per_step": round(oth["ms"], 2),
                      "step_gflop_per_image": round((conv["flops"] + wg["flops"]) / B / 1e9, 1),
                      "whole_step_tflops_fp32_equivalent": round((conv["flops"] + wg["flops"]) / (leg["ms_per_step"] * 1e-3) / 1e12, 2)})
            out["roofline"] = r
            out["workspace_gb"] = round(leg["workspace_gb"], 2)
        return out

    def forward_report(mode, leg):
        out = {"images_per_sec": round(leg["images_per_sec"], 2), "ms_per_step": round(leg["ms_per_step"], 3),
               "dtype": MODES[mode]["dtype"]}
        if "cost" in leg:
            cost, fprof = leg["cost"], leg["profile"]
            conv = {"flops": cost["conv_flops"], "ms": fprof["conv_ms"], "launches": fprof["n_conv"], "bytes": cost["conv_bytes"]}
            r = roofline_of(mode, conv, "%d launches per forward" % fprof["n_conv"])
            r.update({"conv_ms": round(fprof["conv_ms"], 3), "other_ms": round(fprof["other_ms"], 3)})
            fl = cost["conv_flops"] + cost["other_flops"]
            by = cost["conv_bytes"] + cost["other_bytes"]
            # the bound of north_star's "fraction of the HBM roofline" in this mode: the forward cannot run faster than its
            # matrix work allows (algorithmic FLOPs x MFMAs per multiply-add / dense peak of the pipe it runs on)
            t_hbm = by / (PEAK_HBM_GBS * 1e9)
            t_mfma = cost["conv_flops"] * MODES[mode]["mfma_per_mac"] / (MODES[mode]["peak"] * 1e12)
            out.update({"roofline": r, "gflop_per_image": round(fl / B / 1e9, 2), "model_hbm_mb_per_image": round(by / B / 1e6, 1),
                        "whole_forward_tflops_fp32_equivalent": round(fl / (leg["ms_per_step"] * 1e-3) / 1e12, 2),
                        "whole_forward_frac_of_hbm_peak": round(by / (leg["ms_per_step"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                        "max_attainable_hbm_frac": round(t_hbm / max(t_hbm, t_mfma), 4),
                        "max_attainable_note": "HBM time of the fusion-model bytes (%.2f ms) / max(that, matrix-pipe time of the "
                                               "convolutions at the dense peak of this mode's pipe (%.2f ms))" % (t_hbm * 1e3, t_mfma * 1e3)})
        return out

    # ---------------------------------------------------------------- the train step (headline)
    _phase("model + batch resident; headline mode %s: first step builds and autotunes the train plan" % headline_mode)
    head = train_leg(headline_mode, args.steps, args.warmup, profile=True)
    _phase("timed region done: %.2f ms/step" % head["ms_per_step"])
    fwd_head = forward_leg(headline_mode, args.forward_steps) if args.forward_steps > 0 else None

    extra = {}
    if args.forward_steps > 0 and not args.no_extra_modes:
        for other in [o for o in ("fp32", "bf16x3", "f16x2") if o != headline_mode]:
            _phase("mode %s" % other)
            extra[other] = (train_leg(other, args.steps, 2, profile=True), forward_leg(other, args.forward_steps))
        _phase("mode bf16")
        extra["bf16"] = (train_leg("bf16", args.steps, 2, profile=False), forward_leg("bf16", args.forward_steps))
    real = None
    if args.realistic_steps > 0:
        _phase("realistic loop (fresh labels, H2D of uint8 frames on a second stream, loss.item() per step)")
        real = realistic_leg(headline_mode, args.realistic_steps)
    feed = kfeed = None
    if args.feed_steps > 0 and world == 1:
        _phase("engine feed (RingLoader workers -> pinned ring -> copy stream -> step)")
        try:
            feed = engine_feed_leg(headline_mode, args.feed_steps, args.feed_workers)
        except Exception as e:      # noqa: BLE001  (a host without /dev/shm room, ...: reported, the bench line stands)
            feed = {"error": "%s: %s" % (type(e).__name__, e)}
        _phase("engine feed on real KITTI frames (PNG decode + train augmentations in the workers, image work on the device)")
        try:
            kfeed = kitti_feed_leg(headline_mode, args.feed_steps, args.feed_workers)
        except Exception as e:      # noqa: BLE001
            kfeed = {"error": "%s: %s" % (type(e).__name__, e)}
    m.set_precision(headline_mode)
    eng = m.eval()._engine()

    # ---------------------------------------------------------------- decode only (configs[4])
    dec = None
    if args.forward_steps > 0 and rank == 0:
        _phase("decode")
        K, DB = 100, 64
        from hipmonocon.engine import p2_inverse
        dpred = {k: torch.from_numpy(v).cuda() for k, v in synth.make_decode_inputs(77, DB, H // 4, W // 4, topk=K).items()}
        P2np = np.stack([synth.KITTI_P2] * DB)
        P2, P2inv = torch.from_numpy(P2np).cuda(), torch.from_numpy(p2_inverse(P2np)).cuda()
        for _ in range(3):
            eng.decode(dpred, P2, P2inv, (H, W), K, 0.4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            eng.decode(dpred, P2, P2inv, (H, W), K, 0.4)
        torch.cuda.synchronize()
        dms = (time.perf_counter() - t0) / 20 * 1e3
        dec = {"workload": "BASELINE configs[4]: decode of the ten maps, batch=%d, top-k=%d (local max + exact top-K + box "
                           "assembly; indices / keep masks bit-exact vs the oracle in tests/test_hip_decode.py)" % (DB, K),
               "images_per_sec": round(DB / (dms * 1e-3), 1), "ms_per_batch": round(dms, 3),
               "heatmap_gb_per_s": round(DB * 3 * (H // 4) * (W // 4) * 4 / (dms * 1e-3) / 1e9, 1),
               "note": "two launches (filter + candidate compaction, per-image select + box assembly); the only "
                       "algorithmic traffic is one read of the 23.6 MB heat map, so this is launch- and latency-bound"}

        # ---- the evaluator's device kernels: rotated BEV IoU / 3D IoU of 4096 x 4096 camera-frame boxes
        _phase("kitti_eval overlaps")
        NB = 4096
        rs = np.random.RandomState(5)
        bx = np.stack([rs.uniform(-40, 40, NB), rs.uniform(1.0, 2.5, NB), rs.uniform(5, 70, NB), rs.uniform(0.6, 4.5, NB),
                       rs.uniform(1.2, 2.2, NB), rs.uniform(0.5, 2.0, NB), rs.uniform(-np.pi, np.pi, NB)], 1)
        b7 = torch.from_numpy(bx).cuda()
        b5 = b7[:, [0, 2, 3, 5, 6]].float().contiguous()
        ev_ms = {}
        for nm, fn in (("bev", lambda: eng.rotate_iou(b5, b5)), ("3d", lambda: eng.box3d_overlap(b7, b7))):
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            ev_ms[nm] = (time.perf_counter() - t0) / 5 * 1e3
        evl = {"workload": "KITTI AP evaluator, device part: pairwise rotated overlaps of %d x %d boxes (mc_rotate_iou_eval, "
                           "mc_box3d_overlap; vs the oracle in tests/test_kitti_eval.py)" % (NB, NB),
               "bev_ms": round(ev_ms["bev"], 3), "bev_gpairs_per_s": round(NB * NB / (ev_ms["bev"] * 1e-3) / 1e9, 2),
               "box3d_ms": round(ev_ms["3d"], 3), "box3d_gpairs_per_s": round(NB * NB / (ev_ms["3d"] * 1e-3) / 1e9, 2),
               "note": "boxes scattered over an 80 m x 65 m scene, as KITTI frames are: most pairs are disjoint and end after the 8 + 16 corner / edge tests; "
                       "compute-bound scalar geometry (<= 16 candidate vertices per pair in LDS stripes); the reference runs this "
                       "through a numba.cuda JIT kernel plus a numba CPU pass for the 3D part"}

    if rank == 0:
        rep = train_report(headline_mode, head)
        out = {
            "metric": "images/sec (384x1280) fwd+bwd",
            "value": rep["images_per_sec"],
            "unit": "images/sec",
            "n_gpus": world, "world_size": dist.get_world_size() if dist_on else 1,
            "collective_backend": (dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else "")) if dist_on else None,
            "steps": args.steps, "warmup": max(args.warmup, 1),
            "ms_per_step": rep["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": MODES[headline_mode]["dtype"], "data": "synthetic",
            "config": {"workload": "full train step (train-mode fwd + targets + 10 losses + bwd + %sclip + AdamW + cyclic "
                                   "schedule), DLA-34 + DLAUp + MonoCon heads, batch=%d/GPU, 3x%dx%d synthetic KITTI-shaped "
                                   "(BASELINE configs[2] shape at the reference's fp32 precision: fp32 values, precision mode %s)"
                                   % ("RCCL grad all-reduce + " if world > 1 else "", B, H, W, headline_mode),
                       "global_batch": world * B, "parallelism": "dp%d" % world, "precision_mode": headline_mode},
            "roofline": rep.get("roofline"),
            "workspace_gb": rep.get("workspace_gb"),
        }
        if fwd_head is not None:
            fr = forward_report(headline_mode, fwd_head)
            fr["workload"] = "BASELINE configs[1]: eval forward only, batch=%d/GPU, fp32 values, precision mode %s" % (B, headline_mode)
            fr["steps"] = args.forward_steps
            out["forward_only"] = fr
        for mode, (tleg, fleg) in extra.items():
            blk = {"train": train_report(mode, tleg), "forward": forward_report(mode, fleg)}
            if mode == "bf16":
                blk["workload"] = ("BASELINE configs[2] as written: the same train step / eval forward with plain bf16 MFMA operands "
                                   "(fp32 accumulation, activations, master weights, BN statistics, losses).  NOT a parity path: "
                                   "on the conditioned fixtures its predictions sit ~1e-1 relative L2 and its flat gradient at "
                                   "cosine 0.92-0.97 from the fp64 reference (scratch/bf16_probe.py, DESIGN.md 3a)")
                out["mixed_precision"] = blk
            elif mode == "fp32":
                blk["workload"] = "the same train step / eval forward on the native fp32 MFMA (v_mfma_f32_32x32x2_f32)"
                out["native_fp32"] = blk
            elif mode == "bf16x3":
                blk["workload"] = "the same train step / eval forward with fp32 emulated on the bf16 matrix pipe (3-way operand split)"
                out["fp32_emulated_bf16x3"] = blk
            else:
                blk["workload"] = "the same train step / eval forward with fp32 emulated on the fp16 matrix pipe (2-way operand split)"
                out["fp32_emulated_f16x2"] = blk
        if real is not None:
            out["realistic_loop"] = real
        if feed is not None:
            out["engine_feed"] = feed
        if kfeed is not None:
            out["engine_feed_kitti"] = kfeed
        if dist_on:
            out["multi_gpu"] = {k: head[k] for k in ("per_rank_ms_per_step", "per_rank_exposed_allreduce_ms", "comm") if k in head}
            comm = head.get("comm") or {}
            out["comm_world"] = comm.get("world", 0)                         # ranks in the handle's own RCCL communicator (0: none)
            out["n_collectives"] = comm.get("collectives_per_exchange", 1 if "path" in comm else None)
            if comm.get("world"):
                out["comm_path"] = "rccl (handle-owned communicator, overlapped buckets)"
            elif comm.get("fallback_reason"):
                out["comm_path"] = "torch (FALLBACK: %s)" % comm["fallback_reason"]
            else:
                out["comm_path"] = "torch"
            # a number measured on the fallback exchange is NOT the overlapped-RCCL result; consumers key on this flag
            # (tests/test_abi.py refuses a committed report that carries it), --strict-comm turns it into a non-zero exit
            out["comm_fallback"] = bool(out["comm_path"].startswith("torch (FALLBACK"))
            out["per_rank_ms_per_step"] = head.get("per_rank_ms_per_step")
            out["per_rank_exposed_allreduce_ms"] = head.get("per_rank_exposed_allreduce_ms")
        if args.n1_json:
            try:
                with open(args.n1_json) as f:
                    n1 = json.loads([ln for ln in f.read().splitlines() if ln.strip().startswith("{")][-1])
                if n1.get("n_gpus") == 1 and n1.get("value"):
                    out["scaling_vs_n1"] = round(out["value"] / (world * float(n1["value"])), 4)
            except (OSError, ValueError, IndexError) as e:
                print("[bench] --n1-json unreadable: %s" % e, file=sys.stderr)
        if dec is not None:
            out["decode_only"] = dec
            out["kitti_eval_overlaps"] = evl
        if not args.no_cpu_baseline:
            _phase("CPU baseline (oracle: train step B=2, eval forward B=2 / B=32, decode B=64)")
            out["cpu_baseline"] = cpu_baseline(sd, H, W, args.cpu_seconds)
        if real is not None:
            _phase("input feed capacity (one DataLoader worker on the mini KITTI tree)")
            try:
                out["input_feed"] = input_feed_capacity(rep["images_per_sec"] / world)
                if out["input_feed"] is not None:
                    out["input_feed"]["device_image_kernel"] = device_image_leg(eng)
            except Exception as e:      # noqa: BLE001  (a host-side side figure must not kill the measurement)
                out["input_feed"] = {"error": str(e)[:200]}
        _phase("done")
        # ---- the verbose report (every leg, with its prose) goes to a file and to stderr; stdout carries ONE compact,
        #      numbers-only line so that every BASELINE config survives a log tail
        full_path = args.full_json
        if full_path is None:
            d = os.path.join(REPO, "gpurun_out")
            full_path = os.path.join(d if os.path.isdir(d) else REPO, "bench_full.json")
        try:
            with open(full_path, "w") as f:
                json.dump(out, f, indent=1)
        except OSError:
            full_path = None
        print("[bench full report] " + json.dumps(out), file=sys.stderr, flush=True)
        print(json.dumps(compact_line(out, full_path)), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
