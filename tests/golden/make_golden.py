#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by running the REAL reference.

Runs only in the build container (needs /root/reference, read-only).  Nothing here
travels as code the tests execute: the tests read the .npz files this script
writes.  Inputs and parameters come from hipmonocon.synth (bit-reproducible from a
seed), so only reference OUTPUTS are stored.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import json
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")                       # reference packages win name lookups
sys.path.append(os.path.join(REPO, "monocon-pytorch_amd"))  # only ``hipmonocon`` is taken from here

import numpy as np
import torch
import torch.optim as optim
from torch.nn.utils import clip_grad_norm_

from model import MonoConDetector                           # noqa: E402  (reference)
from solver import CyclicScheduler                          # noqa: E402  (reference)
from utils.target_generator import TargetGenerator          # noqa: E402  (reference)
from hipmonocon import synth, netspec                       # noqa: E402  (this repo)

SEED = 7
torch.set_num_threads(8)
META = {"torch": torch.__version__, "threads": torch.get_num_threads(), "seed": SEED}


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrs.items()})
    print("wrote", name, "%.1f KB" % (os.path.getsize(path) / 1024))


def ref_model(sd, test_config=None, train=False, double=False):
    m = MonoConDetector(34, pretrained_backbone=False, test_config=test_config)
    m.load_state_dict(sd, strict=True)
    if double:
        m = m.double()
    return m.train() if train else m.eval()


def calibrate(seed):
    """One train-mode pass with momentum 1.0 so running stats == batch stats
    (raw random-init eval statistics make activations blow up)."""
    sd = synth.make_state_dict(seed)
    m = ref_model(sd, train=True)
    for mod in m.modules():
        if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
            mod.momentum = 1.0
    batch = synth.make_batch(seed + 100, 4, 128, 256, with_labels=False)
    with torch.no_grad():
        feat = m.neck(m.backbone(batch["img"]))[0]
        m.head._get_predictions(feat)
    out = {k: v for k, v in m.state_dict().items()
           if k.endswith("running_mean") or k.endswith("running_var")}
    return out


def strided(t, step=97):
    return t.detach().reshape(-1)[::step].clone()


def main():
    # ---------------------------------------------------------------- (0) BN calibration
    stats = calibrate(SEED)
    save("bn_calib_seed%d.npz" % SEED, **stats)
    sd = synth.make_state_dict(SEED, bn_stats={k: v.numpy() for k, v in stats.items()})

    # ---------------------------------------------------------------- (1) small-res eval forward
    b = synth.make_batch(SEED + 1, 2, 64, 128, with_labels=False)
    m = ref_model(sd)
    with torch.no_grad():
        levels = m.backbone(b["img"])
        feat = m.neck(levels)[0]
        pred = m.head.forward_test(feat)
    m64 = ref_model(sd, double=True)
    with torch.no_grad():
        pred64 = m64({"img": b["img"].double()})
    out = {"feat": feat}
    for i, l in enumerate(levels):
        out["level%d_sample" % i] = strided(l, 13)
        out["level%d_absmax" % i] = l.abs().max()
    for k, v in pred.items():
        out[k] = v
        out["f64." + k] = pred64[k]
    save("fwd_small_eval.npz", **out)

    # ---------------------------------------------------------------- (2) full-res eval forward
    b = synth.make_batch(SEED + 2, 2, 384, 1280, with_labels=False)
    with torch.no_grad():
        pred = m({"img": b["img"]})
        pred64 = m64({"img": b["img"].double()})
    out = {}
    for k, v in pred.items():
        out[k + ".sample"] = strided(v)
        out[k + ".f64sample"] = strided(pred64[k])
        out[k + ".sum"] = v.double().sum()
        out[k + ".absmax"] = v.abs().max()
    save("fwd_full_eval.npz", **out)

    # ---------------------------------------------------------------- (3) target generator
    lab = synth.make_labels(SEED + 3, 3, 384, 1280)
    # hand-made edge cases on image 2: border-clipped splat, radius 0, out-of-map keypoints
    lab["gt_bboxes"][2, 0] = [0.0, 100.0, 30.0, 380.0]       # hugging the left border
    lab["gt_bboxes"][2, 1] = [1270.0, 370.0, 1279.0, 383.0]  # tiny box in the corner -> radius 0
    lab["gt_kpts_2d"][2, 0, 0:4] = [-20.0, 50.0, 1300.0, 400.0]
    lab["gt_kpts_valid_mask"][2, 0, 0:2] = 1
    lab["mask"][2, 0:2] = 1
    lab_t = {k: torch.from_numpy(v.copy()) for k, v in lab.items()}
    tg = TargetGenerator()
    data = {"img": torch.zeros(3, 3, 384, 1280), "img_metas": {"pad_shape": [(384, 1280)] * 3}, "label": lab_t}
    T = tg(data, feat_shape=(3, 64, 96, 320))
    save("targets.npz", **{"in." + k: v for k, v in lab.items()}, **T)

    # ---------------------------------------------------------------- (4) train forward/backward
    hb, wb = 192, 384
    b = synth.make_batch(SEED + 4, 2, hb, wb)
    mt = ref_model(sd, train=True)
    pred, loss = mt(b)
    total = sum(v for v in loss.values())
    total.backward()
    out = {"total": total}
    for k, v in loss.items():
        out[k] = v if torch.is_tensor(v) else torch.tensor(float(v))
    dead = []
    for n, p in mt.named_parameters():
        if p.grad is None:
            dead.append(n)
            continue
        out["gnorm." + n] = p.grad.double().norm()
        out["gsample." + n] = strided(p.grad, 101)
    out["dead"] = np.array(dead)
    newsd = mt.state_dict()
    for k in newsd:
        if k.endswith("running_mean") or k.endswith("running_var"):
            out["buf." + k] = newsd[k]
    for k, v in pred.items():
        out["pred." + k + ".sample"] = strided(v, 31)
    # the same step in fp64: the yard-stick for gradient parity (on this B=2 train-mode-BN fixture the
    # reference's own fp32 gradients sit 5e-3..2e-2 from its fp64 gradients in the backbone)
    mt64 = ref_model(sd, train=True, double=True)
    b64 = synth.make_batch(SEED + 4, 2, hb, wb)
    b64["img"] = b64["img"].double()
    _, loss64 = mt64(b64)
    sum(v for v in loss64.values()).backward()
    for n, p in mt64.named_parameters():
        if p.grad is not None:
            out["gnorm64." + n] = p.grad.norm()
            out["gsample64." + n] = strided(p.grad, 101)
    for k, v in loss64.items():
        out["f64." + k] = v if torch.is_tensor(v) else torch.tensor(float(v))
    save("train_step.npz", **out)
    assert sorted(dead) == sorted(netspec.DEAD_PARAMS), dead

    # ---------------------------------------------------------------- (6) clip + AdamW + cyclic scheduler
    names = ["backbone.level2.tree1.conv1.weight", "neck.ida_2.up_3.weight", "head.wh_head.3.bias",
             "head.dim_head.1.weight_", "backbone.base_layer.1.weight"]
    opt = optim.AdamW(mt.parameters(), lr=2.25e-4, weight_decay=1e-5, betas=(0.95, 0.99))
    sch = CyclicScheduler(opt, total_steps=1000, target_lr_ratio=(10, 1e-4),
                          target_momentum_ratio=(0.85 / 0.95, 1.0), period_up=0.4)
    sched = []
    out = {}
    pd = dict(mt.named_parameters())
    for step in range(3):
        if step > 0:
            opt.zero_grad()
            bb = synth.make_batch(SEED + 4 + step, 2, hb, wb)
            _, loss = mt(bb)
            sum(v for v in loss.values()).backward()
        for n in names:                                   # pre-clip gradients of the sampled tensors
            out["grad%d.%s" % (step, n)] = pd[n].grad.detach().clone()
        norm = clip_grad_norm_(mt.parameters(), max_norm=35, norm_type=2.0)
        lr_used, b1_used = opt.param_groups[0]["lr"], opt.param_groups[0]["betas"][0]
        opt.step()
        sch.step()
        sched.append([lr_used, b1_used, float(norm)])
        for n in names:
            out["step%d.%s" % (step, n)] = pd[n].detach().clone()
    for extra in range(3, 12):
        sched.append([opt.param_groups[0]["lr"], opt.param_groups[0]["betas"][0], 0.0])
        opt.step(); sch.step()
    out["sched"] = np.array(sched, dtype=np.float64)
    save("adamw.npz", **out)

    # ---------------------------------------------------------------- (5) decode
    for K in (30, 100):
        seed = SEED + 50
        while True:
            d = synth.make_decode_inputs(seed, 4, 96, 320, topk=K)
            heat = torch.from_numpy(d["center_heatmap_pred"])
            hmax = torch.nn.functional.max_pool2d(heat, 3, 1, 1)
            filt = (heat * (hmax == heat).float()).view(4, -1)
            top = torch.topk(filt, K + 1)[0]
            if (top[:, 1:] < top[:, :-1]).all():
                break
            seed += 1
        mdec = ref_model(sd, test_config={"topk": K, "local_maximum_kernel": 3, "max_per_img": 30,
                                          "test_thres": 0.4})
        pd_ = {k: torch.from_numpy(v.copy()) for k, v in d.items()}
        data = {"img": torch.zeros(4, 3, 384, 1280), "img_metas": {"pad_shape": [(384, 1280)] * 4}, "calib": [synth.SynthCalib() for _ in range(4)]}
        # dense intermediates through the reference's own helpers
        from utils.tensor_ops import get_local_maximum, get_topk_from_heatmap
        filt_ref = get_local_maximum(pd_["center_heatmap_pred"], kernel=3)
        sc, ind, cls, ys, xs = get_topk_from_heatmap(filt_ref, k=K)
        b2d, b3d, labs = mdec.head._get_bboxes(data, {k: v.clone() for k, v in pd_.items()})
        out = {"seed": seed, "keep_packed": np.packbits((filt_ref > 0).numpy()),
               "scores": sc, "ind": ind, "cls": cls, "ys": ys, "xs": xs}
        for i in range(4):
            out["box2d.%d" % i] = b2d[i]
            out["box3d.%d" % i] = b3d[i]
            out["label.%d" % i] = labs[i]
        if K == 30:      # KITTI annotation dicts produced by the reference for the same decode
            data["img_metas"]["ori_shape"] = [(375, 1242)] * 4
            data["img_metas"]["sample_idx"] = [11, 12, 13, 14]
            fmt = mdec.head._get_eval_formats(data, {k: v.clone() for k, v in pd_.items()})
            for i in range(4):
                for field in ("img_bbox", "img_bbox2d"):
                    a = fmt[field][i]
                    for kk in ("alpha", "bbox", "dimensions", "location", "rotation_y", "score", "sample_idx"):
                        out["kitti.%s.%d.%s" % (field, i, kk)] = np.asarray(a[kk], dtype=np.float64)
                    out["kitti.%s.%d.name" % (field, i)] = np.array([("Pedestrian", "Cyclist", "Car").index(n) for n in a["name"]], dtype=np.int64)
        save("decode_k%d.npz" % K, **out)

    json.dump(META, open(os.path.join(HERE, "meta.json"), "w"), indent=1)
    cond_train(sd_stats=stats)
    dp_shards(sd_stats=stats)
    init_pins()
    train_full(sd_stats=stats)
    dataset_pins()


# ==================================================================== round-2 additions
def _grads(sd, batch, double):
    m = ref_model(sd, train=True, double=double)
    if double:
        batch = dict(batch)
        batch["img"] = batch["img"].double()
    pred, loss = m(batch)
    sum(v for v in loss.values()).backward()
    g = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    return g, {k: (v.detach() if torch.is_tensor(v) else torch.tensor(float(v))) for k, v in loss.items()}, m, pred


def sample_step(numel, target=2048):
    """stride that keeps <= ~2048 evenly spaced elements of a tensor (tests use the same rule)."""
    return max(1, numel // target)


def gsample(t):
    """float32 strided sample of a (possibly fp64) gradient tensor: parity is judged at 1e-4..1e-3, fp32 storage
    (6e-8) is ample and keeps a fixture under 2 MB."""
    return t.detach().reshape(-1)[::sample_step(t.numel())].float().clone()


def _tensor_errs(ga, gb):
    return {n: float((ga[n].double() - gb[n].double()).norm() / max(float(gb[n].double().norm()), 1e-30)) for n in gb}


COND_CASES = ((4, 64, 64), (8, 32, 64), (4, 64, 64), (8, 32, 64))   # (B, H, W) of fixture 0, 1, 2, 3


def cond_train(sd_stats):
    """(4b) conditioned gradient fixtures, selected free of ReLU / max-pool decision flips.

    A ReLU whose pre-activation sits within round-off of zero flips between two correct fp32
    implementations; one flip on a 24x40 map moves every upstream gradient tensor by ~4e-4 relative L2
    (measured with the reference itself), so any fixed fixture bounds gradient parity
    by sqrt(#flips / #activations), not by kernel accuracy.  Like the tie-free decode fixtures, these are
    therefore *selected*: a seed is kept only if (a) the reference's fp64 gradients move < 1e-4 on every
    tensor when the image is perturbed by 3e-7 relative noise (5 fp32 ulps: no decision within a few
    round-offs of its threshold) and (b) the reference's own fp32 run (8 threads and 1 thread) agrees with its fp64 run
    to < 2e-4 on every tensor."""
    sd = synth.make_conditioned_state_dict(SEED, bn_stats={k: v.numpy() for k, v in sd_stats.items()})
    seed, picked = 400, []
    for case, (B, H, W) in enumerate(COND_CASES):
        for _attempt in range(200):
            seed += 1
            b = synth.make_conditioned_batch(seed, B, H, W)
            g64, l64, m64, _ = _grads(sd, b, True)
            noise = torch.from_numpy(synth.uniform(seed, "cond.noise", tuple(b["img"].shape), -1.0, 1.0))
            bp = dict(b)
            bp["img"] = (b["img"].double() * (1.0 + 3e-7 * noise))
            g64p = _grads(sd, bp, True)[0]
            e_margin = max(_tensor_errs(g64p, g64).values())
            if e_margin >= 1e-4:
                print("  cond case %d seed %d rejected: perturbed-fp64 max %.1e" % (case, seed, e_margin)); continue
            g32, l32, m32, pred32 = _grads(sd, b, False)
            torch.set_num_threads(1)
            g32s = _grads(sd, b, False)[0]
            torch.set_num_threads(META["threads"])
            e32 = max(max(_tensor_errs(g32, g64).values()), max(_tensor_errs(g32s, g64).values()))
            if e32 >= 2e-4:
                print("  cond case %d seed %d rejected: fp32 max %.1e" % (case, seed, e32)); continue
            break
        else:
            raise RuntimeError("no flip-free seed found for case %d" % case)
        print("cond case %d: B=%d %dx%d seed %d  perturbed-fp64 %.1e  ref fp32-vs-fp64 %.1e" % (case, B, H, W, seed, e_margin, e32))
        out = {"seed": seed, "shape": np.array([B, H, W]), "ref32_max_err": e32, "margin_err": e_margin}
        for k in l64:
            out["f64." + k] = l64[k]
            out[k] = l32[k]
        for n in g64:
            out["gnorm64." + n] = g64[n].norm()
            out["g64." + n] = gsample(g64[n])
            out["gnorm." + n] = g32[n].double().norm()
        new64 = m64.state_dict()
        for k in new64:
            if k.endswith("running_mean") or k.endswith("running_var"):
                out["buf64." + k] = new64[k]
        save("train_cond_%d.npz" % case, **out)
        picked.append(seed)
    return picked


def train_full(sd_stats, tries=3):
    """(4c) the headline workload itself: B=2, 384x1280, train mode (batch-statistics BatchNorm), conditioned
    parameters / images, reference in fp64 -- 10 losses, updated BN buffers, per-tensor gradient norms + strided
    samples, strided samples of the 10 prediction maps.  A map of 2x16x384x1280 activations cannot be selected
    flip-free (expected flips ~ N*e/sigma >> 1), but one flip moves a gradient tensor by ~1/sqrt(N) here, so the
    screening keeps, of ``tries`` seeds, the one whose fp64 gradients move least under a 3e-7 image perturbation,
    and the fixture RECORDS per tensor (a) that margin and (b) the reference's own fp32-vs-fp64 error -- the
    yard-sticks the GPU test bounds against."""
    sd = synth.make_conditioned_state_dict(SEED, bn_stats={k: v.numpy() for k, v in sd_stats.items()})
    B, H, W = 2, 384, 1280
    best = None
    for t in range(tries):
        seed = 700 + t
        b = synth.make_conditioned_batch(seed, B, H, W)
        g64, l64, m64, p64 = _grads(sd, b, True)
        noise = torch.from_numpy(synth.uniform(seed, "cond.noise", tuple(b["img"].shape), -1.0, 1.0))
        bp = dict(b)
        bp["img"] = (b["img"].double() * (1.0 + 3e-7 * noise))
        g64p = _grads(sd, bp, True)[0]
        margin = _tensor_errs(g64p, g64)
        print("train_full seed %d: perturbed-fp64 max %.2e median %.2e" % (seed, max(margin.values()), float(np.median(list(margin.values())))), flush=True)
        if best is None or max(margin.values()) < max(best[1].values()):
            new64 = {k: v.clone() for k, v in m64.state_dict().items() if k.endswith("running_mean") or k.endswith("running_var")}
            best = (seed, margin, g64, l64, new64, {k: v.detach().clone() for k, v in p64.items()})
        del g64p, m64, p64
    seed, margin, g64, l64, new64, p64 = best
    b = synth.make_conditioned_batch(seed, B, H, W)
    g32, l32, _, _ = _grads(sd, b, False)
    e32 = _tensor_errs(g32, g64)
    print("train_full: kept seed %d; ref fp32-vs-fp64 max %.2e median %.2e" % (seed, max(e32.values()), float(np.median(list(e32.values())))))
    out = {"seed": seed, "shape": np.array([B, H, W])}
    for k in l64:
        out["f64." + k] = l64[k]
        out[k] = l32[k]
    for n in g64:
        out["gnorm64." + n] = g64[n].norm()
        out["g64." + n] = gsample(g64[n])
        out["gerr32." + n] = e32[n]
        out["gmargin." + n] = margin[n]
    for k, v in new64.items():
        out["buf64." + k] = v
    for k, v in p64.items():
        out["pred64." + k] = v.reshape(-1)[::sample_step(v.numel(), 4096)].float().clone()
    save("train_full.npz", **out)


def dp_shards(sd_stats):
    """(8) data parallelism: N-rank gradients == mean over ranks of the per-shard gradients (each loss is
    normalised by its *local* object count).  Global batch of 8 at 64x64 split into 2 shards of 4
    and into 4 shards of 2; reference fp64 and fp32, strided samples + norms of the mean gradient, and the
    per-shard losses."""
    sd = synth.make_conditioned_state_dict(SEED, bn_stats={k: v.numpy() for k, v in sd_stats.items()})
    gb = synth.make_conditioned_batch(SEED + 60, 8, 64, 64)
    out = {"seed": SEED + 60, "shape": np.array([8, 64, 64])}
    for world in (2, 4):
        per = 8 // world
        mean64, mean32 = None, None
        for r in range(world):
            sl = slice(r * per, (r + 1) * per)
            b = {"img": gb["img"][sl].clone(), "label": {k: v[sl].clone() for k, v in gb["label"].items()},
                 "img_metas": {k: v[sl] for k, v in gb["img_metas"].items()}, "calib": gb["calib"][sl]}
            g64, l64, _, _ = _grads(sd, b, True)
            g32, l32, _, _ = _grads(sd, b, False)
            for k in l64:
                out["w%d.r%d.f64.%s" % (world, r, k)] = l64[k]
            mean64 = g64 if mean64 is None else {n: mean64[n] + g64[n] for n in g64}
            mean32 = g32 if mean32 is None else {n: mean32[n] + g32[n] for n in g32}
        for n in mean64:
            out["w%d.gnorm64.%s" % (world, n)] = (mean64[n] / world).norm()
            out["w%d.g64.%s" % (world, n)] = gsample(mean64[n] / world)
            out["w%d.gerr32.%s" % (world, n)] = _tensor_errs({n: mean32[n]}, {n: mean64[n]})[n]
    save("dp_shards.npz", **out)


def dp_shards8(sd_stats):
    """(8b, round 6) the same contract for EIGHT ranks: a global batch of 16 at 64x64 in 8 shards of 2 (a rank's batch must hold
    two images: AttnBN normalises its attention vector over the batch).  Own file, so that dp_shards.npz keeps its bytes."""
    sd = synth.make_conditioned_state_dict(SEED, bn_stats={k: v.numpy() for k, v in sd_stats.items()})
    gb = synth.make_conditioned_batch(SEED + 61, 16, 64, 64)
    world, per = 8, 2
    out = {"seed": SEED + 61, "shape": np.array([16, 64, 64])}
    mean64, mean32 = None, None
    for r in range(world):
        sl = slice(r * per, (r + 1) * per)
        b = {"img": gb["img"][sl].clone(), "label": {k: v[sl].clone() for k, v in gb["label"].items()},
             "img_metas": {k: v[sl] for k, v in gb["img_metas"].items()}, "calib": gb["calib"][sl]}
        g64, l64, _, _ = _grads(sd, b, True)
        g32, l32, _, _ = _grads(sd, b, False)
        for k in l64:
            out["w%d.r%d.f64.%s" % (world, r, k)] = l64[k]
        mean64 = g64 if mean64 is None else {n: mean64[n] + g64[n] for n in g64}
        mean32 = g32 if mean32 is None else {n: mean32[n] + g32[n] for n in g32}
    for n in mean64:
        out["w%d.gnorm64.%s" % (world, n)] = (mean64[n] / world).norm()
        out["w%d.g64.%s" % (world, n)] = gsample(mean64[n] / world)
        out["w%d.gerr32.%s" % (world, n)] = _tensor_errs({n: mean32[n]}, {n: mean64[n]})[n]
    save("dp_shards8.npz", **out)


def init_pins():
    """(9) initialisers: the reference detector built under torch.manual_seed(5) --
    mean / std / min / max and a CRC-32 of the raw bytes of all 449 state_dict entries.  The product's
    ``init_weights`` consume the torch generator in the reference's order, so on the same torch build the
    tensors are bit-identical (CRC) and on any build the moments agree."""
    import zlib
    torch.manual_seed(5)
    m = MonoConDetector(34, pretrained_backbone=False)
    out = {"torch_seed": 5}
    names, crc, mom = [], [], []
    for k, v in m.state_dict().items():
        names.append(k)
        crc.append(zlib.crc32(v.detach().contiguous().numpy().tobytes()))
        f = v.detach().double().reshape(-1)
        mom.append([float(f.mean()), float(f.std()) if f.numel() > 1 else 0.0, float(f.min()), float(f.max())])
    out["names"] = np.array(names)
    out["crc32"] = np.array(crc, dtype=np.int64)
    out["moments"] = np.array(mom, dtype=np.float64)
    save("init_pins.npz", **out)


# ==================================================================== round-3 additions: input pipeline
KITTI_MINI = os.path.join(HERE, "kitti_mini")
# one line per labelled object: class, truncation, occlusion, alpha, bbox(4), h w l, x y z (camera 0, bottom centre), ry.
# Frame 000007 is built so that every filter rule of dataset/monocon_dataset.py:96-125 fires exactly once:
#   row 0 kept; row 1 occlusion 3; row 2 truncation 0.8; row 3 box height 10 < 25; row 4 depth 80 > 65; (DontCare: not an
#   object row at all); row 5 kept, partly outside the image (keypoints off-frame); row 6 depth 1.2 < 2.
MINI_LABELS = {
    "000007": [
        "Car 0.00 0 -1.58 587.01 173.33 614.12 200.12 1.65 1.67 3.64 -0.65 1.71 46.70 -1.59",
        "Pedestrian 0.00 3 0.21 423.17 173.67 433.17 224.03 1.60 0.38 0.30 -5.87 1.63 23.11 -0.03",
        "Cyclist 0.80 1 1.92 0.00 192.37 200.17 374.00 1.72 0.78 1.71 -4.61 1.70 5.10 1.19",
        "Car 0.00 0 1.63 700.10 180.00 720.55 190.00 1.50 1.60 3.90 6.10 1.60 60.00 1.73",
        "Car 0.00 1 -1.61 600.00 160.00 640.00 195.00 1.55 1.62 3.70 1.20 1.50 80.00 -1.60",
        "DontCare -1 -1 -10 503.89 169.71 590.61 190.13 -1 -1 -1 -1000 -1000 -1000 -10",
        "Car 0.30 0 -2.10 1100.00 150.00 1241.00 330.00 1.48 1.58 4.10 4.70 1.55 6.80 -1.50",
        "Pedestrian 0.00 0 0.10 500.00 100.00 560.00 370.00 1.75 0.60 0.80 0.10 1.60 1.20 0.10",
    ],
    "000011": [
        "Cyclist 0.00 0 -1.70 676.60 163.95 688.98 193.93 1.86 0.60 2.02 4.59 1.32 45.84 -1.60",
        "Car 0.10 2 1.55 280.38 185.10 344.90 215.59 1.49 1.76 4.01 -15.71 2.16 38.26 1.16",
        "DontCare -1 -1 -10 365.14 169.00 405.25 184.84 -1 -1 -1 -1000 -1000 -1000 -10",
    ],
}
MINI_SHAPES = {"000007": (375, 1242), "000011": (370, 1224)}


def write_kitti_mini():
    """a two-frame KITTI tree: calibration in the benchmark's text layout (KITTI-typical numbers, jittered per frame),
    the labels above, and smooth synthetic PNG frames (a few KB each)"""
    from PIL import Image
    for sub in ("image_2", "calib", "label_2"):
        os.makedirs(os.path.join(KITTI_MINI, "training", sub), exist_ok=True)
    os.makedirs(os.path.join(KITTI_MINI, "ImageSets"), exist_ok=True)
    with open(os.path.join(KITTI_MINI, "ImageSets", "val.txt"), "w") as f:
        f.write("\n".join(sorted(MINI_LABELS)) + "\n")
    for k, (pid, lines) in enumerate(sorted(MINI_LABELS.items())):
        H, W = MINI_SHAPES[pid]
        yy, xx = np.mgrid[0:H, 0:W]
        img = np.stack([(xx * 255 // W + 17 * k) % 256, (yy * 255 // H) % 256, ((xx // 64 + yy // 32) * 40 + 60 * k) % 256], -1)
        Image.fromarray(img.astype(np.uint8)).save(os.path.join(KITTI_MINI, "training", "image_2", pid + ".png"), optimize=True)
        fx, cx, cy = 721.5377 - 3.7 * k, 609.5593 + 2.2 * k, 172.854 - 1.1 * k
        def P(bx, by=0.0, bz=0.0):
            return "%.6e %.6e %.6e %.6e %.6e %.6e %.6e %.6e %.6e %.6e %.6e %.6e" % (fx, 0, cx, bx, 0, fx, cy, by, 0, 0, 1, bz)
        calib = ["P0: " + P(0.0), "P1: " + P(-387.5744), "P2: " + P(44.85728, 0.2163791, 0.002745884), "P3: " + P(-339.5242, 2.199936, 0.002729905),
                 "R0_rect: 9.999239e-01 9.837760e-03 -7.445048e-03 -9.869795e-03 9.999421e-01 -4.278459e-03 7.402527e-03 4.351614e-03 9.999631e-01",
                 "Tr_velo_to_cam: 7.533745e-03 -9.999714e-01 -6.166020e-04 -4.069766e-03 1.480249e-02 7.280733e-04 -9.998902e-01 -7.631618e-02 9.998621e-01 7.523790e-03 1.480755e-02 -2.717806e-01",
                 "Tr_imu_to_velo: 9.999976e-01 7.553071e-04 -2.035826e-03 -8.086759e-01 -7.854027e-04 9.998898e-01 -1.482298e-02 3.195559e-01 2.024406e-03 1.482454e-02 9.998881e-01 -7.997231e-01"]
        with open(os.path.join(KITTI_MINI, "training", "calib", pid + ".txt"), "w") as f:
            f.write("\n".join(calib) + "\n")
        with open(os.path.join(KITTI_MINI, "training", "label_2", pid + ".txt"), "w") as f:
            f.write("\n".join(lines) + "\n")


def dataset_pins():
    """(10) the reference's KITTICalibration / KITTISingleObject / KITTIMultiObjects (utils/data_classes.py, importable)
    on the mini tree: calibration matrices and derived intrinsics, and per object -- after convert_cam(0 -> 2) and
    convert_yaw(global -> local), in the order dataset/monocon_dataset.py:84-125 reads them -- class, box, location,
    dimensions, local yaw, projected centre (+ depth), the 9 projected keypoints with their in-front flags, level.
    (dataset/, transforms/ themselves need cv2 and cannot be imported: the label loop is NOT pinned by this.)"""
    from utils.data_classes import KITTICalibration, KITTIMultiObjects            # noqa: E402  (reference)
    write_kitti_mini()
    out = {}
    for pid in sorted(MINI_LABELS):
        cf = os.path.join(KITTI_MINI, "training", "calib", pid + ".txt")
        lf = os.path.join(KITTI_MINI, "training", "label_2", pid + ".txt")
        calib = KITTICalibration(cf)
        for k in ("P0", "P1", "P2", "P3", "R0", "V2C", "C2V", "I2V", "V2I"):
            out["%s.calib.%s" % (pid, k)] = getattr(calib, k)
        out["%s.calib.intr" % pid] = np.array([calib.cu, calib.cv, calib.fu, calib.fv, calib.tx, calib.ty], dtype=np.float64)
        objs = KITTIMultiObjects.get_objects_from_label(lf, calib)
        out["%s.n" % pid] = len(objs)
        objs.convert_cam(src_cam=0, dst_cam=2)
        objs.convert_yaw(src_type="global", dst_type="local")
        for i, o in enumerate(objs):
            tag = "%s.obj%d." % (pid, i)
            out[tag + "cls"] = o.cls_num
            out[tag + "occ_trunc_level"] = np.array([o.occlusion, o.truncation, o.level], dtype=np.float64)
            out[tag + "box2d"] = o.box2d.copy()
            out[tag + "box3d"] = np.concatenate([o.loc, o.dim, np.array([o.ry])], axis=0).astype(np.float64)
            pc = o.projected_center
            out[tag + "center"] = np.asarray(pc, dtype=np.float64).copy()
            kp = o.projected_kpts
            out[tag + "kpts"] = np.zeros((0, 3)) if kp is None else np.asarray(kp, dtype=np.float64)
        info = objs.original_objects.info_dict
        for k, v in info.items():
            if k != "name":
                out["%s.info.%s" % (pid, k)] = np.asarray(v, dtype=np.float64)
    save("kitti_objects.npz", **out)



def ref_checkpoint(sd_stats):
    """A checkpoint in the reference's own layout (engine/base_engine.py:155-189) WITHOUT its 235 MB of tensors.

    Real reference objects produce it: MonoConDetector (parameters = the seed-7 synthetic state dict), torch.optim.AdamW
    and the reference's CyclicScheduler as MonoconEngine.build_solver makes them (engine/monocon_engine.py:35-55), two
    optimizer + scheduler steps on seeded synthetic gradients.  The two optimizer steps run with the group's lr forced to
    0 for the duration of step() (restored before scheduler.step()): the moments and step counts become realistic while
    the parameters stay BIT-identical to the synthetic state dict -- so a loader test can (a) rebuild every tensor from
    the seed and (b) expect the eval forward of the loaded model to equal fwd_small_eval.npz.
    What is stored: the nested dict exactly as torch.save would get it, every tensor replaced by {"__t__": i} with its
    shape / dtype / fp64 sum / strided samples in the .npz.  engine_attrs: the attributes BaseEngine.__init__ sets that
    survive its filter (base_engine.py:171-174 -- the missing comma there also lets `test_dataset` through: a pickled
    MonoConDataset, which cannot be constructed here (transforms import cv2); the loader test adds a stand-in object of
    that class path and one of an unknown class path instead)."""
    sd = synth.make_state_dict(SEED, bn_stats=sd_stats)
    m = ref_model(sd, train=True)
    opt = optim.AdamW(m.parameters(), lr=2.25e-4, weight_decay=1e-5, betas=(0.95, 0.99))
    sch = CyclicScheduler(opt, total_steps=1000)
    betas_used, lrs = [], []
    for step in range(2):
        for n, p in m.named_parameters():
            if n in netspec.DEAD_PARAMS:
                p.grad = None
            else:
                p.grad = torch.from_numpy(synth.normalish(9000 + step, n, tuple(p.shape), 0.0, 1e-3).astype(np.float32))
        g = opt.param_groups[0]
        betas_used.append(g["betas"][0]); lrs.append(g["lr"])
        keep = g["lr"]
        g["lr"] = 0.0
        opt.step()
        g["lr"] = keep
        sch.step()
    for k, v in m.state_dict().items():            # parameters untouched (lr 0), buffers untouched (no forward ran)
        assert torch.equal(v, sd[k].to(v.dtype).reshape(v.shape)), k
    engine_dict = {
        "engine_attrs": {"version": "v1.0.3", "description": "MonoCon Default Configuration", "epochs": 3, "target_epochs": 200,
                         "global_iters": 931, "log_period": 5, "val_period": 0, "root": "./exps/ref", "writer_dir": "./exps/ref/tf_logs",
                         "weight_dir": "./exps/ref/checkpoints", "epoch_times": [101.5, 99.25], "entire_losses": [12.5, 11.75]},
        "state_dict": {"model": m.state_dict(), "optimizer": opt.state_dict(), "scheduler": sch.state_dict()},
    }
    tensors = []

    def strip(o):
        if torch.is_tensor(o):
            tensors.append(o)
            return {"__t__": len(tensors) - 1}
        if isinstance(o, dict):
            return {"__dict__": [[strip_key(k), strip(v)] for k, v in o.items()]}       # order + non-string keys survive JSON
        if isinstance(o, (list, tuple)):
            return {"__%s__" % type(o).__name__: [strip(v) for v in o]}
        if isinstance(o, (int, float, str, bool)) or o is None:
            return o
        raise TypeError("unexpected %r in the checkpoint" % type(o))

    def strip_key(k):
        return {"__int__": k} if isinstance(k, int) else k

    skeleton = strip(engine_dict)
    out = {"skeleton": np.frombuffer(json.dumps(skeleton).encode(), dtype=np.uint8), "n_tensors": len(tensors),
           "betas1": np.asarray(betas_used), "lrs": np.asarray(lrs), "meta": json.dumps(META)}
    for i, t in enumerate(tensors):
        out["t%d.shape" % i] = np.asarray(t.shape, dtype=np.int64)
        out["t%d.dtype" % i] = str(t.dtype)
        out["t%d.sum" % i] = float(t.double().sum())
        flat = t.reshape(-1)
        out["t%d.samples" % i] = flat[::max(1, flat.numel() // 16)][:16].double().numpy() if flat.numel() else np.zeros(0)
    save("ref_checkpoint.npz", **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "dataset":
        dataset_pins()
    elif len(sys.argv) > 1 and sys.argv[1] == "train_full":
        _stats = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(HERE, "bn_calib_seed%d.npz" % SEED)).items()}
        train_full(sd_stats=_stats)
    elif len(sys.argv) > 1 and sys.argv[1] == "ref_checkpoint":
        _stats = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(HERE, "bn_calib_seed%d.npz" % SEED)).items()}
        ref_checkpoint(sd_stats=_stats)
    elif len(sys.argv) > 1 and sys.argv[1] == "dp_shards8":
        _stats = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(HERE, "bn_calib_seed%d.npz" % SEED)).items()}
        dp_shards8(sd_stats=_stats)
    elif len(sys.argv) > 1 and sys.argv[1] == "round2":
        _stats = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(HERE, "bn_calib_seed%d.npz" % SEED)).items()}
        cond_train(sd_stats=_stats)
        dp_shards(sd_stats=_stats)
        init_pins()
    else:
        main()
