"""Soak: 300 optimiser steps at the headline shape in f16x2 and fp32 from the same start; the loss curves must track each
other (same data every step: the loss falls monotonically at first) and stay finite."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "monocon-pytorch_amd")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from hipmonocon import synth
from model import MonoConDetector
from solver import AdamW, CyclicScheduler
B, H, W = 8, 384, 1280
stats = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "bn_calib_seed7.npz"))
sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
curves = {}
for prec in ("f16x2+1", "f16x2+2", "f16x2+3", "fp32+1", "fp32+2", "fp32+3", "bf16x3+1", "bf16x3+2"):
    m = MonoConDetector(34, pretrained_backbone=False); m.load_state_dict(sd); m = m.cuda().train(); m.set_precision(prec.split('+')[0])
    if '+' in prec:
        with torch.no_grad(): m.backbone.level2.tree1.conv1.weight[int(prec.split('+')[1]), 0, 0, 0] *= 1.0 + 1e-6      # one weight, one part in a million
    opt = AdamW(m.parameters(), lr=2.25e-4, weight_decay=1e-5, betas=(0.95, 0.99), max_grad_norm=35.0)
    sch = CyclicScheduler(opt, total_steps=400)
    batches = []
    for i in range(4):
        b = synth.make_batch(600 + i, B, H, W)
        batches.append({"img": b["img"].cuda(), "label": {k: v.cuda() for k, v in b["label"].items()}, "img_metas": {"pad_shape": [(H, W)] * B}})
    hist = []
    t0 = time.perf_counter()
    for it in range(200):
        opt.zero_grad(); _, loss = m(batches[it % 4]); t = sum(loss.values()); t.backward(); opt.step(); sch.step()
        if it % 10 == 0: hist.append(float(t))
    torch.cuda.synchronize()
    curves[prec] = hist
    print(prec, "%.1f ms/step" % ((time.perf_counter() - t0) / 200 * 1e3), " ".join("%.3f" % v for v in hist))
    del m, opt
print("all finite", all(np.isfinite(v).all() for v in curves.values()))
