#!/usr/bin/env python
"""Time every distinct fused-conv shape of the B=32 forward under each workgroup shape (GPU box)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "monocon-pytorch_amd"))
import torch
from hipmonocon.engine import Engine
from hipmonocon import lib

B = int(os.environ.get("TUNE_B", "32"))
CFG = {1: "128x128", 4: "128x64", 5: "128x64m", 6: "128x32", 7: "64x128", 8: "64x64",
       17: "w128x128", 20: "w128x64", 21: "w128x64m", 22: "w128x32", 23: "w64x128", 24: "w64x64"}
# (name, Hin, Win, [Cin], Cout, k, stride, count per forward)
LAYERS = [
    ("l0 16->16", 384, 1280, [16], 16, 3, 1, 1),
    ("l1 16->32 s2", 384, 1280, [16], 32, 3, 2, 1),
    ("s2 32->64", 192, 640, [32], 64, 3, 2, 1),
    ("s2 64->128", 96, 320, [64], 128, 3, 2, 1),
    ("s2 128->256", 48, 160, [128], 256, 3, 2, 1),
    ("s2 256->512", 24, 80, [256], 512, 3, 2, 1),
    ("64->64", 96, 320, [64], 64, 3, 1, 3),
    ("node 64+64->64", 96, 320, [64, 64], 64, 3, 1, 3),
    ("head 64->576", 96, 320, [64], 576, 3, 1, 1),
    ("128->128", 48, 160, [128], 128, 3, 1, 7),
    ("node 128+128->128", 48, 160, [128, 128], 128, 3, 1, 2),
    ("proj 128->64", 48, 160, [128], 64, 3, 1, 3),
    ("256->256", 24, 80, [256], 256, 3, 1, 7),
    ("node 256+256->256", 24, 80, [256, 256], 256, 3, 1, 1),
    ("proj 256->128", 24, 80, [256], 128, 3, 1, 2),
    ("512->512", 12, 40, [512], 512, 3, 1, 3),
    ("proj 512->256", 12, 40, [512], 256, 3, 1, 1),
    ("root 64+64->64", 96, 320, [64, 64], 64, 1, 1, 1),
    ("root 128x2->128", 48, 160, [128, 128], 128, 1, 1, 1),
    ("root 128,128,64,128->128", 48, 160, [128, 128, 64, 128], 128, 1, 1, 1),
    ("root 256x2->256", 24, 80, [256, 256], 256, 1, 1, 1),
    ("root 256,256,128,256->256", 24, 80, [256, 256, 128, 256], 256, 1, 1, 1),
    ("root 512,512,256->512", 12, 40, [512, 512, 256], 512, 1, 1, 1),
    ("project 32->64", 96, 320, [32], 64, 1, 1, 1),
    ("project 64->128", 48, 160, [64], 128, 1, 1, 1),
    ("project 128->256", 24, 80, [128], 256, 1, 1, 1),
    ("project 256->512", 12, 40, [256], 512, 1, 1, 1),
]

if os.environ.get('TUNE_NOWS'):
    CFG = {k: v for k, v in CFG.items() if k < 16}
eng = Engine()
L = lib.load()
only = sys.argv[1:] if len(sys.argv) > 1 else None
total_best = 0.0
print("%-28s %9s | " % ("layer", "GFLOP") + " ".join("%9s" % CFG[c] for c in sorted(CFG)) + " | best TF")
for name, H, W, cins, cout, k, s, cnt in LAYERS:
    if only and not any(o in name for o in only):
        continue
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    gf = 2.0 * B * Ho * Wo * cout * sum(cins) * k * k / 1e9
    coutp = (cout + (128 if cout >= 128 else 64 if cout > 32 else 32) - 1) // (128 if cout >= 128 else 64 if cout > 32 else 32) * (128 if cout >= 128 else 64 if cout > 32 else 32)
    row = []
    for c in sorted(CFG):
        bnt = int(CFG[c].split("x")[1].rstrip("m"))
        if coutp % bnt:
            row.append(None)
            continue
        ms = C.c_float()
        arr = (C.c_int * len(cins))(*cins)
        iters = max(3, min(20, int(30.0 / max(gf / 100.0, 0.05))))
        rc = L.mc_bench_conv(eng.h, B, H, W, len(cins), arr, cout, k, s, c, iters, C.byref(ms))
        row.append(ms.value if rc == 0 else None)
    best = min(v for v in row if v is not None)
    total_best += best * cnt
    print("%-28s %9.1f | " % (name, gf) + " ".join("%9s" % ("%.3f" % v if v is not None else "-") for v in row)
          + " | %6.1f  (x%d)" % (gf / best / 1e3 * 1e3 / 1e3, cnt), flush=True)
print("sum of best x count: %.2f ms" % total_best)
