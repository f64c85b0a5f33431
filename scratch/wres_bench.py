"""op-level timing of the weight-resident conv (cfg flag 64) against the tiled shapes, B=32 full-size layers (mc_bench_conv)"""
import sys, os, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "monocon-pytorch_amd"), REPO):
    sys.path.insert(0, p)
import torch
from hipmonocon.engine import Engine
e = Engine(); e.set_precision(3)
def bench(B, H, W, cins, cout, k, cfg, iters=20):
    ms = C.c_float(0)
    arr = (C.c_int * len(cins))(*cins)
    rc = e.lib.mc_bench_conv(e.h, B, H, W, len(cins), arr, cout, k, 1, cfg, iters, C.byref(ms))
    return ms.value if rc == 0 else float("nan")
B = int(os.environ.get("WB", "32"))
for name, H, W, cins, cout in (("64->64 @96x320", 96, 320, [64], 64), ("64->576 head @96x320", 96, 320, [64], 576)):
    flop = 2.0 * B * H * W * cout * sum(cins) * 9
    for cfg in (4, 5, 8, 1, 7, 68):
        if cfg in (1, 7) and cout % 128: continue
        ts = [bench(B, H, W, cins, cout, 3, cfg) for _ in range(3)]
        t = min(ts)
        print("%-22s cfg %3d  %8.1f us  %6.1f TF fp32-equiv  (x3 = %6.0f TF fp16)  runs %s" % (name, cfg, t * 1e3, flop / t / 1e9, 3 * flop / t / 1e9, " ".join("%.1f" % (x * 1e3) for x in ts)), flush=True)
