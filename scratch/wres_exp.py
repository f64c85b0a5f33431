"""phase removal on conv_wres_kernel (library built with EXTRA=-DMC_WRES_EXP): one process per variant"""
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
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 68
ts = [bench(32, 96, 320, [64], 64, 3, cfg) for _ in range(3)]
print("EXP=%-3s ZERO=%s cfg %d  64->64@96x320: %s us" % (os.environ.get("MONOCON_WRES_EXP", "0"), os.environ.get("MONOCON_BENCH_ZERO", "-"), cfg, " ".join("%.1f" % (x * 1e3) for x in ts)), flush=True)
