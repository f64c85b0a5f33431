"""Where does a conv launch's time go?  Times fixed-shape launches (mc_bench_conv, precision mode 3) with the library
variants of scratch/exp/ (lib_base.so and -DMC_EXP_NO_{MFMA,BLOAD,ALOAD,STAGE,EPI}: the named phase compiled out, results
WRONG) -- the difference to the base is what that phase costs in situ (DVFS included)."""
import ctypes as C, os, sys, shutil
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "monocon-pytorch_amd"))
import torch
from hipmonocon import lib as L
variant = sys.argv[1]
L.LIB_PATH = os.path.join(ROOT, "scratch", "exp", "lib_%s.so" % variant)
from hipmonocon.engine import Engine
eng = Engine()
eng.set_precision(int(os.environ.get("PREC", "3")))
LAYERS = [("64->64 @96x320", 96, 320, [64], 64, 3, 1, (4, 5, 8)),
          ("128->128 @48x160", 48, 160, [128], 128, 3, 1, (7, 4, 1)),
          ("256->256 @24x80", 24, 80, [256], 256, 3, 1, (1, 7)),
          ("head 64->576", 96, 320, [64], 576, 3, 1, (5,)),
          ("root1x1 256->128", 48, 160, [128, 128], 128, 1, 1, (7, 8))]
for name, H, W, cins, cout, k, s, cfgs in LAYERS:
    for cfg in cfgs:
        sc = (C.c_int * len(cins))(*cins)
        ms = C.c_float(0)
        rc = eng.lib.mc_bench_conv(eng.h, 32, H, W, len(cins), sc, cout, k, s, cfg, 30, C.byref(ms))
        gf = 2.0 * 32 * (H // s) * (W // s) * cout * sum(cins) * k * k / 1e9
        print("%-9s %-22s cfg %2d  %8.1f us  %7.1f TF-equiv" % (variant, name, cfg, ms.value * 1e3, gf / ms.value / 1e3 if rc == 0 and ms.value > 0 else 0), flush=True)
