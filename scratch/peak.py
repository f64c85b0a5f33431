import ctypes as C, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "monocon-pytorch_amd"))
from hipmonocon.engine import Engine
from hipmonocon import lib
e = Engine(); L = lib.load()
for w in (2, -2, 4, -4):
    for iters in (20000, 200000):
        t = C.c_float()
        L.mc_bench_mfma_peak(e.h, w, iters, C.byref(t))
        print("waves/simd", w, "iters", iters, "TF", round(t.value, 1), flush=True)
