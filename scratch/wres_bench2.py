"""op-level timing of the 64 -> 64 @96x320 layer in its launch kinds (plain, statistics, backward-statistics twins): tiled kernel vs conv_wres"""
import sys, os, ctypes as C, subprocess
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    for p in (os.path.join(REPO, "monocon-pytorch_amd"), REPO):
        sys.path.insert(0, p)
    from hipmonocon.engine import Engine
    e = Engine(); e.set_precision(3)
    def bench(cfg, cout=64, iters=20):
        ms = C.c_float(0); arr = (C.c_int * 1)(64)
        rc = e.lib.mc_bench_conv(e.h, 32, 96, 320, 1, arr, cout, 3, 1, cfg, iters, C.byref(ms))
        return ms.value * 1e3 if rc == 0 else float("nan")
    out = []
    for cfg in (4, 68):
        out.append("cfg %2d: %s" % (cfg, " ".join("%.1f" % bench(cfg) for _ in range(3))))
    print("%-28s %s" % (sys.argv[1], "   ".join(out)), flush=True)
else:
    for name, env in (("plain", {}), ("stats", {"MONOCON_BENCH_STATS": "1"}), ("twin (mask from y)", {"MONOCON_BENCH_BM": "1"}),
                      ("twin (stored mask)", {"MONOCON_BENCH_BM": "2"}), ("twin (stored mask + acc)", {"MONOCON_BENCH_BM": "3"})):
        subprocess.run([sys.executable, __file__, name], env=dict(os.environ, **env))
