"""Worst prediction-map deviation from the CPU oracle (fp32) for the shape-sweep cases, per precision mode."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "monocon-pytorch_amd")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from hipmonocon import synth
from hipmonocon.engine import Engine
from oracle import monocon_oracle as O
G = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
stats = np.load(os.path.join(G, "bn_calib_seed7.npz"))
sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
def rel(a, b): return float((a.double() - b.double()).abs().max() / b.double().abs().max())
for mode in ("fp32", "bf16x3"):
    os.environ["MONOCON_HIP_PRECISION"] = mode
    eng = Engine(); st = {k: v.cuda() for k, v in sd.items()}; eng.bind_state(st)
    for (B, H, W) in [(1, 64, 128), (2, 128, 512), (5, 64, 64), (2, 32, 2048), (1, 160, 1280), (1, 384, 1248)]:
        img = synth.make_batch(1000 + B + H + W, B, H, W, with_labels=False)["img"]
        with torch.no_grad():
            ref, _, _ = O.forward(sd, img)
            ref64, _, _ = O.forward({k: (v.double() if v.dtype == torch.float32 else v) for k, v in sd.items()}, img.double())
        p = eng.forward_infer(img.cuda())
        w = max(((rel(v.cpu(), ref[k]), k) for k, v in p.items()))
        w64 = max(((rel(v.cpu(), ref64[k]), k) for k, v in p.items()))
        o64 = max(((rel(ref[k], ref64[k]), k) for k in p))
        print("%-7s B%d %dx%d  vs fp32 oracle %.2e (%s)   vs fp64 oracle %.2e (%s)   [fp32 oracle vs fp64: %.2e]" % (mode, B, H, W, w[0], w[1], w64[0], w64[1], o64[0]))
