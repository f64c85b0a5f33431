import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "monocon-pytorch_amd")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from hipmonocon import synth
from hipmonocon.engine import Engine
stats = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "bn_calib_seed7.npz"))
sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
dsd = {k: v.cuda() for k, v in sd.items()}
B = 32
img = torch.randn((B, 3, 384, 1280), device="cuda")
eng = Engine()
for mode in (0, 1):
    eng.set_precision(mode); eng.bind_state(dsd)
    for _ in range(3): eng.forward_infer(img)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): eng.forward_infer(img)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    pr = eng.profile_forward(iters=3)
    print("mode", mode, "forward ms %.2f  img/s %.0f  conv_ms %.2f other_ms %.2f" % (dt * 1e3, B / dt, pr["conv_ms"], pr["other_ms"]))
