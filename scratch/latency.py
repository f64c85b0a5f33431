import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "monocon-pytorch_amd")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from hipmonocon import synth
from hipmonocon.engine import Engine
stats = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "bn_calib_seed7.npz"))
sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
dsd = {k: v.cuda() for k, v in sd.items()}
eng = Engine(); eng.bind_state(dsd)
for B in (1, 2, 4, 8):
    img = torch.randn((B, 3, 384, 1280), device="cuda")
    for _ in range(3): eng.forward_infer(img)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 30
    for _ in range(n): eng.forward_infer(img)
    t_issue = (time.perf_counter() - t0) / n
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    pr = eng.profile_forward(iters=3)
    print("B=%d  wall %.3f ms/forward (%.0f img/s)  cpu issue %.3f ms  gpu kernels %.3f ms (conv %.3f)" % (B, dt * 1e3, B / dt, t_issue * 1e3, pr["total_ms"], pr["conv_ms"]))
