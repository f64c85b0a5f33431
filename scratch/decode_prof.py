import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "monocon-pytorch_amd")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from hipmonocon import synth
from hipmonocon.engine import Engine, p2_inverse
eng = Engine()
B, K = 64, 100
d = synth.make_decode_inputs(9001, B, 96, 320, topk=K)
pred = {k: torch.from_numpy(v).to(eng.device) for k, v in d.items()}
P2 = np.stack([synth.KITTI_P2] * B)
P2d, P2i = torch.from_numpy(P2).to(eng.device), torch.from_numpy(p2_inverse(P2)).to(eng.device)
for _ in range(5): eng.decode(pred, P2d, P2i, (384, 1280), K, 0.4)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): eng.decode(pred, P2d, P2i, (384, 1280), K, 0.4)
torch.cuda.synchronize(); print("decode ms/batch", (time.perf_counter() - t0) / 50 * 1e3)
