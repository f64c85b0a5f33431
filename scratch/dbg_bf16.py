import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "monocon-pytorch_amd")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
from hipmonocon import synth
from hipmonocon.engine import Engine
def rel(a, b): return (float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)), float((a - b).norm() / b.norm().clamp_min(1e-30)))
stats = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "bn_calib_seed7.npz"))
sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
dsd = {k: v.cuda() for k, v in sd.items()}
img = synth.make_batch(8, 2, 64, 128, with_labels=False)["img"].cuda()
eng = Engine(); eng.bind_state(dsd)
lv32 = [t.clone() for t in eng.backbone_forward(img)]
f32 = eng.neck_forward(lv32)[0].clone() if isinstance(eng.neck_forward(lv32), tuple) else eng.neck_forward(lv32).clone()
p32 = {k: v.clone() for k, v in eng.forward_infer(img).items()}
eng.set_precision(1); eng.bind_state(dsd)
lv16 = eng.backbone_forward(img)
for i, (a, b) in enumerate(zip(lv16, lv32)): print("level", i, rel(a, b))
f16 = eng.neck_forward(lv32); f16 = f16[0] if isinstance(f16, tuple) else f16
print("neck (fp32 levels in)", rel(f16, f32))
p16 = eng.forward_infer(img)
for k in p32: print(k, rel(p16[k], p32[k]))
