"""which torch ops (and their kernels) run inside one headline train step (the loop of bench.py)"""
import sys, os, torch, collections
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "monocon-pytorch_amd"))
from hipmonocon import synth
from model import MonoConDetector
from solver import AdamW, CyclicScheduler
from torch.profiler import profile, ProfilerActivity
B, H, W = 32, 384, 1280
m = MonoConDetector(34, pretrained_backbone=False).cuda().train().set_precision("f16x2")
batch = synth.make_batch(5, 8, H, W)
rep = B // 8
gb = {"img": torch.randn(B, 3, H, W, device="cuda"),
      "label": {k: v.repeat(rep, *([1] * (v.dim() - 1)))[:B].cuda() for k, v in batch["label"].items()},
      "img_metas": {"pad_shape": [(H, W)] * B}}
opt = AdamW(m.parameters(), lr=2.25e-4, weight_decay=1e-5, betas=(0.95, 0.99), max_grad_norm=35.0)
sch = CyclicScheduler(opt, total_steps=1000)
def step():
    opt.zero_grad()
    _, loss = m(gb)
    total = sum(v for v in loss.values())
    total.backward()
    opt.step()
    sch.step()
    return total
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
c = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::") and e.cpu_parent is not None:
        # top-level aten ops only (parent is not another aten op)
        par = e.cpu_parent
        if par.name.startswith("aten::"): continue
        c[(par.name[:60], e.name)] += 1
for (par, name), n in sorted(c.items(), key=lambda kv: -kv[1]): print("%3d  %-40s under %s" % (n, name, par))
ker = collections.Counter(e.name[:110] for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA and not e.name.startswith("mc::") and "mc::" not in e.name)
for k, v in ker.most_common(): print(v, k)
