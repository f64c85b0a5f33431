"""which torch ops (and their kernels) run inside one headline train step"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "monocon-pytorch_amd"))
from hipmonocon import synth
from model import MonoConDetector
from torch.profiler import profile, ProfilerActivity
B, H, W = 32, 384, 1280
m = MonoConDetector(34, pretrained_backbone=False).cuda().train().set_precision("f16x2")
batch = synth.make_batch(5, 8, H, W)
rep = B // 8
gb = {"img": torch.randn(B, 3, H, W, device="cuda"),
      "label": {k: v.repeat(rep, *([1] * (v.dim() - 1)))[:B].cuda() for k, v in batch["label"].items()},
      "img_metas": {"pad_shape": [(H, W)] * B}}
opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
def step():
    opt.zero_grad()
    _, loss = m(gb)
    total = sum(v for v in loss.values())
    total.backward()
    opt.step()
    return total
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=60, max_name_column_width=90))
ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA and ("elementwise" in e.name or "fill" in e.name.lower())]
import collections
c = collections.Counter(e.name[:200] for e in ev)
for k, v in c.most_common(): print(v, k)
