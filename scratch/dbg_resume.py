import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "monocon-pytorch_amd")); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
from conftest import load_golden
from hipmonocon import synth
from model import MonoConDetector
from solver import AdamW, CyclicScheduler
stats = load_golden("bn_calib_seed7.npz")
sd0 = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
def to_cuda(b):
    d = dict(b); d["img"] = b["img"].cuda(); d["label"] = {k: v.cuda() for k, v in b["label"].items()}; return d
batches = [to_cuda(synth.make_batch(300 + i, 2, 96, 160)) for i in range(5)]
def make():
    m = MonoConDetector(34, pretrained_backbone=False); m.load_state_dict(sd0, strict=True); m = m.cuda().train()
    opt = AdamW(m.parameters(), lr=2.25e-4, weight_decay=1e-5, betas=(0.95, 0.99), max_grad_norm=35.0)
    return m, opt, CyclicScheduler(opt, total_steps=50)
def run(m, opt, sch, bs, tag=""):
    for b in bs:
        opt.zero_grad(); _, loss = m(b); t = sum(loss.values()); t.backward()
        g = torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None])
        print(tag, "loss %.9f gnorm %.9f lr %.6e b1 %.6f" % (float(t), float(g.double().norm()), opt.param_groups[0]["lr"], opt.param_groups[0]["betas"][0]))
        opt.step(); sch.step()
def snap(m, opt):
    ps = torch.cat([p.detach().flatten() for p in m.parameters()]).double()
    bs = torch.cat([b.detach().flatten().double() for b in m.buffers()])
    ms = torch.cat([opt.state[p]["exp_avg"].flatten() for p in m.parameters() if p in opt.state]).double() if len(opt.state) else torch.zeros(1)
    vs = torch.cat([opt.state[p]["exp_avg_sq"].flatten() for p in m.parameters() if p in opt.state]).double() if len(opt.state) else torch.zeros(1)
    return float(ps.sum()), float(bs.sum()), float(ms.sum()), float(vs.sum()), float(ps.abs().sum())
mb, ob, sb = make(); run(mb, ob, sb, batches[:3], "b")
ck = {"model": {k: v.cpu().clone() for k, v in mb.state_dict().items()}, "optimizer": ob.state_dict(), "scheduler": sb.state_dict()}
import copy; ck = copy.deepcopy(ck)
mc, oc, sc = make(); mc.load_state_dict(ck["model"]); oc.load_state_dict(ck["optimizer"]); sc.load_state_dict(ck["scheduler"])
print("c after load", snap(mc, oc))
md, od, sdd = make(); run(md, od, sdd, batches[:1], "d0")
md.load_state_dict(ck["model"]); od.load_state_dict(ck["optimizer"]); sdd.load_state_dict(ck["scheduler"])
print("d after load", snap(md, od))
run(mc, oc, sc, batches[3:4], "c"); print("c after step4", snap(mc, oc))
run(md, od, sdd, batches[3:4], "d"); print("d after step4", snap(md, od))
