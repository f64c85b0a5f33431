"""mode 4 run-to-run determinism, node by node (activations and node gradients; MONOCON_HIP_GRAD_POOL=0)"""
import sys, os, ctypes as C
os.environ["MONOCON_HIP_GRAD_POOL"] = "0"
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(REPO, "monocon-pytorch_amd"), REPO):
    sys.path.insert(0, p)
import numpy as np, torch
from hipmonocon import synth
from model import MonoConDetector
mode = sys.argv[1]; NB = int(sys.argv[2])
stats = np.load(os.path.join(REPO, "tests", "golden", "bn_calib_seed7.npz"))
sd = synth.make_state_dict(7, bn_stats={k: stats[k] for k in stats.files})
batch = synth.make_conditioned_batch(11, NB, 384, 1280)
batch = {"img": batch["img"].cuda(), "label": {k: v.cuda() for k, v in batch["label"].items()}, "img_metas": batch["img_metas"]}
def node(eng, i, which):
    dims = (C.c_int * 4)()
    if eng.lib.mc_train_debug_node(eng.h, i, which, None, dims, None): return None
    t = torch.empty(tuple(dims), dtype=torch.float32, device="cuda")
    rc = eng.lib.mc_train_debug_node(eng.h, i, which, C.c_void_p(t.data_ptr()), dims, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return t.cpu() if rc == 0 else None
m = MonoConDetector(34, pretrained_backbone=False); m.load_state_dict(sd, strict=True)
m = m.cuda().train().set_precision(mode)
runs = []
for it in range(3):
    m.zero_grad(set_to_none=True)
    m.load_state_dict(sd, strict=True)       # (BN buffers back to the start)
    _, loss = m(batch); sum(loss.values()).backward(); torch.cuda.synchronize()
    eng = m._rt.engine
    acts, grads, i = [], [], 0
    while True:
        a = node(eng, i, 0)
        if a is None: break
        acts.append(a); grads.append(node(eng, i, 1)); i += 1
    runs.append((acts, grads, {k: float(v.detach()) for k, v in loss.items()}))
for r in (1, 2):
    print("run 0 vs %d: losses equal %s" % (r, runs[0][2] == runs[r][2]))
    for i in range(len(runs[0][0])):
        ea = torch.equal(runs[0][0][i], runs[r][0][i])
        g0, g1 = runs[0][1][i], runs[r][1][i]
        eg = None if g0 is None or g1 is None else torch.equal(g0, g1)
        if not ea or eg is False:
            nd = int((g0 != g1).sum()) if eg is False else 0
            print("  node %2d %-20s act equal %s  grad equal %s (%d elements differ, max %.3e)" % (i, tuple(runs[0][0][i].shape), ea, eg, nd,
                  float((g0 - g1).abs().max()) if eg is False else 0.0))
