#!/usr/bin/env python
"""Golden vectors for the decode's peak filter with windows other than 3 (test_config['local_maximum_kernel']):
outputs of the REAL reference's get_local_maximum / get_topk_from_heatmap on a seeded heat map.
Runs only in the build container (needs /root/reference, read-only); the tests read tests/golden/localmax_windows.npz.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_localmax_golden.py
"""
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
sys.path.append(os.path.join(REPO, "monocon-pytorch_amd"))

import numpy as np
import torch
from utils.tensor_ops import get_local_maximum, get_topk_from_heatmap      # noqa: E402  (reference)
from hipmonocon import synth                                                # noqa: E402  (this repo)

K, B, H, W = 20, 2, 24, 44            # W % 4 == 0 on purpose: the 3x3 case would take the vector kernel, the others must not
out = {}
seed = 515
while True:                            # tie-free top-(K+1) for every window, so that torch.topk's order is the canonical one
    d = synth.make_decode_inputs(seed, B, H, W, topk=K)
    heat = torch.from_numpy(d["center_heatmap_pred"].copy())
    ok = True
    for k in (1, 3, 5, 7):
        top = torch.topk(get_local_maximum(heat, kernel=k).view(B, -1), K + 1)[0]
        ok = ok and bool((top[:, 1:] < top[:, :-1]).all())
    if ok:
        break
    seed += 1
out["seed"] = seed
for k in (1, 3, 5, 7):
    filt = get_local_maximum(heat, kernel=k)
    sc, ind, cls, ys, xs = get_topk_from_heatmap(filt, k=K)
    out["keep_packed.%d" % k] = np.packbits((filt > 0).numpy())
    out["scores.%d" % k] = sc.numpy()
    out["ind.%d" % k] = ind.numpy()
    out["cls.%d" % k] = cls.numpy()
np.savez_compressed(os.path.join(HERE, "localmax_windows.npz"), **out)
print("wrote localmax_windows.npz, seed", seed, {k: int(np.unpackbits(out["keep_packed.%d" % k]).sum()) for k in (1, 3, 5, 7)})
