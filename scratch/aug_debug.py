import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "monocon-pytorch_amd"), REPO, os.path.join(REPO, "tests")]
import numpy as np, torch
from test_augment_device import _pair
from hipmonocon.engine import Engine
from transforms.augmentations import bgr_to_hsv, hsv_to_bgr
eng = Engine()
for seed in range(6):
    host, dev = _pair(seed, 0)
    got = eng.preprocess_augmented(dev["img"][None].cuda(), dev["img_aug"][None].cuda()).cpu()[0]
    bad = (got != host["img"]).nonzero()
    prm = dev["img_aug"].numpy()
    print("seed", seed, "flags", int(prm[2]), "bad", len(bad), "max", float((got - host["img"]).abs().max()))
    if len(bad) and int(prm[2]) in (1, 3):
        c, y, x = (int(v) for v in bad[0])
        rgb = dev["img"][y, x].numpy()
        print(" pixel", y, x, "rgb", rgb, "chan", c, "got", got[:, y, x].tolist(), "want", host["img"][:, y, x].tolist())
        img = rgb[None, None, ::-1].astype(np.float32)
        if int(prm[2]) & 2:
            img = img + prm[3]
        print(" bgr in", img.ravel().tolist(), "brightness", float(prm[3]))
        hsv = bgr_to_hsv(img)
        print(" hsv", [repr(float(v)) for v in hsv.ravel()])
        out = hsv_to_bgr(hsv)
        print(" bgr out", [repr(float(v)) for v in out.ravel()])
        mean = np.array([123.675, 116.28, 103.53]); std = np.array([58.395, 57.12, 57.375])
        print(" norm", [repr(float(np.float32((np.float64(v) - m) / s))) for v, m, s in zip(out.ravel()[::-1], mean, std)])
        break
