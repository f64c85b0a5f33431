"""Counter-based synthetic data: parameters, images, KITTI-shaped labels, heatmaps.

Everything is a pure function of ``(seed, stream name, element index)`` through a
splitmix64 hash and exact integer->float arithmetic only (no libm calls), so the
same arrays are regenerated bit-for-bit in the build container (where the
reference is imported to make goldens) and on the GPU box (where only the
goldens travel).  Shapes follow SURVEY.md §8(b)/(d): ``collate_fn`` layout of
reference dataset/monocon_dataset.py:160-200.
"""
import math
import zlib
import numpy as np

from . import netspec

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
KITTI_P2 = np.array(
    [[721.5377, 0.0, 609.5593, 44.85728],
     [0.0, 721.5377, 172.854, 0.2163791],
     [0.0, 0.0, 1.0, 0.002745884]], dtype=np.float32)


def _splitmix64(x):
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _stream_base(seed, name):
    h = zlib.crc32(name.encode()) & 0xFFFFFFFF
    return _splitmix64(np.array([(int(seed) << 32) ^ h], dtype=np.uint64))[0]


def bits(seed, name, n, lane=0):
    """n uint64 hashes of stream (seed, name); ``lane`` selects an independent substream."""
    base = _stream_base(seed, name)
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(8) + np.uint64(lane)
        return _splitmix64((idx + base) & _M64)


def uniform(seed, name, shape, lo=0.0, hi=1.0, lane=0):
    n = int(np.prod(shape)) if len(shape) else 1
    u = (bits(seed, name, n, lane) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return (lo + (hi - lo) * u).reshape(shape)


def normalish(seed, name, shape, mean=0.0, std=1.0):
    """Irwin-Hall(4) 'normal': sum of four uniforms, unit variance, exact arithmetic."""
    n = int(np.prod(shape)) if len(shape) else 1
    acc = np.zeros(n, dtype=np.float64)
    for lane in range(4):
        acc += (bits(seed, name, n, lane) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    z = (acc - 2.0) * 1.7320508075688772  # var of sum = 4/12
    return (mean + std * z).reshape(shape)


def integers(seed, name, shape, lo, hi, lane=0):
    """Integers in [lo, hi)."""
    n = int(np.prod(shape)) if len(shape) else 1
    b = bits(seed, name, n, lane) >> np.uint64(16)
    return (lo + (b % np.uint64(hi - lo)).astype(np.int64)).reshape(shape)


# --------------------------------------------------------------------------- parameters
def make_state_dict(seed=0, bn_stats=None, as_torch=True):
    """All 449 entries.  Distributions are chosen for *test power* (non-degenerate
    outputs), not to imitate the reference initialisers (those live in the product
    modules' ``init_weights``).  ``bn_stats``: optional mapping key -> array that
    overrides running_mean / running_var (calibrated statistics fixture)."""
    out = {}
    for key, shape, dt, role in netspec.state_fields():
        if dt == "i64":
            arr = np.zeros(shape, dtype=np.int64)
        elif key.endswith("running_mean"):
            arr = normalish(seed, key, shape, 0.0, 0.1)
        elif key.endswith("running_var"):
            arr = uniform(seed, key, shape, 0.5, 1.5)
        elif key.endswith(".weight_"):
            arr = normalish(seed, key, shape, 1.0, 0.1)
        elif key.endswith(".bias_"):
            arr = normalish(seed, key, shape, 0.0, 0.1)
        elif len(shape) == 4 and shape[1] == 1 and shape[2] == 4:      # depthwise deconv
            f = np.array([0.25, 0.75, 0.75, 0.25])
            arr = np.outer(f, f)[None, None] * (1.0 + normalish(seed, key, shape, 0.0, 0.05))
        elif len(shape) == 4:                                            # conv weight OIHW
            o, i, kh, kw = shape
            if key.startswith("head.") and kh == 1 and "attention" not in key:
                std = 0.08                                               # output 1x1 convs
            elif "attention" in key:
                std = math.sqrt(2.0 / shape[0]) * 0.5
            else:
                std = math.sqrt(2.0 / (kh * kw * i)) * 0.9                 # fan-in scaled
            arr = normalish(seed, key, shape, 0.0, std)
        elif key.startswith("head.") and key.endswith(".bias"):
            arr = normalish(seed, key, shape, 0.0, 0.05)
            if key in ("head.heatmap_head.3.bias", "head.kpt_heatmap_head.3.bias"):
                arr = arr - 2.19
            if key == "head.dim_head.3.bias":
                arr = arr + 1.5          # keep predicted dimensions away from 0 (dim loss divides by them)
        elif key.endswith(".weight"):                                    # BN gamma
            arr = normalish(seed, key, shape, 1.0, 0.1)
        elif key.endswith(".bias"):                                      # BN beta
            arr = normalish(seed, key, shape, 0.0, 0.1)
        else:
            raise KeyError(key)
        if dt == "f32":
            arr = arr.astype(np.float32)
        if bn_stats is not None and key in bn_stats:
            arr = np.asarray(bn_stats[key]).astype(arr.dtype).reshape(shape)
        out[key] = arr
    if as_torch:
        import torch
        return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in out.items()}
    return out


# --------------------------------------------------------------------------- inputs
def make_image(seed, batch, height, width):
    """Normalised-image stand-in, N(0,1)-like, (B,3,H,W) float32 NCHW."""
    return normalish(seed, "img", (batch, 3, height, width)).astype(np.float32)


def make_labels(seed, batch, height, width, max_objs=30, num_kpts=9, min_objs=1, max_gen=8):
    """KITTI-shaped label dict of float32 arrays."""
    L = {
        "gt_bboxes": np.zeros((batch, max_objs, 4), np.float32),
        "gt_labels": np.zeros((batch, max_objs), np.float32),
        "gt_bboxes_3d": np.zeros((batch, max_objs, 7), np.float32),
        "gt_labels_3d": np.zeros((batch, max_objs), np.float32),
        "centers2d": np.zeros((batch, max_objs, 2), np.float32),
        "depths": np.zeros((batch, max_objs), np.float32),
        "gt_kpts_2d": np.zeros((batch, max_objs, num_kpts * 2), np.float32),
        "gt_kpts_valid_mask": np.zeros((batch, max_objs, num_kpts), np.float32),
        "mask": np.zeros((batch, max_objs), np.float32),
    }
    nobj = integers(seed, "lab.n", (batch,), min_objs, max_gen + 1)
    typ_dims = np.array([[0.84, 1.76, 0.66], [1.76, 1.74, 0.60], [3.88, 1.53, 1.63]])  # l,h,w
    for b in range(batch):
        n = int(nobj[b])
        tag = "lab.%d." % b
        cls = integers(seed, tag + "cls", (n,), 0, 3)
        cx = uniform(seed, tag + "cx", (n,), 64.0, width - 64.0)
        cy = uniform(seed, tag + "cy", (n,), 96.0, height - 32.0)
        bw = uniform(seed, tag + "bw", (n,), 24.0, 256.0)
        bh = uniform(seed, tag + "bh", (n,), 25.0, 192.0)
        x1 = np.clip(cx - bw / 2, 0, width - 1); x2 = np.clip(cx + bw / 2, 0, width - 1)
        y1 = np.clip(cy - bh / 2, 0, height - 1); y2 = np.clip(cy + bh / 2, 0, height - 1)
        depth = uniform(seed, tag + "z", (n,), 2.0, 65.0)
        yaw = uniform(seed, tag + "yaw", (n,), -np.pi, np.pi)
        dims = typ_dims[cls] * (1.0 + 0.1 * normalish(seed, tag + "dim", (n, 3)))
        loc = np.stack([(cx - KITTI_P2[0, 2]) * depth / KITTI_P2[0, 0],
                        (cy - KITTI_P2[1, 2]) * depth / KITTI_P2[1, 1], depth], 1)
        # 9 keypoints: 8 jittered box-corner-like points + centre; some fall outside
        kj = uniform(seed, tag + "kpt", (n, num_kpts, 2), -0.7, 0.7)
        kx = cx[:, None] + kj[..., 0] * bw[:, None]
        ky = cy[:, None] + kj[..., 1] * bh[:, None]
        kv = integers(seed, tag + "kv", (n, num_kpts), 0, 10)
        kvalid = np.where(kv == 0, 0.0, np.where(kv < 6, 1.0, 2.0))
        L["gt_bboxes"][b, :n] = np.stack([x1, y1, x2, y2], 1)
        L["gt_labels"][b, :n] = cls
        L["gt_labels_3d"][b, :n] = cls
        L["gt_bboxes_3d"][b, :n] = np.concatenate([loc, dims, yaw[:, None]], 1)
        L["centers2d"][b, :n] = np.stack([cx, cy], 1)
        L["depths"][b, :n] = depth
        L["gt_kpts_2d"][b, :n] = np.stack([kx, ky], -1).reshape(n, -1)
        L["gt_kpts_valid_mask"][b, :n] = kvalid
        L["mask"][b, :n] = 1.0
    return L


class SynthCalib:
    """Minimal stand-in for reference utils/data_classes.py:10 KITTICalibration: the
    hot path only reads ``.P2`` (monocon_heads.py:501,543)."""

    def __init__(self, P2=None):
        self.P2 = KITTI_P2.copy() if P2 is None else np.asarray(P2, np.float32)
        self.P0 = self.P2.copy()
        self.V2C = np.eye(4, dtype=np.float32)[:3]


def make_batch(seed, batch, height=384, width=1280, with_labels=True, as_torch=True):
    """The ``data_dict`` the detector consumes (monocon_dataset.py:173-200)."""
    d = {
        "img": make_image(seed, batch, height, width),
        "img_metas": {
            "pad_shape": [(height, width)] * batch,
            "ori_shape": [(height, width)] * batch,
            "sample_idx": list(range(batch)),
        },
        "calib": [SynthCalib() for _ in range(batch)],
    }
    if with_labels:
        d["label"] = make_labels(seed, batch, height, width)
    if as_torch:
        import torch
        d["img"] = torch.from_numpy(d["img"])
        if with_labels:
            d["label"] = {k: torch.from_numpy(v) for k, v in d["label"].items()}
    return d


def make_decode_inputs(seed, batch, feat_h=96, feat_w=320, topk=100):
    """Config #5 inputs: post-sigmoid heatmap in [1e-4, 1-1e-4] plus regression maps.
    All heat values are distinct by construction (a random permutation-like ramp
    plus hash jitter), so top-(K+1) is tie-free; see tests for the explicit check."""
    hw = feat_h * feat_w
    heat = uniform(seed, "dec.heat", (batch, 3, feat_h, feat_w), 0.0, 1.0)
    # sparsify: most pixels low, a few hundred strong peaks per image
    h2 = heat * heat
    heat = h2 * h2 * h2            # exact IEEE products (no libm pow)
    heat = np.clip(heat * 0.9998 + 1e-4, 1e-4, 1 - 1e-4).astype(np.float32)
    d = {
        "center_heatmap_pred": heat,
        "wh_pred": (normalish(seed, "dec.wh", (batch, 2, feat_h, feat_w)) * 4 + 12).astype(np.float32),
        "offset_pred": uniform(seed, "dec.off", (batch, 2, feat_h, feat_w)).astype(np.float32),
        "center2kpt_offset_pred": normalish(seed, "dec.c2k", (batch, 18, feat_h, feat_w), 0, 3).astype(np.float32),
        "kpt_heatmap_offset_pred": normalish(seed, "dec.kho", (batch, 2, feat_h, feat_w)).astype(np.float32),
        "kpt_heatmap_pred": uniform(seed, "dec.kh", (batch, 9, feat_h, feat_w), 1e-4, 1 - 1e-4).astype(np.float32),
        "dim_pred": (np.abs(normalish(seed, "dec.dim", (batch, 3, feat_h, feat_w))) + 1.0).astype(np.float32),
        "depth_pred": np.concatenate([
            uniform(seed, "dec.z", (batch, 1, feat_h, feat_w), 2.0, 60.0),
            uniform(seed, "dec.s", (batch, 1, feat_h, feat_w), -1.0, 3.0)], 1).astype(np.float32),
        "alpha_cls_pred": normalish(seed, "dec.acls", (batch, 12, feat_h, feat_w)).astype(np.float32),
        "alpha_offset_pred": normalish(seed, "dec.aoff", (batch, 12, feat_h, feat_w), 0, 0.3).astype(np.float32),
    }
    return d


# --------------------------------------------------------------------------- conditioned train fixtures
# Gradient-parity fixtures (tests/golden/train_cond_*.npz).  make_state_dict()/make_batch() are built for
# test *power*; for gradient parity they are badly conditioned in two ways that have nothing to do with the
# kernels: (1) head output logits of O(1) drive the depth / uncertainty losses into e^{-s} * 1/sigmoid(x)
# regimes with gradient norms of 1e6..1e11, (2) i.i.d. noise images have identical per-image channel
# statistics, so the BatchNorm over the batch inside AttnBN normalises round-off.  The conditioned variant
# keeps every code path but uses small head output weights (as the reference initialiser does,
# monocon_heads.py:134-146) and gives every image its own contrast / brightness.
HEAD_OUT_SCALE = np.float32(0.02)


def is_head_output_weight(key):
    return key.startswith("head.") and (key.endswith(".3.weight") or key.endswith("dir_cls.0.weight")
                                        or key.endswith("dir_reg.0.weight"))


def make_conditioned_state_dict(seed=0, bn_stats=None, as_torch=True):
    sd = make_state_dict(seed, bn_stats=bn_stats, as_torch=False)
    for k in sd:
        if is_head_output_weight(k):
            sd[k] = (sd[k] * HEAD_OUT_SCALE).astype(np.float32)
    if as_torch:
        import torch
        return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
    return sd


def make_conditioned_batch(seed, batch, height, width, as_torch=True):
    d = make_batch(seed, batch, height, width, as_torch=False)
    t = (np.arange(batch, dtype=np.float64) / max(batch - 1, 1))
    sc = (0.4 + 1.2 * t).astype(np.float32).reshape(batch, 1, 1, 1)
    sh = ((t - 0.5).reshape(batch, 1) * np.array([1.0, -0.5, 0.3])).astype(np.float32).reshape(batch, 3, 1, 1)
    d["img"] = (d["img"] * sc + sh).astype(np.float32)          # two IEEE fp32 roundings, identical on every box
    if as_torch:
        import torch
        d["img"] = torch.from_numpy(d["img"])
        d["label"] = {k: torch.from_numpy(v) for k, v in d["label"].items()}
    return d


# ---- KITTI annotation dicts for the AP evaluator (tests/test_kitti_eval.py, tests/golden/make_f4_golden.py)
KITTI_NAMES = ("Car", "Pedestrian", "Cyclist", "Van", "Person_sitting", "DontCare", "Truck")


def random_kitti_annos(seed, frames=10):
    """seeded KITTI-format ground-truth / detection annotation dicts (the evaluator's input, reference
    engine/kitti_eval/eval.py:347-453): mixed classes incl. DontCare, jittered detections, false positives"""
    rng = np.random.default_rng(seed)
    gts, dts = [], []
    for f in range(frames):
        n = int(rng.integers(0, 9)) if f else 6
        names = rng.choice(KITTI_NAMES, n, p=[0.4, 0.2, 0.12, 0.08, 0.05, 0.1, 0.05])
        loc = np.stack([rng.uniform(-15, 15, n), rng.uniform(1.2, 2.0, n), rng.uniform(6, 55, n)], 1)
        dims = np.stack([rng.uniform(0.6, 4.5, n), rng.uniform(1.3, 2.0, n), rng.uniform(0.5, 2.0, n)], 1)      # l, h, w
        ry = rng.uniform(-math.pi, math.pi, n)
        x1, y1 = rng.uniform(0, 1000, n), rng.uniform(100, 250, n)
        bbox = np.stack([x1, y1, x1 + rng.uniform(20, 200, n), y1 + rng.uniform(15, 120, n)], 1)
        gt = {"name": names, "truncated": np.round(rng.uniform(0, 0.6, n), 2), "occluded": rng.integers(0, 4, n).astype(np.float64),
              "alpha": rng.uniform(-math.pi, math.pi, n), "bbox": bbox, "dimensions": dims, "location": loc, "rotation_y": ry,
              "score": np.zeros(n)}
        for i in range(n):
            if names[i] == "DontCare":
                gt["truncated"][i], gt["occluded"][i], gt["alpha"][i] = -1, -1, -10
                dims[i], loc[i], ry[i] = -1, -1000, -10
        gts.append(gt)
        real = [i for i in range(n) if names[i] != "DontCare"]
        keep = [i for i in real if rng.uniform() < 0.8]
        m_fp = int(rng.integers(0, 4))
        dn = [names[i] if rng.uniform() < 0.9 else "Car" for i in keep] + list(rng.choice(KITTI_NAMES[:3], m_fp))
        jit = lambda a, s: a + rng.normal(0, s, a.shape)       # noqa: E731
        dloc = np.concatenate([jit(loc[keep], 0.15), np.stack([rng.uniform(-15, 15, m_fp), rng.uniform(1.2, 2, m_fp), rng.uniform(6, 55, m_fp)], 1)])
        ddim = np.concatenate([np.abs(jit(dims[keep], 0.08)) + 0.05, np.stack([rng.uniform(0.6, 4.5, m_fp), rng.uniform(1.3, 2, m_fp), rng.uniform(0.5, 2, m_fp)], 1)])
        dry = np.concatenate([jit(ry[keep], 0.1), rng.uniform(-3, 3, m_fp)])
        fx1, fy1 = rng.uniform(0, 1000, m_fp), rng.uniform(100, 250, m_fp)
        dbox = np.concatenate([jit(bbox[keep], 3.0), np.stack([fx1, fy1, fx1 + rng.uniform(20, 200, m_fp), fy1 + rng.uniform(15, 120, m_fp)], 1)])
        k = len(keep) + m_fp
        dts.append({"name": np.array(dn, dtype=object).reshape(-1), "truncated": np.zeros(k), "occluded": np.zeros(k),
                    "alpha": np.concatenate([jit(gt["alpha"][keep], 0.2), rng.uniform(-3, 3, m_fp)]), "bbox": dbox.reshape(-1, 4),
                    "dimensions": ddim.reshape(-1, 3), "location": dloc.reshape(-1, 3), "rotation_y": dry,
                    "score": np.round(rng.uniform(0.05, 1.0, k), 3), "sample_idx": np.full(k, f)})
    return gts, dts
