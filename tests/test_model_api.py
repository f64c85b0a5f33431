"""Drop-in surface: the product modules keep the reference's names / state_dict / call signatures
 and route every computation to the HIP library."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err, GOLDEN_SEED
from hipmonocon import netspec, synth


def build(golden_sd=None, test_config=None):
    from model import MonoConDetector
    m = MonoConDetector(34, pretrained_backbone=False, test_config=test_config)
    if golden_sd is not None:
        m.load_state_dict(golden_sd, strict=True)
    return m


def test_state_dict_matches_reference_layout(golden_sd):
    m = build()
    sd = m.state_dict()
    spec = netspec.state_shapes()
    assert list(sd.keys()) == list(spec.keys())          # same 449 keys, same order
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(spec[k][0]), k
        assert (v.dtype == torch.int64) == (spec[k][1] == "i64"), k
    m.load_state_dict(golden_sd, strict=True)            # reference-format checkpoint loads strictly
    assert sum(p.numel() for p in m.parameters()) == 19620261
    assert len(list(m.parameters())) == 242


def test_public_names_import():
    from model import DLA, DLAUp, IDAUp, MonoConDenseHeads, AttnBatchNorm2d, MonoConDetector  # noqa: F401
    m = build()
    assert hasattr(m, "backbone") and hasattr(m, "neck") and hasattr(m, "head")
    for k in ("topk", "local_maximum_kernel", "max_per_img", "test_thres"):
        assert hasattr(m.head, k)
    m2 = build(test_config={"topk": 100, "local_maximum_kernel": 3, "max_per_img": 30, "test_thres": 0.4})
    assert m2.head.topk == 100


def test_init_distributions_follow_reference():
    m = build()
    w = m.backbone.level2.tree1.conv1.weight
    assert abs(float(w.std()) - (2.0 / (9 * 64)) ** 0.5) < 0.01
    assert float(m.head.heatmap_head[-1].bias[0]) == pytest.approx(-2.1972246, abs=1e-5)
    assert float(m.head.wh_head[0].weight.std()) < 2e-3
    up = m.neck.ida_0.up_1.weight
    assert torch.allclose(up[0, 0], torch.tensor([0.25, 0.75, 0.75, 0.25])[:, None] * torch.tensor([0.25, 0.75, 0.75, 0.25]))
    assert torch.equal(up[0], up[17])


def test_initialisers_pinned_to_the_reference():
    """all 449 state_dict entries of a freshly built detector against the reference built under
    the same torch seed (tests/golden/init_pins.npz): moments always, raw bytes (CRC-32) on the torch build that
    made the golden -- init_weights draws from the generator in the reference's order."""
    import zlib
    g = load_golden("init_pins.npz")
    torch.manual_seed(int(g["torch_seed"]))
    from model import MonoConDetector
    m = MonoConDetector(34, pretrained_backbone=False)
    sd = m.state_dict()
    assert list(sd.keys()) == g["names"].tolist()
    same_build = torch.__version__ == __import__("json").load(open(__import__("os").path.join(
        __import__("os").path.dirname(__file__), "golden", "meta.json")))["torch"]
    for i, (k, v) in enumerate(sd.items()):
        f = v.detach().double().reshape(-1)
        mom = np.array([float(f.mean()), float(f.std()) if f.numel() > 1 else 0.0, float(f.min()), float(f.max())])
        ref = g["moments"][i]
        if same_build:
            assert zlib.crc32(v.detach().contiguous().numpy().tobytes()) == int(g["crc32"][i]), k
            # identical bytes; the fp64 mean / std are reduced in an order that depends on the host's threads / vector width
            assert np.allclose(mom, ref, rtol=1e-9, atol=1e-12), k
        else:   # another torch build may draw different normals: distribution-level agreement
            n = max(f.numel(), 1)
            assert abs(mom[0] - ref[0]) <= 6 * max(ref[1], 1e-6) / n ** 0.5 + 1e-6, k
            assert abs(mom[1] - ref[1]) <= 0.1 * ref[1] + 6 * ref[1] / n ** 0.5 + 1e-6, k


def test_cpu_forward_is_a_loud_error(golden_sd):
    from hipmonocon.lib import MonoconHipError
    m = build(golden_sd).eval()
    with pytest.raises(MonoconHipError):
        m({"img": torch.zeros(1, 3, 64, 64)})
    with pytest.raises(RuntimeError):
        m.backbone.level0[0](torch.zeros(1, 16, 8, 8))      # holders never compute


@pytest.mark.gpu
def test_detector_eval_forward_and_submodules(golden_sd):
    from oracle import monocon_oracle as O
    m = build(golden_sd).cuda().eval()
    img = synth.make_batch(GOLDEN_SEED + 1, 2, 64, 128, with_labels=False)["img"]
    g = load_golden("fwd_small_eval.npz")
    pred = m({"img": img.cuda()})
    assert list(pred.keys()) == [k for k, _ in netspec.PRED_KEYS]
    for k, v in pred.items():
        assert rel_err(v.cpu(), g["f64." + k]) < 1e-4, k
    # sub-module API: DLA.forward -> DLAUp.forward -> head.forward_test chain equals the fused path
    with torch.no_grad():
        ref_pred, ref_feat, ref_levels, _ = O.forward(golden_sd, img, return_levels=True)
    levels = m.backbone(img.cuda())
    assert len(levels) == 6
    for a, b in zip(levels, ref_levels):
        assert rel_err(a.cpu(), b) < 1e-4
    feat = m.neck(levels)[0]
    assert rel_err(feat.cpu(), ref_feat) < 1e-4
    p2 = m.head.forward_test(feat)
    for k in pred:
        assert rel_err(p2[k].cpu(), pred[k].cpu()) < 1e-5, k
    assert rel_err(m._extract_feat_from_data_dict({"img": img.cuda()}).cpu(), ref_feat) < 1e-4


@pytest.mark.gpu
def test_batch_eval_vis_format_matches_oracle(golden_sd):
    from oracle import monocon_oracle as O
    m = build(golden_sd, test_config={"topk": 50, "local_maximum_kernel": 3, "max_per_img": 30,
                                      "test_thres": 0.05}).cuda().eval()
    batch = synth.make_batch(31, 2, 96, 160, with_labels=False)
    data = {"img": batch["img"].cuda(), "img_metas": batch["img_metas"], "calib": batch["calib"]}
    out = m.batch_eval(data, get_vis_format=True)
    pred = m(data)
    ref = O.decode({k: v.cpu() for k, v in pred.items()}, np.stack([c.P2 for c in batch["calib"]]), (96, 160),
                   topk=50, thres=0.05)
    assert len(out) == 2
    for i, r in enumerate(out):
        mk = ref["box_mask"][i]
        assert r["img_bbox"]["boxes_3d"].shape == (int(mk.sum()), 7)
        assert rel_err(r["img_bbox"]["boxes_3d"], ref["box3d_shift"][i][mk]) < 1e-4
        assert torch.equal(r["img_bbox"]["labels_3d"], ref["cls"][i][mk])
        assert rel_err(r["img_bbox"]["scores_3d"], ref["box2d"][i][mk][:, 4]) < 1e-4
        assert len(r["img_bbox2d"]) == 3
        assert sum(len(x) for x in r["img_bbox2d"]) == int(mk.sum())
    b2, b3, lab = m.head.decode_heatmap(data, pred)
    assert rel_err(b3[0], ref["box3d"][0][ref["box_mask"][0]]) < 1e-4      # un-shifted centres


@pytest.mark.gpu
def test_batch_eval_kitti_format(golden_sd):
    """default batch_eval output: KITTI annotation dicts (reference monocon_heads.py:363-376)."""
    m = build(golden_sd, test_config={"topk": 30, "local_maximum_kernel": 3, "max_per_img": 30,
                                      "test_thres": 0.05}).cuda().eval()
    batch = synth.make_batch(31, 2, 96, 160, with_labels=False)
    data = {"img": batch["img"].cuda(), "img_metas": batch["img_metas"], "calib": batch["calib"]}
    out = m.batch_eval(data)
    assert set(out) == {"img_bbox", "img_bbox2d"} and len(out["img_bbox"]) == len(out["img_bbox2d"]) == 2
    for a in out["img_bbox"] + out["img_bbox2d"]:
        n = len(a["score"])
        assert a["bbox"].shape == (n, 4) and a["location"].shape == (n, 3) and a["dimensions"].shape == (n, 3)
        assert len(a["name"]) == len(a["alpha"]) == len(a["rotation_y"]) == len(a["sample_idx"]) == n


@pytest.mark.gpu
def test_empty_targets_raise_like_the_reference(golden_sd):
    """reference losses/l1_loss.py:15 (README.MD:208-210): a batch without a single valid object asserts -- before
    anything is launched.  A validated mask tensor is not read back again (no per-step host sync on a resident
    batch), but an in-place edit or a new tensor is."""
    m = build(golden_sd).cuda().train()
    batch = synth.make_batch(11, 2, 64, 128)
    dd = {"img": batch["img"].cuda(), "label": {k: v.cuda() for k, v in batch["label"].items()},
          "img_metas": {"pad_shape": [(64, 128)] * 2}}
    _, losses = m(dd)
    assert all(torch.isfinite(v) for v in losses.values())
    _, losses2 = m(dd)                                   # same tensor object: validated once
    # (not bit-equal: the batch statistics are accumulated around the running mean, which the first pass moved)
    assert torch.allclose(torch.stack(list(losses.values())), torch.stack(list(losses2.values())), rtol=1e-3)
    dd["label"]["mask"].zero_()                          # in-place edit bumps the version: checked again
    with pytest.raises(AssertionError):
        m(dd)
    dd["label"]["mask"] = torch.zeros_like(dd["label"]["mask"])
    with pytest.raises(AssertionError):
        m(dd)
