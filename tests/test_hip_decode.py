"""Decode parity: bit-exact keep mask / top-K indices / classes / threshold mask, floats to 1e-4.
GPU-only."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from hipmonocon import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from hipmonocon.engine import Engine
    return Engine()


def run(eng, d, K, thres=0.4, want_keep=True, pad_hw=(384, 1280), window=3):
    from hipmonocon.engine import p2_inverse
    B = d["center_heatmap_pred"].shape[0]
    pred = {k: torch.from_numpy(v).to(eng.device) for k, v in d.items()}
    P2 = np.stack([synth.KITTI_P2] * B)
    return eng.decode(pred, torch.from_numpy(P2).to(eng.device), torch.from_numpy(p2_inverse(P2)).to(eng.device),
                      pad_hw, K, thres, want_keep=want_keep, local_maximum_kernel=window)


@pytest.mark.parametrize("K", [30, 100])
def test_decode_vs_reference_golden(eng, K):
    g = load_golden("decode_k%d.npz" % K)
    d = synth.make_decode_inputs(int(g["seed"]), 4, 96, 320, topk=K)
    R = run(eng, d, K)
    HW = 96 * 320
    assert np.array_equal(np.packbits(R["keep"].cpu().numpy().astype(bool)), g["keep_packed"])
    assert np.array_equal((R["flat_index"] % HW).cpu().numpy(), g["ind"])
    assert np.array_equal(R["cls"].cpu().numpy(), g["cls"])
    assert np.array_equal(R["scores"].cpu().numpy(), g["scores"])
    for i in range(4):
        mk = R["box_mask"][i].cpu()
        assert int(mk.sum()) == g["box2d.%d" % i].shape[0]           # threshold mask identical
        assert rel_err(R["box2d"][i].cpu()[mk], g["box2d.%d" % i]) < 1e-4
        assert rel_err(R["box3d"][i].cpu()[mk], g["box3d.%d" % i]) < 1e-4
        assert np.array_equal(R["cls"][i].cpu()[mk].numpy(), g["label.%d" % i])


@pytest.mark.parametrize("window", [1, 5, 7, 3])
def test_decode_peak_filter_windows_vs_reference_golden(eng, window):
    """test_config['local_maximum_kernel'] other than 3: bit-exact against the reference's own filter + top-k
    (tests/golden/localmax_windows.npz) and, for the boxes, the oracle run with the same window; 3 last -- the handle's
    setting must come back"""
    from oracle import monocon_oracle as O
    g = load_golden("localmax_windows.npz")
    K, B, H, W = 20, 2, 24, 44
    d = synth.make_decode_inputs(int(g["seed"]), B, H, W, topk=K)
    R = run(eng, d, K, pad_hw=(4 * H, 4 * W), window=window)
    assert np.array_equal(np.packbits(R["keep"].cpu().numpy().astype(bool)), g["keep_packed.%d" % window])
    assert np.array_equal((R["flat_index"] % (H * W)).cpu().numpy(), g["ind.%d" % window])
    assert np.array_equal(R["cls"].cpu().numpy(), g["cls.%d" % window])
    assert np.array_equal(R["scores"].cpu().numpy(), g["scores.%d" % window])
    ref = O.decode({k: torch.from_numpy(v) for k, v in d.items()}, np.stack([synth.KITTI_P2] * B), (4 * H, 4 * W), topk=K,
                   thres=0.4, kernel=window)
    assert torch.equal(R["box_mask"].cpu(), ref["box_mask"])
    assert rel_err(R["box2d"].cpu(), ref["box2d"]) < 1e-4
    assert rel_err(R["box3d"].cpu(), ref["box3d_shift"]) < 1e-4


def test_decode_even_peak_filter_window_is_an_error(eng):
    from hipmonocon.lib import MonoconHipError
    d = synth.make_decode_inputs(5, 1, 8, 16, topk=4)
    with pytest.raises(MonoconHipError, match="odd"):
        run(eng, d, 4, pad_hw=(32, 64), window=4)
    R = run(eng, d, 4, pad_hw=(32, 64))        # the handle keeps its previous (valid) setting
    assert R["scores"].shape == (1, 4)


def test_decode_config5_b64_k100_vs_oracle(eng):
    """BASELINE config #5: B=64, K=100, thr 0.4 -- indices / masks bit-exact vs the CPU oracle."""
    from oracle import monocon_oracle as O
    d = synth.make_decode_inputs(4242, 64, 96, 320, topk=100)
    R = run(eng, d, 100)
    ref = O.decode({k: torch.from_numpy(v) for k, v in d.items()}, np.stack([synth.KITTI_P2] * 64), (384, 1280),
                   topk=100, thres=0.4)
    assert torch.equal(R["keep"].cpu().bool(), ref["keep"])
    assert torch.equal(R["flat_index"].cpu(), ref["flat_index"])
    assert torch.equal(R["cls"].cpu(), ref["cls"])
    assert torch.equal(R["scores"].cpu(), ref["scores"])
    assert torch.equal(R["box_mask"].cpu(), ref["box_mask"])
    assert rel_err(R["box2d"].cpu(), ref["box2d"]) < 1e-4
    assert rel_err(R["box3d"].cpu(), ref["box3d_shift"]) < 1e-4


def test_decode_ties_canonical_order(eng):
    """plateaus and clamped maxima: ties resolve by ascending flat index (oracle's canonical order)."""
    from oracle import monocon_oracle as O
    d = synth.make_decode_inputs(77, 2, 16, 32, topk=40)
    h = d["center_heatmap_pred"]
    h[:] = np.float32(1e-4)                   # everything tied at the clamp floor ...
    h[0, 1, 3:5, 7:9] = np.float32(1 - 1e-4)  # ... except a 2x2 plateau at the ceiling
    h[1, 2, 10, 20] = 0.5
    R = run(eng, d, 40, pad_hw=(64, 128))
    ref = O.decode({k: torch.from_numpy(v) for k, v in d.items()}, np.stack([synth.KITTI_P2] * 2), (64, 128),
                   topk=40, thres=0.4)
    assert torch.equal(R["keep"].cpu().bool(), ref["keep"])
    assert torch.equal(R["flat_index"].cpu(), ref["flat_index"])
    assert torch.equal(R["scores"].cpu(), ref["scores"])


def test_decode_fewer_positive_maxima_than_k(eng):
    """fewer than K positive local maxima: exact zeros reach the top-K (ascending flat index) -- the selection then
    runs over the whole filtered map instead of the compacted candidate list; a second call checks that the
    candidate counters were left clean."""
    from oracle import monocon_oracle as O
    d = synth.make_decode_inputs(78, 2, 16, 32, topk=12)
    h = d["center_heatmap_pred"]
    h[:] = 0.0
    h[0, 0, 5, 5] = 0.9; h[0, 2, 9, 30] = 0.7; h[0, 1, 0, 0] = 0.2
    h[1, 1, 15, 31] = 0.6
    for _ in range(2):
        R = run(eng, d, 12, pad_hw=(64, 128))
        ref = O.decode({k: torch.from_numpy(v) for k, v in d.items()}, np.stack([synth.KITTI_P2] * 2), (64, 128),
                       topk=12, thres=0.4)
        assert torch.equal(R["keep"].cpu().bool(), ref["keep"])
        assert torch.equal(R["flat_index"].cpu(), ref["flat_index"])
        assert torch.equal(R["scores"].cpu(), ref["scores"])
        assert torch.equal(R["cls"].cpu(), ref["cls"].to(R["cls"].dtype))


def test_decode_max_k_vs_oracle(eng):
    """K = 1024 (the C-ABI maximum, every LDS slot of the select kernel in use) on a 3x48x96 map."""
    from oracle import monocon_oracle as O
    d = synth.make_decode_inputs(31, 2, 48, 96, topk=1024)
    R = run(eng, d, 1024, pad_hw=(192, 384))
    ref = O.decode({k: torch.from_numpy(v) for k, v in d.items()}, np.stack([synth.KITTI_P2] * 2), (192, 384),
                   topk=1024, thres=0.4)
    assert torch.equal(R["flat_index"].cpu(), ref["flat_index"])
    assert torch.equal(R["scores"].cpu(), ref["scores"])
    assert torch.equal(R["box_mask"].cpu(), ref["box_mask"])


def test_decode_k_limits(eng):
    from hipmonocon.lib import MonoconHipError
    d = synth.make_decode_inputs(5, 1, 8, 8, topk=4)
    R = run(eng, d, 1)
    assert R["scores"].shape == (1, 1)
    with pytest.raises(MonoconHipError):
        run(eng, d, 2000)


def test_decode_full_size_properties(eng):
    """B=64, K=100 (BASELINE config 5) size-independent properties: scores sorted descending per image, flat
    indices unique, every kept candidate is a local maximum, classes derived from the flat index."""
    d = synth.make_decode_inputs(9001, 64, 96, 320, topk=100)
    R = run(eng, d, 100)
    sc, fi = R["scores"].cpu(), R["flat_index"].cpu()
    assert bool((sc[:, :-1] >= sc[:, 1:]).all())
    HW = 96 * 320
    for b in range(64):
        assert len(set(fi[b].tolist())) == 100
    assert torch.equal(R["cls"].cpu(), (fi // HW).to(R["cls"].dtype))
    heat = torch.from_numpy(d["center_heatmap_pred"]).reshape(64, -1)
    assert torch.equal(torch.gather(heat, 1, fi), sc)                       # scores are the heat-map values at the indices
    keep = R["keep"].cpu().bool().reshape(64, -1)
    assert bool(torch.gather(keep, 1, fi).all())                            # ... and every one of them is a 3x3 local maximum

