"""Engine-level drop-in surface: config shim, checkpoint layout, and -- on the GPU --
the reference's train loop end to end on the synthetic KITTI-shaped dataset."""
import os

import pytest
import torch


def small_cfg(tmp_path, epochs=2):
    from utils.engine_utils import get_default_cfg
    cfg = get_default_cfg()
    cfg.set_new_allowed(True)
    cfg.OUTPUT_DIR = str(tmp_path)
    cfg.SEED = 3
    cfg.DATA.ROOT = 'synthetic'
    cfg.DATA.SYNTHETIC_LENGTH = 8
    cfg.DATA.SYNTHETIC_HW = [192, 384]
    cfg.DATA.BATCH_SIZE = 4
    cfg.DATA.NUM_WORKERS = 0
    cfg.MODEL.BACKBONE.IMAGENET_PRETRAINED = False
    cfg.SOLVER.OPTIM.NUM_EPOCHS = epochs
    cfg.PERIOD.EVAL_PERIOD = 1
    cfg.PERIOD.LOG_PERIOD = 1
    return cfg


def test_cfg_shim_roundtrip(tmp_path):
    from utils.engine_utils import export_cfg, get_default_cfg, load_cfg
    c = get_default_cfg()
    assert c.SOLVER.OPTIM.LR == 2.25e-4 and c.SOLVER.CLIP_GRAD.MAX_NORM == 35 and c.MODEL.HEAD.MAX_OBJS == 30
    assert c.get('USE_BENCHMARK', False) is True
    c.DATA.BATCH_SIZE = 2
    p = os.path.join(tmp_path, "cfg.yaml")
    export_cfg(c, p)
    c2 = load_cfg(p)
    assert c2.DATA.BATCH_SIZE == 2 and c2.DATA.FILTER.MAX_DEPTH == 65
    with pytest.raises(KeyError):
        get_default_cfg().merge_from_file(_write(tmp_path, "NOT_A_KEY: 1\n"))


def _write(tmp_path, text):
    p = os.path.join(tmp_path, "x.yaml")
    open(p, "w").write(text)
    return p


def test_synthetic_dataset_collate_contract():
    from dataset.synthetic_dataset import SyntheticMonoConDataset
    ds = SyntheticMonoConDataset(length=3, height=64, width=96)
    b = ds.collate_fn([ds[0], ds[1]])
    assert b['img'].shape == (2, 3, 64, 96) and b['img'].dtype == torch.float32
    assert b['label']['gt_bboxes'].shape == (2, 30, 4) and b['label']['gt_kpts_2d'].shape == (2, 30, 18)
    assert all(v.dtype == torch.float32 for v in b['label'].values())
    assert b['img_metas']['pad_shape'] == [(64, 96), (64, 96)] and b['calib'][0].P2.shape == (3, 4)


@pytest.mark.gpu
def test_engine_trains_checkpoints_and_resumes(tmp_path):
    from engine.monocon_engine import MonoconEngine
    cfg = small_cfg(tmp_path)
    eng = MonoconEngine(cfg)
    assert eng.optimizer.__class__.__name__ == 'AdamW' and eng.scheduler.total_steps == 4
    eng.train()
    ck = sorted(os.listdir(os.path.join(tmp_path, 'checkpoints')))
    assert 'epoch_001.pth' in ck and 'epoch_002_final.pth' in ck
    assert os.path.isfile(os.path.join(tmp_path, 'config.yaml'))
    assert len(eng.entire_losses) == 4 and all(l == l for l in eng.entire_losses)
    d = torch.load(os.path.join(tmp_path, 'checkpoints', 'epoch_002_final.pth'), weights_only=False)
    assert set(d) == {'engine_attrs', 'state_dict'} and len(d['state_dict']['model']) == 449
    assert d['engine_attrs']['epochs'] == 3 and d['engine_attrs']['global_iters'] == 5
    # auto-resume picks the lexicographically last checkpoint (reference base_engine.py:63-71)
    eng2 = MonoconEngine(small_cfg(tmp_path, epochs=3))
    assert eng2.epochs == 3 and eng2.global_iters == 5
    a, b = eng.model.state_dict(), eng2.model.state_dict()
    assert all(torch.equal(a[k].cpu(), b[k].cpu()) for k in a)
    # the optimizer comes back with torch.optim.AdamW's per-parameter state (step restored, warm moments) and the
    # one-cycle schedule at the step it was saved at (reference base_engine.py:155-219)
    st1, st2 = eng.optimizer.state_dict()['state'], eng2.optimizer.state_dict()['state']
    assert st1.keys() == st2.keys() and len(st2) == 236
    for k in st1:
        assert float(st2[k]['step']) == float(st1[k]['step']) == 4.0
        assert torch.equal(st1[k]['exp_avg'].cpu(), st2[k]['exp_avg'].cpu())
        assert torch.equal(st1[k]['exp_avg_sq'].cpu(), st2[k]['exp_avg_sq'].cpu())
    assert eng2.scheduler._step_count == eng.scheduler._step_count
    assert eng2.optimizer.param_groups[0]['lr'] == eng.optimizer.param_groups[0]['lr']
    ev = eng2.evaluate()
    assert ev['num_results'] == 2.0


@pytest.mark.gpu
def test_engine_loop_reports_the_losses_of_the_reference_loop(tmp_path, monkeypatch):
    """the default loop uploads a batch ahead on a copy stream, checks the labels on the host and reads each step's loss one
    step late (hipmonocon/feed.py); MONOCON_HIP_SYNC_LOOP=1 is the reference's loop as written (monocon_engine.py:84-102:
    move_data_device + total_loss.item() inside the step).  Same shuffled batches, same steps: the same loss per step, in the
    same order, and the same weights at the end.  LOG_PERIOD=3 leaves steps whose loss is only read afterwards."""
    from engine.monocon_engine import MonoconEngine
    from utils.engine_utils import set_random_seed
    res = []
    for tag, sync in (("a", "0"), ("b", "1")):
        monkeypatch.setenv("MONOCON_HIP_SYNC_LOOP", sync)
        set_random_seed(3)                  # as train.py does before it builds the engine: same weights, same shuffles
        cfg = small_cfg(os.path.join(tmp_path, tag))
        cfg.PERIOD.LOG_PERIOD = 3
        os.makedirs(cfg.OUTPUT_DIR, exist_ok=True)
        eng = MonoconEngine(cfg)
        eng.train()
        res.append((list(eng.entire_losses), {k: v.detach().cpu().clone() for k, v in eng.model.state_dict().items()}))
    assert len(res[0][0]) == 4 and res[0][0] == res[1][0]
    assert all(torch.equal(res[0][1][k], res[1][1][k]) for k in res[0][1])


@pytest.mark.gpu
def test_train_script_with_worker_processes(tmp_path):
    """`python train.py ... DATA.NUM_WORKERS 2` (reference train.py:19-45): with workers and a device the engine feeds itself
    through hipmonocon.feed.RingLoader -- fork-server workers (the script is imported again by each of them: its body sits
    behind the __main__ guard), frames through the page-locked ring, two epochs over the same persistent workers"""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(repo, "monocon-pytorch_amd", "train.py"),
           "OUTPUT_DIR", str(tmp_path), "SEED", "5", "DATA.ROOT", "synthetic", "DATA.SYNTHETIC_LENGTH", "8",
           "DATA.SYNTHETIC_HW", "[96, 224]", "DATA.BATCH_SIZE", "2", "DATA.NUM_WORKERS", "2",
           "MODEL.BACKBONE.IMAGENET_PRETRAINED", "False", "SOLVER.OPTIM.NUM_EPOCHS", "2", "PERIOD.EVAL_PERIOD", "1",
           "PERIOD.LOG_PERIOD", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("- Average Loss:") == 2 and "Using Random Seed 5" in r.stdout
    assert "RingLoader unavailable" not in r.stdout
    assert os.path.isfile(os.path.join(tmp_path, "checkpoints", "epoch_002_final.pth"))


@pytest.mark.gpu
@pytest.mark.parametrize("device_augment,workers", ((True, 1), (False, 1), (True, 0)))
def test_engine_on_the_kitti_file_dataset(tmp_path, device_augment, workers):
    """MonoconEngine on a KITTI tree (tests/golden/kitti_mini: two real-size frames, 'train' and 'val' splits) with and without a
    loader worker: the train split's random augmentations, frames through RingLoader (uint8 frames + mc_preprocess_augmented with
    DATA.DEVICE_AUGMENT, the default; float32 frames from the host transforms without), one epoch, the AP evaluation of the
    'val' split (reference engine/monocon_engine.py:84-143, dataset/monocon_dataset.py:22-42)"""
    from engine.monocon_engine import MonoconEngine
    from hipmonocon.feed import RingLoader
    cfg = small_cfg(tmp_path, epochs=1)
    cfg.DATA.ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kitti_mini")
    cfg.DATA.BATCH_SIZE = 2
    cfg.DATA.NUM_WORKERS = workers
    cfg.DATA.DEVICE_AUGMENT = device_augment
    eng = MonoconEngine(cfg)
    if workers:
        assert isinstance(eng.train_loader, RingLoader) and isinstance(eng.test_loader, RingLoader)
        want = (torch.uint8, (2, 384, 1248, 3)) if device_augment else (torch.float32, (2, 3, 384, 1248))
        assert (eng.train_loader.ring.dtype, tuple(eng.train_loader.ring.shape[1:])) == want
    else:                                    # no workers: torch's DataLoader in this process, the same deferred samples
        assert not isinstance(eng.train_loader, RingLoader)
        assert next(iter(eng.train_loader))["img"].dtype == torch.uint8
    eng.train()
    assert len(eng.entire_losses) == 1 and eng.entire_losses[0] == eng.entire_losses[0]
    ap = eng.evaluate()
    assert len(ap) == 4 * 21 and all(0.0 <= v <= 100.0 for v in ap.values())


def test_kitti_conversion_matches_reference_golden():
    """utils/kitti_convert_utils.py (host-side) against the reference's annos for the same decode."""
    import numpy as np
    from conftest import load_golden
    from hipmonocon import synth
    from utils.kitti_convert_utils import CLASSES, convert_to_kitti_2d, convert_to_kitti_3d
    g = load_golden("decode_k30.npz")
    metas = {"ori_shape": [(375, 1242)] * 4, "sample_idx": [11, 12, 13, 14]}
    res3d, res2d = [], []
    for i in range(4):
        b2, b3, lab = g["box2d.%d" % i], g["box3d.%d" % i], g["label.%d" % i]
        res3d.append({"boxes_3d": torch.from_numpy(b3), "scores_3d": torch.from_numpy(b2[:, 4]), "labels_3d": torch.from_numpy(lab)})
        res2d.append([b2[lab == c] for c in range(3)])
    k3 = convert_to_kitti_3d(res3d, metas, [synth.SynthCalib() for _ in range(4)])
    k2 = convert_to_kitti_2d(res2d, metas)
    for field, ours in (("img_bbox", k3), ("img_bbox2d", k2)):
        for i in range(4):
            a = ours[i]
            names = g["kitti.%s.%d.name" % (field, i)]
            assert [CLASSES.index(n) for n in a["name"]] == names.tolist()
            for kk in ("alpha", "bbox", "dimensions", "location", "rotation_y", "score", "sample_idx"):
                ref = g["kitti.%s.%d.%s" % (field, i, kk)]
                got = np.asarray(a[kk], dtype=np.float64)
                assert got.shape == ref.shape, (field, i, kk, got.shape, ref.shape)
                assert np.allclose(got, ref, rtol=1e-5, atol=1e-4), (field, i, kk)
