"""Loading a checkpoint written by the REFERENCE engine.

tests/golden/ref_checkpoint.npz (make_golden.py ref_checkpoint) holds the reference's own checkpoint dict
(engine/base_engine.py:155-189: real MonoConDetector / torch.optim.AdamW / CyclicScheduler state after two steps) with its
235 MB of tensors replaced by placeholders + shape / dtype / checksum / samples.  Here every tensor is rebuilt from the
seeds (the parameters ARE the seed-7 synthetic state dict: the reference's two optimizer steps ran at lr 0), checked
against the recorded samples, put back into the recorded structure, torch.save'd -- together with a pickled dataset object
of the reference's class path and one of a class path this repository does not have, as the reference's missing comma at
base_engine.py:171 lets `test_dataset` into the file -- and loaded through this repository's loaders."""
import json
import os
import sys
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN_SEED, load_golden, rel_err
from hipmonocon import netspec, synth


def _rebuild(golden_sd):
    g = load_golden("ref_checkpoint.npz")
    skeleton = json.loads(bytes(g["skeleton"]).decode())
    names = [n for n, _, _, role in netspec.state_fields() if role == "param"]
    tensors = {}

    def build(o, path):
        if isinstance(o, dict) and "__t__" in o:
            return ("T", o["__t__"], path)
        if isinstance(o, dict) and "__dict__" in o:
            return {(k["__int__"] if isinstance(k, dict) else k): build(v, path + ((k["__int__"] if isinstance(k, dict) else k),))
                    for k, v in o["__dict__"]}
        if isinstance(o, dict) and "__list__" in o:
            return [build(v, path + (i,)) for i, v in enumerate(o["__list__"])]
        if isinstance(o, dict) and "__tuple__" in o:
            return tuple(build(v, path + (i,)) for i, v in enumerate(o["__tuple__"]))
        return o

    tree = build(skeleton, ())
    b1, b2 = [float(x) for x in g["betas1"]], 0.99

    def moments(pidx):
        n = names[pidx]
        shape = tuple(golden_sd[n].shape)
        gs = [torch.from_numpy(synth.normalish(9000 + s, n, shape, 0.0, 1e-3).astype(np.float32)) for s in range(2)]
        m = torch.zeros(shape); v = torch.zeros(shape)
        for s in range(2):
            m.mul_(b1[s]).add_(gs[s], alpha=1.0 - b1[s])
            v.mul_(b2).addcmul_(gs[s], gs[s], value=1.0 - b2)
        return m, v

    def fill(o):
        if isinstance(o, tuple) and len(o) == 3 and o[0] == "T":
            _, i, path = o
            if path[:2] == ("state_dict", "model"):
                t = golden_sd[path[2]].clone().reshape(tuple(int(x) for x in g["t%d.shape" % i]))     # (0-dim num_batches_tracked)
                tol = 0.0
            else:
                assert path[:3] == ("state_dict", "optimizer", "state"), path
                pidx, field = path[3], path[4]
                if field == "step":
                    t, tol = torch.tensor(2.0), 0.0
                else:
                    m, v = moments(pidx)
                    t, tol = (m if field == "exp_avg" else v), 1e-5
            assert tuple(t.shape) == tuple(int(x) for x in g["t%d.shape" % i]), (path, t.shape)
            assert str(t.dtype) == str(g["t%d.dtype" % i]), (path, t.dtype, str(g["t%d.dtype" % i]))
            flat = t.reshape(-1)
            smp = flat[::max(1, flat.numel() // 16)][:16].double().numpy() if flat.numel() else np.zeros(0)
            ref = g["t%d.samples" % i]
            assert np.allclose(smp, ref, rtol=tol, atol=tol * float(np.abs(ref).max() if ref.size else 0.0)), path
            assert abs(float(t.double().sum()) - float(g["t%d.sum" % i])) <= max(tol, 1e-12) * max(1.0, float(t.double().abs().sum())), path
            tensors[i] = t
            return t
        if isinstance(o, dict):
            return {k: fill(v) for k, v in o.items()}
        if isinstance(o, list):
            return [fill(v) for v in o]
        if isinstance(o, tuple):
            return tuple(fill(v) for v in o)
        return o

    ck = fill(tree)
    assert len(tensors) == int(g["n_tensors"])
    return ck, g


def _save_like_the_reference(ck, path):
    """torch.save with the two extra objects a real reference file carries in engine_attrs"""
    from dataset.monocon_dataset import MonoConDataset
    ds = object.__new__(MonoConDataset)                          # the reference pickles its test dataset (same class path)
    ds.__dict__.update({"base_root": "/data/kitti", "split": "val", "max_objs": 30, "pad_divisor": 32})
    mod = types.ModuleType("transforms.geo_aware_transforms")    # a class path only the reference has
    cls = type("RandomCrop3D", (), {"__module__": "transforms.geo_aware_transforms"})
    mod.RandomCrop3D = cls
    sys.modules["transforms.geo_aware_transforms"] = mod
    try:
        alien = cls()
        alien.prob, alien.crop_size = 0.5, (320, 960)
        ck = dict(ck)
        ck["engine_attrs"] = dict(ck["engine_attrs"], test_dataset=ds, leftover_transform=alien)
        torch.save(ck, path)
    finally:
        del sys.modules["transforms.geo_aware_transforms"]


def test_reference_checkpoint_layout_and_host_side_load(golden_sd, tmp_path):
    """CPU: the file loads (tolerant unpickler), the detector adopts the 449 tensors, the fused AdamW adopts the moments /
    step counts / param_groups the reference's torch.optim.AdamW wrote, the scheduler its counters, the engine its
    attributes -- through MonoConDetector.load_checkpoint and BaseEngine.load_checkpoint."""
    from engine.base_engine import BaseEngine
    from model import MonoConDetector
    from solver import AdamW, CyclicScheduler
    from utils.engine_utils import OpaqueReferenceObject, load_checkpoint_file
    ck, g = _rebuild(golden_sd)
    path = str(tmp_path / "epoch_002.pth")
    _save_like_the_reference(ck, path)
    raw = load_checkpoint_file(path)
    assert list(raw) == ["engine_attrs", "state_dict"] and list(raw["state_dict"]) == ["model", "optimizer", "scheduler"]
    assert isinstance(raw["engine_attrs"]["leftover_transform"], OpaqueReferenceObject)
    assert raw["engine_attrs"]["leftover_transform"].crop_size == (320, 960)
    assert type(raw["engine_attrs"]["test_dataset"]).__name__ == "MonoConDataset"
    assert list(raw["state_dict"]["model"]) == [k for k, _, _, _ in netspec.state_fields()]      # 449 keys, reference order
    og = raw["state_dict"]["optimizer"]["param_groups"][0]
    assert og["betas"][1] == 0.99 and og["weight_decay"] == 1e-5 and og["lr"] > 2.25e-4
    assert "initial_lr" in og and "initial_momentum" in og
    assert len(raw["state_dict"]["optimizer"]["state"]) == 236                                     # live parameters only

    m = MonoConDetector(34, pretrained_backbone=False)
    m.load_checkpoint(path)
    for k, v in m.state_dict().items():
        assert torch.equal(v, golden_sd[k].to(v.dtype).reshape(v.shape)), k

    class Stub:                                   # BaseEngine.load_checkpoint only needs these attributes
        model = MonoConDetector(34, pretrained_backbone=False)
        world, rank, local_rank = 1, 0, 0
        epochs, global_iters = 1, 1
        _say = staticmethod(lambda *a, **k: None)
    st = Stub()
    st.optimizer = AdamW(st.model.parameters(), lr=2.25e-4, weight_decay=1e-5, betas=(0.95, 0.99), max_grad_norm=35.0)
    st.scheduler = CyclicScheduler(st.optimizer, total_steps=1000)
    BaseEngine.load_checkpoint(st, path)
    assert st.epochs == 3 and st.global_iters == 931 and st.epoch_times == [101.5, 99.25] and st.weight_dir == "./exps/ref/checkpoints"
    # the pickled dataset the reference's missing comma let into `engine_attrs`, and the stand-in for a class only the
    # reference has, are NOT adopted as engine attributes
    assert not hasattr(st, "test_dataset") and not hasattr(st, "leftover_transform")
    assert torch.equal(st.model.state_dict()["backbone.level2.tree1.conv1.weight"], golden_sd["backbone.level2.tree1.conv1.weight"])
    osd = st.optimizer.state_dict()
    assert osd["param_groups"][0]["lr"] == og["lr"] and tuple(osd["param_groups"][0]["betas"]) == tuple(og["betas"])
    params = list(st.model.parameters())
    for idx, ref_state in raw["state_dict"]["optimizer"]["state"].items():
        mine = st.optimizer.state[params[idx]]
        assert float(mine["step"]) == 2.0
        assert torch.equal(mine["exp_avg"], ref_state["exp_avg"]) and torch.equal(mine["exp_avg_sq"], ref_state["exp_avg_sq"])
    assert st.scheduler.state_dict()["_step_count"] == raw["state_dict"]["scheduler"]["_step_count"] == 3
    assert st.scheduler.last_epoch == raw["state_dict"]["scheduler"]["last_epoch"] == 2


@pytest.mark.gpu
def test_reference_checkpoint_eval_forward_and_resumed_step(golden_sd, tmp_path):
    """GPU: the model loaded from the reference-layout file reproduces fwd_small_eval.npz (the reference's own forward of
    these parameters), and the optimizer resumed from the reference's moments takes its THIRD step exactly as the oracle's
    restatement of torch.optim.AdamW does (step count 3: bias corrections 1 - beta^3)."""
    from model import MonoConDetector
    from oracle import monocon_oracle as O
    from solver import AdamW, CyclicScheduler
    ck, g = _rebuild(golden_sd)
    path = str(tmp_path / "epoch_002.pth")
    _save_like_the_reference(ck, path)
    m = MonoConDetector(34, pretrained_backbone=False)
    m.load_checkpoint(path)
    m = m.cuda().eval()
    gf = load_golden("fwd_small_eval.npz")
    img = synth.make_batch(GOLDEN_SEED + 1, 2, 64, 128, with_labels=False)["img"].cuda()
    preds = m({"img": img}, return_loss=False)
    for k, v in preds.items():
        assert rel_err(v.cpu(), gf["f64." + k]) < 1e-4, k
    # resumed optimizer step
    from utils.engine_utils import load_checkpoint_file
    raw = load_checkpoint_file(path)
    opt = AdamW(m.parameters(), lr=2.25e-4, weight_decay=1e-5, betas=(0.95, 0.99), max_grad_norm=35.0)
    sch = CyclicScheduler(opt, total_steps=1000)
    opt.load_state_dict(raw["state_dict"]["optimizer"])
    sch.load_state_dict(raw["state_dict"]["scheduler"])
    names = [n for n, _ in m.named_parameters()]
    live = [(n, p) for n, p in m.named_parameters() if n not in netspec.DEAD_PARAMS]
    grads = {n: torch.from_numpy(synth.normalish(9002, n, tuple(p.shape), 0.0, 1e-3).astype(np.float32)) for n, p in live}
    for n, p in live:
        p.grad = grads[n].cuda()
    lr, (b1, b2) = opt.param_groups[0]["lr"], opt.param_groups[0]["betas"]
    ps = [golden_sd[n].clone() for n, _ in live]
    gs = [grads[n].clone() for n, _ in live]
    st = raw["state_dict"]["optimizer"]["state"]
    ms = [st[names.index(n)]["exp_avg"].clone() for n, _ in live]
    vs = [st[names.index(n)]["exp_avg_sq"].clone() for n, _ in live]
    with torch.no_grad():
        O.clip_and_adamw(ps, gs, ms, vs, 3, lr, b1, beta2=b2)
    opt.step()
    torch.cuda.synchronize()
    worst = max(rel_err(p.detach().cpu(), ref) for (n, p), ref in zip(live, ps))
    assert worst < 2e-6, worst
