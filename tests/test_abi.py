"""CPU-side checks of the drop-in boundary: the shared library loads and exports exactly the
entry points include/monocon_hip.h declares (no compute calls without a GPU)."""
import os
import re

import pytest

from conftest import REPO


def header_symbols():
    txt = open(os.path.join(REPO, "include", "monocon_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mc_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree():
    from hipmonocon import lib
    assert sorted(lib.EXPORTS) == header_symbols()


def test_library_exports_every_declared_symbol():
    from hipmonocon import lib
    if not os.path.exists(lib.LIB_PATH):
        pytest.skip("libmonocon_hip.so not built (run __graft_entry__.build())")
    l = lib.load()
    for name in header_symbols():
        assert hasattr(l, name), name
    assert l.mc_version() == 1


def test_no_gpu_is_a_loud_error():
    import torch
    from hipmonocon import lib
    if torch.cuda.is_available() or not os.path.exists(lib.LIB_PATH):
        pytest.skip("needs a CPU-only box with the library built")
    from hipmonocon.engine import Engine
    with pytest.raises(lib.MonoconHipError):
        Engine()


def test_product_never_imports_oracle():
    pkg = os.path.join(REPO, "monocon-pytorch_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), os.path.join(root, f)


def test_bench_cli_contract_without_a_gpu():
    """`python bench.py --gpus N --steps K --warmup W` is the driver's contract: the flags exist, default to one GPU and a
    run of minutes, and the headline precision mode is one of the three fp32 parity modes."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    for flag in ("--gpus", "--steps", "--warmup", "--batch", "--precision", "--no-cpu-baseline"):
        assert flag in r.stdout, flag
    src = open(os.path.join(REPO, "bench.py")).read()
    assert 'choices=("f16x2", "bf16x3", "fp32")' in src and '"--gpus", type=int, default=1' in src


def test_bench_compact_line_carries_every_baseline_config():
    """the ONE stdout line must stay small enough to survive a log tail and still hold the numbers of every
    BASELINE config (train step = configs[2] shape, forward_only = configs[1], decode_only = configs[4], the per-mode
    legs, the roofline and cpu_baseline objects).  Exercised on a committed verbose report of an earlier round."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    full = json.load(open(os.path.join(REPO, "profiles", "r3h_bench.json")))
    line = bench.compact_line(full, os.path.join(REPO, "gpurun_out", "bench_full.json"))
    text = json.dumps(line)
    assert len(text) < 4096, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "forward_only", "decode_only", "modes"):
        assert k in line, k
    assert line["value"] == full["value"] and line["roofline"]["frac"] == full["roofline"]["frac"]
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(line["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert line["forward_only"]["img_s"] == full["forward_only"]["images_per_sec"]
    assert set(line["modes"]) == {"fp32", "bf16x3", "bf16"} and line["modes"]["fp32"]["train_ms"] == full["native_fp32"]["train"]["ms_per_step"]
    assert line["decode_only"]["img_s"] == full["decode_only"]["images_per_sec"]
    assert all(not isinstance(v, str) or len(v) < 120 for v in line["roofline"].values())


def test_committed_bench_reports_are_headline_runs():
    """every bench report kept under profiles/ as evidence of a headline number (`*_bench_full.json`, `*_bench_line.json`,
    `*_bench.json`) is a one-GPU run of the headline workload with its roofline and CPU baseline -- round 4 committed a
    2-rank smoke run under such a name (a test had overwritten the default report path before it was copied)."""
    import glob
    import json
    prof = os.path.join(REPO, "profiles")
    files = sorted(glob.glob(os.path.join(prof, "*_bench_full.json")) + glob.glob(os.path.join(prof, "*_bench_line.json")) +
                   glob.glob(os.path.join(prof, "r[3-9]*_bench.json")))
    assert files, "no bench reports under profiles/"
    for f in files:
        d = json.load(open(f))
        assert d["n_gpus"] == 1 and d.get("world_size", 1) == 1, f
        assert d["config"]["global_batch"] == 32 and d["config"]["parallelism"] == "dp1", f
        assert "roofline" in d and d["roofline"].get("frac"), f
        assert "cpu_baseline" in d and d["cpu_baseline"].get("value"), f
        assert d["steps"] >= 5 and 300 < d["value"] < 2000, f
        # a report measured on the fallback exchange (torch.distributed's all_reduce instead of the handle's
        # overlapped RCCL buckets) is not evidence of the data-parallel path
        assert not d.get("comm_fallback") and not str(d.get("comm_path", "")).startswith("torch (FALLBACK"), f
