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
