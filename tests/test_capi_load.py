"""The C-ABI library loads and exports every symbol include/surfh_amd.h declares.
No compute without a GPU: plan creation must fail loudly, never fall back to a CPU path."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.build()
    from surfh_amd import _lib
    return _lib.load()


def test_exports_match_header(lib):
    hdr = open(os.path.join(ROOT, "include", "surfh_amd.h")).read()
    declared = sorted(set(re.findall(r"\b(surfh_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 25
    from surfh_amd import _lib
    assert sorted(_lib.EXPORTS) == declared
    for name in declared:
        assert hasattr(lib, name), name
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "surfh_amd", "libsurfh_amd.so")]).decode()
    exported = set(re.findall(r" T (surfh_[a-z_0-9]+)", out))
    assert set(declared) <= exported


def test_gfx950_code_object_present():
    so = os.path.join(ROOT, "surfh_amd", "libsurfh_amd.so")
    data = open(so, "rb").read()
    assert b"gfx950" in data


def test_no_cpu_fallback_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import problems
    from helpers import build_model
    with pytest.raises((ValueError, RuntimeError)) as e:
        build_model(problems.config1())
    assert "device" in str(e.value).lower() or "hip" in str(e.value).lower()


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "surfh_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), fn
            assert "surfh_oracle" not in re.sub(r"oracle/surfh_oracle.py", "", src), fn
