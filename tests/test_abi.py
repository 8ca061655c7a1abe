"""The C-ABI library loads without a GPU and exports every symbol include/ncg.h declares;
compute entry points fail loudly (NativeError), never silently fall back."""
import ctypes
import os
import re

import pytest

import noble_curves_amd
from noble_curves_amd._native import NativeError, load_library

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ncg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ncg_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    lib = load_library()
    syms = declared_symbols()
    assert "ncg_msm" in syms and "ncg_mul_var_batch" in syms and len(syms) >= 10
    for s in syms:
        assert hasattr(lib, s), "libncg.so does not export %s" % s


def test_metadata_calls_without_gpu():
    lib = load_library()
    assert lib.ncg_version().startswith(b"noble-curves-amd")
    assert [lib.ncg_point_bytes(c) for c in range(5)] == [64, 64, 96, 192, 0]
    assert lib.ncg_field_bytes(2) == 48


def test_no_silent_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(NativeError, match="noble-gpu"):
        noble_curves_amd.Engine(0)
    h = ctypes.c_void_p()
    assert load_library().ncg_init(0, ctypes.byref(h)) != 0 and not h.value


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "noble-curves_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")) and f != "hosttest.hip":
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "liboracle" not in src, f
