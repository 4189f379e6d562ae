"""The C-ABI library loads, exports every symbol include/sfm_amd.h declares, and fails loudly
without a GPU.  No compute call is made here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "sfm_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sfm_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from sfm_amd import capi
    assert declared_symbols() == sorted(capi.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    from sfm_amd import capi
    assert os.path.exists(capi.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(capi.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    capi.lib()  # argtypes bind


def test_version_string():
    from sfm_amd import capi
    assert b"gfx950" in capi.lib().sfm_version()


def test_no_gpu_means_an_error_not_a_fallback():
    import torch
    from sfm_amd import capi
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.SfmError) as e:
        capi.Context(0)
    assert e.value.code == capi.ERR_NO_DEVICE


def test_product_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/."""
    pkg = os.path.join(ROOT, "sfm_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower(), os.path.join(dirpath, f)


def test_missing_library_raises(monkeypatch):
    from sfm_amd import capi
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", "/nonexistent/libsfm_amd.so")
    with pytest.raises(ImportError):
        capi.lib()
