"""The C-ABI library loads and exports every symbol include/bnpk.h declares (no compute, no GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "bnpk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bnpk_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from bionumpy_amd import _native
    declared = _declared_symbols()
    assert len(declared) >= 40
    for name in declared:
        assert hasattr(_native.lib, name), "libbnpk.so does not export %s" % name
    assert sorted(_native.SIGNATURES) == declared, "ctypes table and include/bnpk.h disagree"


def test_status_strings_and_version():
    from bionumpy_amd import _native
    assert _native.lib.bnpk_version() >= 100
    assert _native.lib.bnpk_strerror(0) == b"ok"
    for code in range(-6, 0):
        assert _native.lib.bnpk_strerror(code) not in (b"ok", b"unknown bnpk status")
    assert _native.lib.bnpk_synth_record_bytes(150) == 316
    assert _native.lib.bnpk_scan_tiles(0) == 0 and _native.lib.bnpk_scan_tiles(16385) == 2


def test_no_cpu_fallback_without_gpu():
    """on a box without a GPU the product ops must raise, not fall back"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from bionumpy_amd import ops, _native
    ops.set_ops(None)
    with pytest.raises(_native.BnpkError):
        ops.get_ops()


def test_product_never_imports_oracle():
    """bionumpy_amd must not reference the oracle package (it is test infrastructure)"""
    pkg = os.path.join(ROOT, "bionumpy_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(base, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(base, f)
