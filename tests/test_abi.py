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


def test_workspace_queries_need_no_gpu():
    """bnpk_count_sparse_workspace / bnpk_index_build_workspace are host arithmetic: callable on a box without a GPU, monotone
    in the mode (plain levels <= claiming level <= any input) and in n, and sized as the header says (about n / 1.4 n / 6 n words)"""
    from bionumpy_amd import _native
    lib = _native.lib
    for n in (1, 1000, 1 << 20, 50_000_000, 6_000_000_000):
        sizes = [lib.bnpk_count_sparse_workspace(n, 62, 0, 0, 0, mode) for mode in (0, 1, 2)]
        assert 0 < sizes[0] <= sizes[1] <= sizes[2], (n, sizes)
        assert sizes[2] >= 6 * 8 * n
        assert lib.bnpk_count_sparse_workspace(2 * n, 62, 0, 0, 0, 2) >= sizes[2]
    # the headline's shape: 6e9 keys behind a 10-bit first level — the claiming level's slots are 2^20 x 7680 keys
    claim = lib.bnpk_count_sparse_workspace(6_000_000_000, 62, 0, 0, 10, 1)
    assert (1 << 20) * 7680 * 8 <= claim <= (1 << 20) * 7680 * 8 + 8 * 6_000_000_000
    assert lib.bnpk_count_sparse_workspace(0, 62, 0, 0, 0, 2) > 0
    assert lib.bnpk_index_build_workspace(12_000_000, 62, 17) >= lib.bnpk_count_sparse_workspace(12_000_000, 62, 0, 0, 0, 2)
