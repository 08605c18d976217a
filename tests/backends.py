"""Backend fixture shared by the API tests.

Every API test runs twice:
  * ``hostlogic`` (CPU, ``-m "not gpu"``): the Python host logic of bionumpy_amd driven by the
    oracle-backed stand-in ops of tests/oracle_ops.py — checks the chunk loop, lazy chunk objects,
    array classes, exception mapping and the API surface;
  * ``hip`` (``-m gpu``): the same test through the real HIP kernels via the C-ABI — the parity test
    proper (expected values come from the reference goldens or from the oracle).
"""
import pytest

BACKENDS = [pytest.param("oracle", id="hostlogic"),
            pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def bnp(request):
    import bionumpy_amd
    from bionumpy_amd import ops as ops_mod
    if request.param == "oracle":
        from oracle_ops import OracleOps
        ops_mod.set_ops(OracleOps())
    else:
        ops_mod.set_ops(None)
        ops_mod.get_ops()            # raises loudly without a GPU / libbnpk.so
    yield bionumpy_amd
    ops_mod.set_ops(None)
