"""DeviceVector (bionumpy_amd/device_vector.py) against numpy: whatever a caller does with the per-row values of a ragged
array — compare, combine masks, assign slices, reduce, index, fall back to arbitrary numpy — gives what numpy gives on
the plain array.  Runs on the host-logic backend (CPU) and through the kernels (-m gpu)."""
import os
import sys

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st, HealthCheck

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from backends import bnp  # noqa: E402,F401


def _vec(values, dtype):
    from bionumpy_amd.device import HArray
    from bionumpy_amd.device_vector import DeviceVector
    a = np.asarray(values, dtype=dtype)
    return DeviceVector(HArray(host=a.copy())), a


_floats = st.lists(st.one_of(st.floats(-100, 100, allow_nan=False), st.just(float("nan")), st.integers(-5, 5).map(float)),
                   min_size=0, max_size=300)
_ops = st.sampled_from(["<", "<=", ">", ">=", "==", "!="])


@settings(max_examples=60, deadline=None, derandomize=True, database=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(values=_floats, op=_ops, scalar=st.one_of(st.floats(-100, 100, allow_nan=False), st.integers(-5, 5)),
       start=st.integers(0, 40), step=st.integers(1, 7), fill=st.booleans())
def test_float_vector_compares_and_masks_like_numpy(bnp, values, op, scalar, start, step, fill):
    import operator
    f = {"<": operator.lt, "<=": operator.le, ">": operator.gt, ">=": operator.ge, "==": operator.eq, "!=": operator.ne}[op]
    dv, a = _vec(values, np.float64)
    with np.errstate(invalid="ignore"):
        mask, ref = f(dv, scalar), f(a, scalar)
    assert np.array_equal(np.asarray(mask), ref) and mask.dtype == np.bool_ and len(mask) == len(a)
    assert mask.sum() == ref.sum() and mask.any() == ref.any() and mask.all() == ref.all()
    assert np.array_equal(np.flatnonzero(mask), np.flatnonzero(ref))
    assert np.array_equal(np.asarray(~mask), ~ref)
    other = np.arange(len(a)) % 3 == 0
    assert np.array_equal(np.asarray(mask & other), ref & other) and np.array_equal(np.asarray(mask | other), ref | other)
    assert np.array_equal(np.asarray(mask ^ (dv == dv)), ref ^ (a == a))
    mask[start::step] = fill
    ref[start::step] = fill
    assert np.array_equal(np.asarray(mask), ref)
    mask[:start] = not fill
    ref[:start] = not fill
    assert np.array_equal(np.asarray(mask), ref)
    assert np.array_equal(np.asarray(dv[mask]), a[ref], equal_nan=True)
    assert np.array_equal(np.asarray(dv[2:9]), a[2:9], equal_nan=True)
    if len(a):
        assert np.array_equal(np.asarray(dv * 2 - 1), a * 2 - 1, equal_nan=True)
        assert np.array_equal(np.isnan(dv), np.isnan(a))


@settings(max_examples=40, deadline=None, derandomize=True, database=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(values=st.lists(st.integers(0, 255), min_size=0, max_size=300), op=_ops,
       scalar=st.one_of(st.integers(-3, 260), st.floats(-3, 260, allow_nan=False)))
def test_uint8_and_int64_vectors_compare_like_numpy(bnp, values, op, scalar):
    import operator
    f = {"<": operator.lt, "<=": operator.le, ">": operator.gt, ">=": operator.ge, "==": operator.eq, "!=": operator.ne}[op]
    for dtype, scale in ((np.uint8, 1), (np.int64, 1000)):
        dv, a = _vec(np.asarray(values, dtype=np.int64) * scale if dtype == np.int64 else values, dtype)
        s = scalar * scale if dtype == np.int64 else scalar
        assert np.array_equal(np.asarray(f(dv, s)), f(a, s)), (dtype, op, s)
    dv, a = _vec(values, np.uint8)
    assert int(np.sum(dv)) == int(a.sum()) and dv.tolist() == a.tolist() and list(dv) == list(a)
    if len(a):
        assert dv[0] == a[0] and dv[-1] == a[-1] and float(np.mean(dv)) == float(a.mean()) and dv.max() == a.max()


@pytest.mark.parametrize("scalar", [float("nan"), float("inf"), float("-inf"), 1 << 70, -(1 << 70), 2.5, np.float64("nan"), 1 << 63])
def test_scalars_no_integer_vector_holds_compare_like_numpy(bnp, scalar):
    """NaN, infinities, fractions and ints beyond int64 against uint8 / int64 vectors: numpy answers with all-False / all-True
    masks; nothing raises (int(nan) and int(inf) do) and nothing is truncated on its way to the kernel"""
    import operator
    for dtype in (np.uint8, np.int64, np.float64):
        dv, a = _vec([0, 1, 2, 200, 255], dtype)
        for f in (operator.lt, operator.le, operator.gt, operator.ge, operator.eq, operator.ne):
            try:
                expect = f(a, scalar)
            except OverflowError:                            # (numpy itself refuses uint8 < 2**70 in some versions)
                continue
            assert np.array_equal(np.asarray(f(dv, scalar)), expect), (dtype, f, scalar)
