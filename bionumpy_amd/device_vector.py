"""Per-row results that stay in HBM until somebody looks at them.

``np.mean(chunk.quality, axis=1)`` of a 50 M-read chunk is 400 MB of float64; the read filters of the reference
(scripts/small_example.py:36-46) compare it with a threshold, combine masks and index the chunk with the result.  Through
numpy arrays every step crossed PCIe (quality filter: 400 ms for 45 ms of kernels, round-1 VERDICT).  ``DeviceVector`` is
what the row reductions of a ragged array return instead: a 1-D array-like whose comparisons, mask logic, strided mask
assignment, ``sum`` / ``np.flatnonzero`` of masks and ``chunk[mask]`` run as kernels on the device copy
(bnpk_vec_compare / bnpk_mask_logic / bnpk_mask_fill / bnpk_byte_census + bnpk_byte_positions), and which turns into a
plain numpy array the moment anything else is asked of it (``__array__``, attribute access, iteration, arithmetic) —
the values and the results are those numpy would give either way.
"""
import math

import numpy as np

from .device import HArray
from .ops import get_ops

_OPS = {"__lt__": "<", "__le__": "<=", "__gt__": ">", "__ge__": ">=", "__eq__": "==", "__ne__": "!="}


class DeviceVector:
    __array_priority__ = 1000
    __hash__ = None

    def __init__(self, data, dtype=None):
        self._data = data                              # HArray; bool vectors are kept as 0/1 uint8
        self._dtype = np.dtype(dtype if dtype is not None else data.dtype)

    # -- what numpy code looks at first ----------------------------------------------------------------------------
    @property
    def dtype(self):
        return self._dtype

    @property
    def shape(self):
        return (self._data.size,)

    @property
    def size(self):
        return self._data.size

    ndim = 1

    def __len__(self):
        return self._data.size

    def harray(self):
        return self._data

    def host(self):
        a = self._data.host()
        return a.view(np.bool_) if self._dtype == np.bool_ else a

    def __array__(self, dtype=None, copy=None):
        a = self.host()
        return a if dtype is None else a.astype(dtype, copy=False)

    def __repr__(self):                                # prints as the numpy array the reference returns here
        return repr(self.host())                       # (docs_source/source/sequences.rst:173-174: ``array([4, 0])``)

    def __str__(self):
        return str(self.host())

    def __iter__(self):
        return iter(self.host())

    def __bool__(self):                                # (as numpy: ambiguous for more than one element)
        return bool(self.host())

    def __getattr__(self, name):                       # anything else: the host array's (astype, tolist, mean, std, ...)
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.host(), name)

    # -- indexing ----------------------------------------------------------------------------------------------------
    def __getitem__(self, idx):
        if isinstance(idx, slice):
            start, stop, step = idx.indices(self.size)
            if step == 1:
                return DeviceVector(get_ops().slice_copy(self._data, start, max(stop, start)), self._dtype)
        if isinstance(idx, DeviceVector) and idx.dtype == np.bool_:
            idx = idx.host()
        return self.host()[idx]

    def __setitem__(self, idx, value):
        if self._dtype == np.bool_ and isinstance(idx, slice) and isinstance(value, (bool, np.bool_, int, np.integer)):
            start, stop, step = idx.indices(self.size)
            if step >= 1:
                count = max(0, (stop - start + step - 1) // step)
                get_ops().mask_fill(self._data, start, step, count, bool(value))
                self._drop_host()
                return
        a = np.array(self.host())                      # anything else: on the host copy, then back
        a[idx.host() if isinstance(idx, DeviceVector) else idx] = value
        self._data = HArray(host=a.view(np.uint8) if self._dtype == np.bool_ else a)

    def _drop_host(self):
        if self._data.on_device:
            self._data.drop_host()

    # -- comparisons with a scalar: a mask on the device ---------------------------------------------------------
    def _compare(self, name, other):
        if isinstance(other, (int, float, np.integer, np.floating)) and not isinstance(other, bool) and \
                self._dtype in (np.float64, np.int64, np.uint8):
            if self._dtype == np.float64:
                on_device = not isinstance(other, (int, np.integer)) or abs(int(other)) <= (1 << 53)   # (exact as a double)
            else:
                # an integer vector against a scalar that is an integer of its range; anything else — a fraction, NaN,
                # an infinity, an int beyond int64 — is numpy's business (all-False / all-True masks, no exceptions)
                if isinstance(other, (int, np.integer)):
                    whole, value = True, int(other)
                else:
                    whole = math.isfinite(float(other)) and float(other) == int(other)
                    value = int(other) if whole else 0
                lo, hi = (0, 255) if self._dtype == np.uint8 else (-(1 << 63), (1 << 63) - 1)
                on_device = whole and lo <= value <= hi
            if on_device:
                return DeviceVector(get_ops().vec_compare(self._data, _OPS[name], other), np.bool_)
        other = other.host() if isinstance(other, DeviceVector) else other
        return getattr(self.host(), name)(other)

    def __lt__(self, other): return self._compare("__lt__", other)
    def __le__(self, other): return self._compare("__le__", other)
    def __gt__(self, other): return self._compare("__gt__", other)
    def __ge__(self, other): return self._compare("__ge__", other)
    def __eq__(self, other): return self._compare("__eq__", other)
    def __ne__(self, other): return self._compare("__ne__", other)

    # -- mask logic ------------------------------------------------------------------------------------------------------
    def _logic(self, op, other):
        if self._dtype == np.bool_:
            if isinstance(other, np.ndarray) and other.dtype == np.bool_ and other.shape == self.shape:
                other = DeviceVector(HArray(host=other.view(np.uint8)), np.bool_)
            if isinstance(other, DeviceVector) and other.dtype == np.bool_ and other.size == self.size:
                return DeviceVector(get_ops().mask_logic(self._data, other._data, op), np.bool_)
        other = other.host() if isinstance(other, DeviceVector) else other
        return {"and": np.bitwise_and, "or": np.bitwise_or, "xor": np.bitwise_xor}[op](self.host(), other)

    def __and__(self, other): return self._logic("and", other)
    def __or__(self, other): return self._logic("or", other)
    def __xor__(self, other): return self._logic("xor", other)
    __rand__, __ror__, __rxor__ = __and__, __or__, __xor__

    def __invert__(self):
        if self._dtype == np.bool_:
            return DeviceVector(get_ops().mask_logic(self._data, None, "not"), np.bool_)
        return ~self.host()

    # -- reductions of masks -------------------------------------------------------------------------------------------
    def nonzero_rows(self):
        """np.flatnonzero of a mask, left on the device (HArray of int64)"""
        assert self._dtype == np.bool_
        return get_ops().mask_rows(self._data)[0]

    def sum(self, *args, **kwargs):
        if self._dtype == np.bool_ and not args and not kwargs:
            return get_ops().mask_rows(self._data)[1]
        return self.host().sum(*args, **kwargs)

    def any(self):
        return bool(self.sum() > 0) if self._dtype == np.bool_ else bool(self.host().any())

    def all(self):
        return bool(self.sum() == self.size) if self._dtype == np.bool_ else bool(self.host().all())

    # -- everything else goes through numpy on the host copy ------------------------------------------------------------
    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        inputs = tuple(x.host() if isinstance(x, DeviceVector) else x for x in inputs)
        if "out" in kwargs:
            kwargs["out"] = tuple(x.host() if isinstance(x, DeviceVector) else x for x in kwargs["out"])
        return getattr(ufunc, method)(*inputs, **kwargs)

    def __array_function__(self, func, types, args, kwargs):
        if func is np.flatnonzero and len(args) == 1 and args[0] is self and self._dtype == np.bool_:
            return self.nonzero_rows().host()
        if func in (np.sum, np.count_nonzero) and len(args) == 1 and args[0] is self and not kwargs and self._dtype == np.bool_:
            return self.sum()

        def plain(x):
            if isinstance(x, DeviceVector):
                return x.host()
            if isinstance(x, (list, tuple)):
                return type(x)(plain(y) for y in x)
            return x
        return func(*plain(args), **{k: plain(v) for k, v in kwargs.items()})


def _binary(name):
    def op(self, other):
        other = other.host() if isinstance(other, DeviceVector) else other
        return getattr(self.host(), name)(other)
    op.__name__ = name
    return op


for _name in ("__add__", "__radd__", "__sub__", "__rsub__", "__mul__", "__rmul__", "__truediv__", "__rtruediv__",
              "__floordiv__", "__rfloordiv__", "__mod__", "__pow__", "__neg__", "__abs__"):
    if _name in ("__neg__", "__abs__"):
        setattr(DeviceVector, _name, (lambda n: lambda self: getattr(self.host(), n)())(_name))
    else:
        setattr(DeviceVector, _name, _binary(_name))
