"""HBM plumbing: one bnpk context per process, torch as the HBM allocator / stream provider.

``HArray`` is a flat typed buffer that lives in HBM (a torch CUDA tensor) and/or on the host
(numpy).  Compute always happens on the device copy; the host copy exists for presentation
(indexing, printing, ``.raw()``) and for data that entered from Python objects.
"""
import ctypes as C

import numpy as np

from . import _native
from ._native import lib, check

_torch = None


def torch():
    global _torch
    if _torch is None:
        import torch as _t
        _torch = _t
    return _torch


_NP2TORCH = None


def _torch_dtype(np_dtype):
    global _NP2TORCH
    t = torch()
    if _NP2TORCH is None:
        _NP2TORCH = {np.dtype(np.uint8): t.uint8, np.dtype(np.int64): t.int64, np.dtype(np.int32): t.int32,
                     np.dtype(np.bool_): t.bool, np.dtype(np.float64): t.float64}
    return _NP2TORCH[np.dtype(np_dtype)]


class Device:
    """The process-wide GPU context (one process per GPU)."""

    _instance = None

    def __init__(self, index=None):
        t = torch()
        if not t.cuda.is_available() or lib.bnpk_device_count() <= 0:
            raise _native.BnpkError(-5, "bionumpy_amd needs a gfx950 GPU; there is no CPU fallback")
        if index is None:
            index = t.cuda.current_device()
        self.index = int(index)
        t.cuda.set_device(self.index)
        self.tdev = t.device("cuda", self.index)
        ctx = C.c_void_p()
        check(lib.bnpk_ctx_create(self.index, C.byref(ctx)))
        self.ctx = ctx
        self._raw_stream = None
        self._tindex = self.index

    @classmethod
    def get(cls):
        if cls._instance is None:
            cls._instance = Device()
        return cls._instance

    @classmethod
    def reset(cls):
        if cls._instance is not None:
            lib.bnpk_ctx_destroy(cls._instance.ctx)
            cls._instance = None

    # -- streams / memory --------------------------------------------------------------------
    def stream(self):
        # torch's current stream of this device as a raw handle (the C-level getter where torch has it: a launch asks for the
        # stream every time — the caller may have switched it — and the Stream object of the public call costs 4 us)
        raw = self._raw_stream
        if raw is None:
            t = torch()
            raw = getattr(t._C, "_cuda_getCurrentRawStream", None)
            if raw is None:
                raw = lambda index: t.cuda.current_stream(index).cuda_stream
            self._raw_stream = raw
            self._tindex = self.tdev.index if self.tdev.index is not None else t.cuda.current_device()
        return C.c_void_p(raw(self._tindex))

    def empty(self, n, dtype):
        return torch().empty(int(n), dtype=_torch_dtype(dtype), device=self.tdev)

    def zeros(self, n, dtype):
        return torch().zeros(int(n), dtype=_torch_dtype(dtype), device=self.tdev)

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        t = torch().from_numpy(arr) if arr.size else torch().empty(0, dtype=_torch_dtype(arr.dtype))
        return t.to(self.tdev)

    def torch_cat(self, tensors):
        return torch().cat(list(tensors))

    def synchronize(self):
        torch().cuda.synchronize(self.tdev)

    def info(self):
        name = C.create_string_buffer(64)
        cus = C.c_int()
        hbm = C.c_int64()
        check(lib.bnpk_device_info(self.ctx, name, C.byref(cus), C.byref(hbm)), self.ctx)
        return {"arch": name.value.decode(), "compute_units": cus.value, "hbm_bytes": hbm.value}

    # -- per-kernel timers ---------------------------------------------------------------------
    def prof_enable(self, on=True):
        check(lib.bnpk_prof_enable(self.ctx, 1 if on else 0), self.ctx)

    def prof_reset(self):
        check(lib.bnpk_prof_reset(self.ctx), self.ctx)

    def prof_report(self):
        n = lib.bnpk_prof_count(self.ctx)
        if n < 0:
            check(n, self.ctx)
        out = {}
        name = C.create_string_buffer(64)
        ms = C.c_double()
        launches = C.c_int64()
        for i in range(n):
            check(lib.bnpk_prof_get(self.ctx, i, name, C.byref(ms), C.byref(launches)), self.ctx)
            out[name.value.decode()] = {"total_ms": ms.value, "launches": launches.value}
        return out


def ptr(t):
    """device pointer of a torch tensor (or None) as c_void_p"""
    if t is None:
        return C.c_void_p(0)
    return C.c_void_p(t.data_ptr())


class HArray:
    """Flat buffer with an HBM copy (torch tensor) and/or a host copy (numpy array)."""

    __slots__ = ("_np", "_t")

    def __init__(self, host=None, dev=None):
        assert host is not None or dev is not None
        self._np = None if host is None else np.ascontiguousarray(host)
        self._t = dev

    @property
    def size(self):
        return int(self._np.size) if self._np is not None else int(self._t.numel())

    def __len__(self):
        return self.size

    @property
    def dtype(self):
        if self._np is not None:
            return self._np.dtype
        return np.dtype(str(self._t.dtype).replace("torch.", ""))

    @property
    def on_device(self):
        return self._t is not None

    def host(self):
        if self._np is None:
            self._np = self._t.cpu().numpy()
        return self._np

    def dev(self):
        if self._t is None:
            self._t = Device.get().upload(self._np)
        return self._t

    def drop_host(self):
        if self._t is not None:
            self._np = None


def as_bool(h):
    """a 0/1 uint8 HArray as a bool one over the same memory (flags stay where the kernel wrote them)"""
    if h.dtype == np.bool_:
        return h
    if h.on_device:
        return HArray(dev=h.dev().view(torch().bool))
    return HArray(host=h.host().view(np.bool_))


def as_u8(h):
    """a bool HArray as 0/1 bytes over the same memory"""
    if h.dtype == np.uint8:
        return h
    if h.on_device:
        return HArray(dev=h.dev().view(torch().uint8))
    return HArray(host=h.host().view(np.uint8))


class SharedSlice(HArray):
    """a part of a device array that others hold parts of too: whoever wants to write into it makes a copy first"""

    __slots__ = ()


class LazyHArray(HArray):
    """An int64 HArray of a known size whose elements are only produced (by ``make() -> HArray``) when somebody reads them: the
    row offsets of a result nobody may ever index by row (a chunk's k-mers that go straight into a histogram)."""

    __slots__ = ("_make", "_n")

    def __init__(self, n, make):
        self._np, self._t, self._n, self._make = None, None, int(n), make

    def _fill(self):
        if self._make is not None:
            made, self._make = self._make(), None
            self._np, self._t = made._np, made._t

    @property
    def size(self):
        return self._n

    @property
    def dtype(self):
        return np.dtype(np.int64)

    @property
    def on_device(self):
        return True

    def host(self):
        self._fill()
        return HArray.host(self)

    def dev(self):
        self._fill()
        return HArray.dev(self)

    def drop_host(self):
        if self._make is None:
            HArray.drop_host(self)


def as_harray(x, dtype=None):
    if isinstance(x, HArray) or (hasattr(x, "host") and hasattr(x, "dev")):
        return x                      # HArray or an HArray-like lazy buffer (e.g. packed DNA)
    if _torch is not None and isinstance(x, _torch.Tensor):
        return HArray(dev=x)
    a = np.asarray(x)
    if dtype is not None:
        a = a.astype(dtype, copy=False)
    return HArray(host=a)
