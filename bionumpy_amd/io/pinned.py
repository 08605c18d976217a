"""Pinned host staging for the chunk reader: file bytes are read straight into page-locked buffers
(``bnpk_host_alloc`` = hipHostMalloc) and streamed to HBM with ``hipMemcpyAsync`` (``bnpk_copy_h2d_async``).

Two buffers alternate so that the next ``file.readinto`` can fill one buffer while the DMA out of the
other is still in flight; a buffer is only reused after its copy has completed (stream sync).
This replaces ``np.frombuffer(file.read(n))`` + ``cp.asanyarray(chunk)`` of the reference
(bionumpy/io/parser.py:203-206, bionumpy/cupy_compatible/parser.py:11-17).
"""
import ctypes as C

import numpy as np

from .._native import lib, check


class PinnedBuffer:
    def __init__(self, nbytes):
        ptr = C.c_void_p()
        check(lib.bnpk_host_alloc(nbytes, C.byref(ptr)))
        self.ptr = ptr
        self.nbytes = nbytes
        self.array = np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr.value))
        self.in_flight = False

    def free(self):
        if self.ptr:
            lib.bnpk_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedPool:
    """two page-locked staging buffers, grown on demand"""

    def __init__(self, n_buffers=2):
        self._buffers = [None] * n_buffers
        self._next = 0

    def acquire(self, nbytes):
        i = self._next
        self._next = (self._next + 1) % len(self._buffers)
        buf = self._buffers[i]
        if buf is not None and buf.in_flight:
            check(lib.bnpk_stream_sync(getattr(buf, "stream", None)))   # its H2D copy must have landed first
            buf.in_flight = False
        if buf is None or buf.nbytes < nbytes:
            if buf is not None:
                buf.free()
            buf = PinnedBuffer(max(nbytes, 1 << 20))
            self._buffers[i] = buf
        return buf

    def owner_of(self, array):
        """the pinned buffer a numpy view lives in (None for ordinary pageable arrays)"""
        if not isinstance(array, np.ndarray) or array.size == 0:
            return None
        addr = array.__array_interface__["data"][0]
        for buf in self._buffers:
            if buf is not None and buf.ptr and buf.ptr.value <= addr and addr + array.nbytes <= buf.ptr.value + buf.nbytes:
                return buf
        return None


_pool = None


def pool():
    global _pool
    if _pool is None:
        _pool = PinnedPool()
    return _pool


def read_into_pinned(file_obj, nbytes, headroom=2):
    """``file.read(nbytes)`` into a pinned buffer; returns a uint8 numpy view of the bytes read.
    ``headroom`` spare bytes after the data let the reader append the EOF newline / marker in place."""
    buf = pool().acquire(nbytes + headroom)
    view = memoryview(buf.array)[:nbytes]
    got = 0
    while got < nbytes:                                 # gzip/BufferedReader may return short reads
        n = file_obj.readinto(view[got:])
        if not n:
            break
        got += n
    return buf.array[:got], buf
