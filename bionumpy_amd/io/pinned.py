"""Pinned host staging for the chunk reader: file bytes are read straight into page-locked buffers
(``bnpk_host_alloc`` = hipHostMalloc) and streamed to HBM with ``hipMemcpyAsync`` (``bnpk_copy_h2d_async``).

Every reader owns two buffers that alternate, so that the next ``file.readinto`` fills one buffer while the DMA out of
the other is still in flight; a buffer is only reused after its copy has completed (stream sync).
This replaces ``np.frombuffer(file.read(n))`` + ``cp.asanyarray(chunk)`` of the reference
(bionumpy/io/parser.py:203-206, bionumpy/cupy_compatible/parser.py:11-17).
"""
import ctypes as C
import threading
import weakref

import numpy as np

from .._native import lib, check

_live = weakref.WeakSet()          # every pinned buffer that exists: lets an upload recognise page-locked memory


class PinnedBuffer:
    def __init__(self, nbytes):
        ptr = C.c_void_p()
        check(lib.bnpk_host_alloc(nbytes, C.byref(ptr)))
        self.ptr = ptr
        self.nbytes = nbytes
        self.array = np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr.value))
        self.in_flight = False
        _live.add(self)

    def wait(self):
        """until the H2D copy out of this buffer (if any) has landed"""
        if self.in_flight:
            check(lib.bnpk_stream_sync(getattr(self, "stream", None)))
            self.in_flight = False

    def free(self):
        if self.ptr:
            self.wait()
            lib.bnpk_host_free(self.ptr)
            self.ptr = None
            self.array = np.zeros(0, dtype=np.uint8)   # (no view of freed memory is handed out again)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# Page-locking a gigabyte costs ~0.1 s: buffers a reader gives back are kept for the next reader of the process (up to
# _CACHE_BYTES), so that opening a file does not pay for its staging again.
_CACHE_BYTES = 8 << 30
_cache, _cache_lock = [], threading.Lock()


def _from_cache(nbytes):
    with _cache_lock:
        fits = [b for b in _cache if b.ptr and b.nbytes >= nbytes]
        if not fits:
            return None
        buf = min(fits, key=lambda b: b.nbytes)
        _cache.remove(buf)
        return buf


def _to_cache(buf):
    if not buf.ptr:                # already freed (the collector may finalise a buffer before the pool that holds it)
        return
    buf.wait()
    with _cache_lock:
        if sum(b.nbytes for b in _cache) + buf.nbytes <= _CACHE_BYTES:
            _cache.append(buf)
            return
    buf.free()


class PinnedPool:
    """page-locked staging buffers that take turns, grown on demand"""

    def __init__(self, n_buffers=2):
        self._buffers = [None] * n_buffers
        self._next = 0

    def acquire(self, nbytes):
        """the next buffer in turn, at least nbytes large, with no copy out of it in flight"""
        i = self._next
        self._next = (self._next + 1) % len(self._buffers)
        buf = self._buffers[i]
        if buf is not None:
            buf.wait()
        if buf is None or buf.nbytes < nbytes:
            if buf is not None:
                _to_cache(buf)
            buf = _from_cache(nbytes) or PinnedBuffer(max(nbytes + (nbytes >> 3), 1 << 20))
            self._buffers[i] = buf
        return buf

    def release(self):
        for i, buf in enumerate(self._buffers):
            if buf is not None:
                _to_cache(buf)
                self._buffers[i] = None

    def __del__(self):                                 # a reader dropped without close() still hands its buffers on
        try:
            self.release()
        except Exception:
            pass


def owner_of(array):
    """the pinned buffer a numpy view lives in (None for ordinary pageable arrays)"""
    if not isinstance(array, np.ndarray) or array.size == 0:
        return None
    addr = array.__array_interface__["data"][0]
    for buf in list(_live):
        if buf.ptr and buf.ptr.value <= addr and addr + array.nbytes <= buf.ptr.value + buf.nbytes:
            return buf
    return None
