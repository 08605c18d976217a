"""Reading ``.gz`` input without leaving the device idle behind one host thread of ``gzip`` (the reference prefers
``isal.igzip`` where it is installed and falls back to ``gzip``: bionumpy/io/gzip_reading.py:1-4; isal is not available
here, so the time has to come from doing the inflate elsewhere than in the caller's thread — and, for BGZF members, from the
system's libdeflate where there is one: twice zlib's rate per thread, its CRC-32 fifteen times).

* **BGZF** (``bgzip``: a gzip file made of members of at most 64 KiB, each carrying its compressed size in a ``BC`` extra
  field — what sequencing pipelines write): the members are cut apart by their headers and inflated by a pool of threads
  (``zlib`` releases the interpreter lock), a few megabytes ahead of the reader, in order.
* any other gzip stream (one member or several): one background thread inflates ahead of the reader, so the inflate of
  the next batch runs while the caller uploads, scans and counts the current one.

Both are file-like objects with ``readinto`` / ``read`` / ``close`` / ``name``; ``NumpyFileReader`` treats them as
streams (``set_prepend_mode``), exactly like a ``gzip.GzipFile``.
"""
import collections
import os
import struct
import threading
import gzip
import zlib
from concurrent.futures import ThreadPoolExecutor

_GROUP_BYTES = 1 << 20          # compressed bytes of BGZF members handed to one pool task
_AHEAD_TASKS = 48               # tasks in flight (~100-150 MB of text ahead of the reader at usual ratios)
_STREAM_PIECE = 1 << 20         # compressed bytes fed to the single-stream inflater at a time
_STREAM_AHEAD = 64              # inflated pieces the background thread may be ahead


def _bgzf_block_size(header):
    """total size of the BGZF member that starts with ``header`` (>= 18 bytes), or None if it is not a BGZF member"""
    if len(header) < 18 or header[:4] != b"\x1f\x8b\x08\x04":
        return None
    xlen = struct.unpack_from("<H", header, 10)[0]
    pos, end = 12, 12 + xlen
    if end > len(header):
        return None
    while pos + 4 <= end:
        si1, si2, slen = header[pos], header[pos + 1], struct.unpack_from("<H", header, pos + 2)[0]
        if si1 == 66 and si2 == 67 and slen == 2:
            return struct.unpack_from("<H", header, pos + 4)[0] + 1
        pos += 4 + slen
    return None


class _LibDeflate:
    """libdeflate (the system's ``libdeflate.so.0``, where there is one) through ctypes: a whole-buffer inflate about twice as
    fast as zlib's and a CRC-32 an order of magnitude faster — what a BGZF member (<= 64 KiB, its text size in the trailer)
    needs, and nothing more.  The calls release the interpreter lock; a decompressor belongs to one thread."""

    def __init__(self):
        import ctypes as C
        lib = C.CDLL("libdeflate.so.0")
        lib.libdeflate_alloc_decompressor.restype = C.c_void_p
        lib.libdeflate_free_decompressor.argtypes = [C.c_void_p]
        lib.libdeflate_deflate_decompress.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                                      C.POINTER(C.c_size_t)]
        lib.libdeflate_deflate_decompress.restype = C.c_int
        lib.libdeflate_crc32.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t]
        lib.libdeflate_crc32.restype = C.c_uint32
        self._C, self._lib, self._local = C, lib, threading.local()
        self._by_thread, self._lock = {}, threading.Lock()     # thread ident -> its decompressor (freed when its pool closes)

    def _decompressor(self):
        d = getattr(self._local, "d", None)
        if d is None:
            d = self._local.d = self._lib.libdeflate_alloc_decompressor()
            if not d:
                raise MemoryError("libdeflate_alloc_decompressor")
            with self._lock:
                self._by_thread[threading.get_ident()] = d
        return d

    def inflate_members(self, blob, members, total):
        """members: (payload offset, payload size, text offset, text size, crc32, compressed offset) per member of ``blob``"""
        C = self._C
        out = bytearray(total)
        if total == 0:
            return out
        if isinstance(blob, bytearray):
            src = (C.c_char * len(blob)).from_buffer(blob)
            src_at = C.addressof(src)
        else:                                              # bytes: its own buffer, not a copy of it (``blob`` outlives the loop)
            src = C.c_char_p(blob)
            src_at = C.cast(src, C.c_void_p).value
        dst = (C.c_char * total).from_buffer(out)
        dst_at = C.addressof(dst)
        d, got = self._decompressor(), C.c_size_t(0)
        for p_off, p_size, t_off, t_size, crc, at in members:
            if t_size == 0:
                continue
            rc = self._lib.libdeflate_deflate_decompress(d, src_at + p_off, p_size, dst_at + t_off, t_size, C.byref(got))
            if rc != 0:                                    # LIBDEFLATE_BAD_DATA 1, SHORT_OUTPUT 2, INSUFFICIENT_SPACE 3
                what = {1: "invalid deflate data", 2: "less text than its trailer claims", 3: "more text than its trailer claims"}
                raise gzip.BadGzipFile("%s in a BGZF member at compressed offset %d" % (what.get(rc, "inflate error %d" % rc), at))
            if got.value != t_size:
                raise gzip.BadGzipFile("a BGZF member at compressed offset %d inflated to %d bytes, its trailer says %d"
                                       % (at, got.value, t_size))
            if self._lib.libdeflate_crc32(0, dst_at + t_off, t_size) != crc:
                raise gzip.BadGzipFile("CRC check failed in a BGZF member at compressed offset %d" % at)
        del src, dst
        return out

    def free_threads(self, idents):
        """frees the decompressors of threads that have ended (a closed reader's inflate pool)"""
        with self._lock:
            found = [self._by_thread.pop(i) for i in list(idents) if i in self._by_thread]
        for d in found:
            self._lib.libdeflate_free_decompressor(d)


def _load_libdeflate():
    if os.environ.get("BNPK_INFLATE", "") == "zlib":
        return None
    try:
        return _LibDeflate()
    except (OSError, AttributeError):
        return None


_libdeflate = _load_libdeflate()


def _inflate_members(blob):
    """the text of whole BGZF members laid end to end in ``blob``"""
    if _libdeflate is not None:
        members, pos, n, total = [], 0, len(blob), 0
        while pos < n:
            size = _bgzf_block_size(blob[pos:pos + 64])
            xlen = struct.unpack_from("<H", blob, pos + 10)[0]
            crc, isize = struct.unpack_from("<II", blob, pos + size - 8)
            if isize > 65536 or size - 20 - xlen < 0:        # (a BGZF member holds at most 64 KiB of text)
                raise gzip.BadGzipFile("not a BGZF member at compressed offset %d: %d bytes of text claimed" % (pos, isize))
            members.append((pos + 12 + xlen, size - 8 - 12 - xlen, total, isize, crc, pos))
            total += isize
            pos += size
        return _libdeflate.inflate_members(blob, members, total)
    out, pos, n = [], 0, len(blob)
    while pos < n:
        size = _bgzf_block_size(blob[pos:pos + 64])
        xlen = struct.unpack_from("<H", blob, pos + 10)[0]
        text = zlib.decompress(blob[pos + 12 + xlen:pos + size - 8], wbits=-15)
        # the member's trailer: CRC32 and length of its text, as gzip.GzipFile (and isal in the reference) check them —
        # a damaged bgzip file raises instead of being parsed
        crc, isize = struct.unpack_from("<II", blob, pos + size - 8)
        if zlib.crc32(text) != crc or (len(text) & 0xFFFFFFFF) != isize:
            raise gzip.BadGzipFile("CRC check failed in a BGZF member at compressed offset %d" % pos)
        out.append(text)
        pos += size
    return b"".join(out)


class _AheadReader:
    """readinto/read over a sequence of byte strings produced ahead of the reader (``_next_piece`` -> bytes, b"" at the end)"""

    def __init__(self, name):
        self.name = name
        self._cur, self._at = b"", 0
        self._done = False

    def readinto(self, target):
        view = memoryview(target).cast("B")
        got = 0
        while got < len(view):
            if self._at == len(self._cur):
                if self._done:
                    break
                self._cur, self._at = self._next_piece(), 0
                if not self._cur:
                    self._done = True
                    break
            n = min(len(view) - got, len(self._cur) - self._at)
            view[got:got + n] = memoryview(self._cur)[self._at:self._at + n]
            got += n
            self._at += n
        return got

    def read(self, n=-1):
        if n is None or n < 0:
            parts = [bytes(self._cur[self._at:])]
            self._cur, self._at = b"", 0
            while not self._done:
                piece = self._next_piece()
                if not piece:
                    self._done = True
                    break
                parts.append(piece)
            return b"".join(parts)
        buf = bytearray(n)
        return bytes(buf[:self.readinto(buf)])

    def readable(self):
        return True

    def __enter__(self):
        return self

    def __exit__(self, *args):
        self.close()


class BgzfReader(_AheadReader):
    """BGZF members inflated by a pool of threads, in order, a bounded number of tasks ahead of the reader"""

    def __init__(self, raw, name, n_threads):
        super().__init__(name)
        self._raw = raw
        self._thread_ids = set()
        self._pool = ThreadPoolExecutor(max_workers=max(1, n_threads), thread_name_prefix="bnpk-inflate",
                                        initializer=lambda: self._thread_ids.add(threading.get_ident()))
        self._tasks = collections.deque()
        self._pending = b""
        self._raw_done = False
        self._closed = False
        self._fill_tasks()

    def _fill_tasks(self):
        while not self._raw_done and len(self._tasks) < _AHEAD_TASKS:
            data = self._pending + self._raw.read(_GROUP_BYTES)
            if not data:
                self._raw_done = True
                break
            pos, n = 0, len(data)                           # whole members only; what is left over starts the next task
            while pos + 18 <= n:
                size = _bgzf_block_size(data[pos:pos + 64])
                if size is None:
                    raise OSError("%s: not a BGZF member at a member boundary (a plain gzip member inside a BGZF file?)" % self.name)
                if pos + size > n:
                    break
                pos += size
            if pos == 0:
                more = self._raw.read(1 << 16)
                if not more:
                    raise EOFError("%s: truncated BGZF member at the end of the file" % self.name)
                self._pending = data + more
                continue
            self._pending = data[pos:]
            self._tasks.append(self._pool.submit(_inflate_members, data[:pos]))

    def _next_piece(self):
        while True:
            if not self._tasks:
                self._fill_tasks()
                if not self._tasks:
                    if self._pending:
                        raise EOFError("%s: truncated BGZF member at the end of the file" % self.name)
                    return b""
            piece = self._tasks.popleft().result()
            self._fill_tasks()
            if piece:                                       # (the empty end-of-file member of bgzip inflates to nothing)
                return piece

    def close(self):
        for t in self._tasks:
            t.cancel()
        self._pool.shutdown(wait=True)
        if _libdeflate is not None and not self._closed:   # the pool's threads are gone: their decompressors go too
            _libdeflate.free_threads(self._thread_ids)
        self._closed = True
        self._raw.close()


class _LimitedRaw:
    """bytes [start, stop) of a file as a raw stream of their own (``read`` only)"""

    def __init__(self, filename, start, stop):
        self._f = open(filename, "rb")
        self._f.seek(start)
        self._left = max(0, stop - start)

    def read(self, n=-1):
        n = self._left if n is None or n < 0 else min(n, self._left)
        data = self._f.read(n) if n else b""
        self._left -= len(data)
        return data

    def close(self):
        self._f.close()


class BgzfShardReader(_AheadReader):
    """One rank's part of the text of a BGZF file (io/sharding.py).  The compressed file is cut at the first member that
    starts at or behind byte r * S / N (m_lo) and (r + 1) * S / N (m_hi); C(m) = the first record that starts in the text at
    or behind the first text byte of member m, found by scanning forwards from there (the byte in front of it is not
    known, so the member's first byte is not taken for the start of a line); this reader's text is [C(m_lo), C(m_hi)).
    Both neighbours compute C at the same member with the same scan, so every record is read by exactly one rank."""

    def __init__(self, filename, n_threads, shard, rule):
        super().__init__(filename)
        from .sharding import bgzf_member_at_or_after, first_record_start
        self._rule, self._first_record_start = rule, first_record_start
        size = os.path.getsize(filename)
        with open(filename, "rb") as probe:
            read_at = lambda off, n: os.pread(probe.fileno(), n, off)
            lo, hi = size * shard.rank // shard.world, size * (shard.rank + 1) // shard.world
            self._m_lo = bgzf_member_at_or_after(read_at, size, lo, _bgzf_block_size)
            self._m_hi = size if shard.rank == shard.world - 1 else \
                max(self._m_lo, bgzf_member_at_or_after(read_at, size, hi, _bgzf_block_size))
        self._size, self._n_threads = size, n_threads
        self._head_newlines = 0
        self._lines_before = None
        self._body = None
        self._tail = b""
        if self._m_hi > self._m_lo:
            # the text this rank reads BEHIND member m_hi: up to C(m_hi) (found now: a few lines of one member, usually)
            if self._m_hi < size:
                self._tail, _ = self._scan_from(self._m_hi)
            self._body = BgzfReader(_LimitedRaw(filename, self._m_lo, self._m_hi), filename, n_threads)
        self._state = "head" if self._m_lo > 0 else "body"
        if self._body is None:
            self._state = "done"

    def _scan_from(self, member):
        """(the text between the first byte of ``member`` and C(member), whether a record starts there at all)"""
        reader = BgzfReader(_LimitedRaw(self.name, member, self._size), self.name, min(2, self._n_threads))
        try:
            acc = bytearray()
            while True:
                piece = reader._next_piece()
                acc += piece
                import numpy as np
                cut = self._first_record_start(np.frombuffer(acc, dtype=np.uint8), False, self._rule, not piece)
                if cut is not None:
                    return (bytes(acc), False) if cut < 0 else (bytes(acc[:cut]), True)
        finally:
            reader.close()

    def _stream_piece(self):
        """the next piece of body ++ tail (b"" at the end)"""
        if self._body is not None:
            piece = self._body._next_piece()
            if piece:
                return piece
            self._body.close()
            self._body = None
            tail, self._tail = self._tail, b""
            return tail
        return b""

    def _next_piece(self):
        if self._state == "done":
            return b""
        if self._state == "body":
            piece = self._stream_piece()
            if not piece:
                self._state = "done"
            return piece
        # "head": drop the text in front of C(m_lo)
        import numpy as np
        acc = bytearray()
        while True:
            piece = self._stream_piece()
            acc += piece
            cut = self._first_record_start(np.frombuffer(acc, dtype=np.uint8), False, self._rule, not piece)
            if cut is None:
                continue
            if cut < 0:                                    # no record starts in this rank's part: it reads nothing
                self._head_newlines = acc.count(b"\n")
                self._state = "done"
                return b""
            self._head_newlines = acc[:cut].count(b"\n")
            self._state = "body"
            rest = bytes(acc[cut:])
            return rest if rest else self._next_piece()

    def lines_before(self):
        """lines of the file's text in front of this reader's first record (inflates the members in front of m_lo: only
        somebody who reports a line number asks)"""
        if self._lines_before is None:
            n = self._head_newlines
            if self._m_hi == self._m_lo and 0 < self._m_lo < self._size:     # an empty part lies at C(m_lo) like any other
                n = self._scan_from(self._m_lo)[0].count(b"\n")
            if self._m_lo > 0:
                reader = BgzfReader(_LimitedRaw(self.name, 0, self._m_lo), self.name, self._n_threads)
                try:
                    while True:
                        piece = reader._next_piece()
                        if not piece:
                            break
                        n += piece.count(b"\n")
                finally:
                    reader.close()
            self._lines_before = n
        return self._lines_before

    def close(self):
        if self._body is not None:
            self._body.close()
            self._body = None


class AheadGzipReader(_AheadReader):
    """a gzip stream (one member or several, any member size) inflated by ONE background thread ahead of the reader"""

    def __init__(self, raw, name):
        super().__init__(name)
        self._raw = raw
        self._queue = collections.deque()
        self._cv = threading.Condition()
        self._stop = False
        self._thread = threading.Thread(target=self._work, name="bnpk-inflate", daemon=True)
        self._thread.start()

    def _put(self, item):
        with self._cv:
            while len(self._queue) >= _STREAM_AHEAD and not self._stop:
                self._cv.wait()
            self._queue.append(item)
            self._cv.notify_all()

    def _work(self):
        try:
            inflater = zlib.decompressobj(wbits=31)
            started = False
            while not self._stop:
                data = self._raw.read(_STREAM_PIECE)
                if not data:
                    if started and not inflater.eof:
                        raise EOFError("%s: compressed file ended before the end-of-stream marker was reached" % self.name)
                    break
                while data and not self._stop:
                    if inflater.eof:                         # the next member of a multi-member file
                        inflater = zlib.decompressobj(wbits=31)
                    started = True
                    text = inflater.decompress(data)
                    data = inflater.unused_data if inflater.eof else b""
                    if data and not data.strip(b"\x00"):    # (zero padding behind the last member, as gzip tolerates)
                        data = b""
                    if text:
                        self._put(text)
            self._put(b"")
        except BaseException as e:                           # noqa: BLE001  (re-raised in the reader's thread)
            self._put(e)

    def _next_piece(self):
        with self._cv:
            while not self._queue:
                self._cv.wait()
            item = self._queue.popleft()
            self._cv.notify_all()
        if isinstance(item, BaseException):
            raise item
        return item

    def close(self):
        with self._cv:
            self._stop = True
            self._queue.clear()
            self._cv.notify_all()
        self._thread.join()
        self._raw.close()


def is_bgzf(filename):
    with open(filename, "rb") as raw:
        return _bgzf_block_size(raw.read(64)) is not None


def open_gzip_for_reading(filename, n_threads=None, shard=None, rule=None):
    """``gzip.open(filename, "rb")`` for the chunk reader: BGZF -> thread pool, anything else -> one thread ahead.
    shard, rule: (BGZF only) one rank's part of the text (io/sharding.py: Shard, RecordRule)."""
    if n_threads is None:
        n_threads = int(os.environ.get("BNPK_INFLATE_THREADS", min(16, os.cpu_count() or 1)))
    if shard is not None:
        assert is_bgzf(filename), "only BGZF files can be entered in the middle"
        return BgzfShardReader(filename, n_threads, shard, rule)
    raw = open(filename, "rb")
    head = raw.read(64)
    raw.seek(0)
    if _bgzf_block_size(head) is not None:
        return BgzfReader(raw, filename, n_threads)
    return AheadGzipReader(raw, filename)
