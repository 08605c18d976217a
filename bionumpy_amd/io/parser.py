"""NumpyFileReader: from a file object to device buffers of complete entries.

What a caller of the reference's reader (bionumpy/io/parser.py:36-206) can observe is kept — ``read`` / ``read_chunk``
/ ``read_chunks`` / iteration, which entries a chunk of ``min_chunk_size`` bytes holds (a plain file is read in
windows of ``min_chunk_size`` bytes that start at the first unconsumed byte, a gzip stream gets ``min_chunk_size``
new bytes on top of what was left over), the newline (and '>' for multi-line FASTA) appended at the end of the file,
``Exception("No complete entry found")`` beyond ``max_chunk_size``, ``n_bytes_read`` / ``n_lines_read`` and the
``FormatException.line_number`` that counts from the start of the file — but the mechanism is built for batches of
gigabytes going to a device, not for 5 MB numpy chunks:

* the bytes are read straight into page-locked staging buffers, three of them: a plain file by ``os.preadv`` of disjoint
  slices from a few threads (one thread copies out of the page cache at ~10 GB/s), anything else by ``readinto``; the
  end-of-file newline / marker is written in place behind the data;
* under ``read_chunks`` with batches of 32 MB and more of a plain file, background threads read the next two batches' new
  bytes while the current batch is uploaded, scanned and worked on by the caller (they are read before the current
  batch's left-over is known, so such a batch is its left-over plus ``min_chunk_size`` new bytes; smaller batches keep
  the reference's window arithmetic exactly);
* on the GPU that background thread also starts the upload: as soon as the new bytes are in their page-locked buffer they
  go to a device staging buffer with ``hipMemcpyAsync`` on a copy stream, while the caller is still working on the batch
  before (the copy of a 256 MB batch takes as long as counting its k-mers: one after the other they were half of a
  file-to-histogram run).  When the caller takes the batch, the left-over in front of it and the end-of-file bytes behind
  it are a few kilobytes more, and the batch is put together on the device;
* nothing is ever seeked back or re-read, for plain files and gzip streams alike: the bytes behind the last complete
  entry (less than one entry) are carried over, by offset, to the front of the other staging buffer;
* whether a batch holds a complete entry, and where its last one ends, is decided by the device scan of the buffer
  class (``from_raw_buffer``) on the uploaded bytes — the host never looks at them (the reference counts the
  newlines of every chunk with numpy before it parses it).
"""
import collections
import io
import os
import threading

import numpy as np

from ..exceptions import FormatException, IncompleteEntryException
from ..ops import get_ops

NEWLINE = 10
_BIG = 32 << 20                    # batches from this size on: parallel file reads, read-ahead under read_chunks
_READ_THREADS = int(os.environ.get("BNPK_READ_THREADS", min(16, os.cpu_count() or 1)))    # 1 = the calling thread only
_READ_AHEAD = os.environ.get("BNPK_READ_AHEAD", "1") != "0"
_FRONT = 4 << 20                   # room kept in front of a read-ahead for the bytes left over by the batch before it
_EARLY_UPLOAD = os.environ.get("BNPK_EARLY_UPLOAD", "1") != "0"     # the read-ahead thread starts the batch's upload too
_READ_DEPTH = int(os.environ.get("BNPK_READ_DEPTH", 2))          # batches read ahead under read_chunks (1 or 2)
_PIECE = int(os.environ.get("BNPK_PIECE_MB", 16)) << 20                   # ... piece by piece, while the reading threads are still at work
# read_chunks with the reference's small windows (its default is 5 MB): the file still comes in, goes up and is scanned in
# batches of _WINDOW_BATCH bytes, and the chunks are cut out of the scanned batch on the device (bnpk_window_cuts) — the same
# chunks, entry for entry, as a reader that reads, uploads and scans 5 MB at a time
_WINDOWED = os.environ.get("BNPK_WINDOWED", "1") != "0"
_WINDOW_MIN = int(os.environ.get("BNPK_WINDOW_MIN", 256 << 10))          # windows below this size keep the plain loop
_WINDOW_BATCH = int(os.environ.get("BNPK_WINDOW_BATCH_MB", 128)) << 20
_SHARE = os.environ.get("BNPK_WINDOW_SHARE", "1") != "0"                   # ... and the sequence column of a batch is encoded once for its chunks
_WINDOW_CUTS = 200                                                       # chunks cut out of one batch at most (bnpk_window_cuts takes 256)


class _Staging:
    """where the bytes of a batch are gathered: page-locked buffers on the GPU path (io/pinned.py), plain numpy arrays
    under the host-only test backend.  ``room(n)`` returns a writable uint8 array of at least n bytes that does not
    alias the arrays returned by the two calls before it."""

    def __init__(self):
        self._pool = None
        if not getattr(get_ops(), "host_only", False):
            from .pinned import PinnedPool
            self._pool = PinnedPool(3)                       # the batch in use and two being read ahead

    def room(self, n):
        if self._pool is not None:
            return self._pool.acquire(n).array
        return np.empty(n, dtype=np.uint8)

    def release(self):
        if self._pool is not None:
            self._pool.release()


_early_cache = []                  # _EarlyUpload objects of closed readers
_EARLY_KEEP_BYTES = 1 << 30        # device staging a closed reader may leave behind for the next one


class _EarlyUpload:
    """Device staging for the read-ahead: three HBM buffers that take turns, a copy stream, and the events that order the
    background thread's ``hipMemcpyAsync`` into a buffer behind the caller's copy out of it three batches earlier."""

    def __init__(self):
        import torch
        from ..device import Device
        from .._native import lib, check
        self.torch, self.dev, self.lib, self.check = torch, Device.get(), lib, check
        self.stream = torch.cuda.Stream(self.dev.tdev)
        self.buffers, self.free_events, self.turn = [None] * 3, [None] * 3, 0

    def take(self, nbytes):
        """(caller's thread) the next staging buffer in turn, at least nbytes large -> its index"""
        i = self.turn
        self.turn = (self.turn + 1) % len(self.buffers)
        if self.buffers[i] is None or self.buffers[i].numel() < nbytes:
            self.buffers[i] = self.torch.empty(int(nbytes), dtype=self.torch.uint8, device=self.dev.tdev)
            ev = self.torch.cuda.Event()
            ev.record(self.torch.cuda.current_stream(self.dev.tdev))     # (the allocation's stream)
            self.free_events[i] = ev
        return i

    def begin(self, i):
        """(background thread, before the first upload of a batch) the copy stream waits until the caller's copy out of
        buffer i, two batches ago, is done"""
        self.torch.cuda.set_device(self.dev.tdev)
        if self.free_events[i] is not None:
            self.stream.wait_event(self.free_events[i])

    def target(self, i, offset):
        """(device pointer of buffers[i][offset:], copy stream) for bnpk_pread_parallel"""
        import ctypes as C
        return C.c_void_p(self.buffers[i].data_ptr() + offset), C.c_void_p(self.stream.cuda_stream)

    def upload(self, i, host, offset):
        """(background thread) host[:] -> buffers[i][offset:], asynchronously on the copy stream"""
        import ctypes as C
        dst = C.c_void_p(self.buffers[i].data_ptr() + offset)
        src = C.c_void_p(host.__array_interface__["data"][0])
        self.check(self.lib.bnpk_copy_h2d_async(dst, src, host.size, C.c_void_p(self.stream.cuda_stream)))

    def finish(self, i, owner):
        """(background thread, after the last upload of a batch) -> the event behind the copies"""
        import ctypes as C
        if owner is not None:
            owner.in_flight, owner.stream = True, C.c_void_p(self.stream.cuda_stream)
        ev = self.torch.cuda.Event()
        ev.record(self.stream)
        return ev

    def assemble(self, i, event, room, first, n, front, got):
        """(caller's thread) the batch room[first:n] as one fresh device array: the new bytes room[front:front + got] are
        in buffers[i] already, what lies in front of and behind them (left-over, end-of-file bytes) comes from the host"""
        import ctypes as C
        from ..device import HArray
        from .pinned import owner_of
        torch, lib = self.torch, self.lib
        cur = torch.cuda.current_stream(self.dev.tdev)
        t = torch.empty(n - first, dtype=torch.uint8, device=self.dev.tdev)
        cur.wait_event(event)
        head = front - first
        t[head:head + got].copy_(self.buffers[i][front:front + got])
        ev = torch.cuda.Event()
        ev.record(cur)
        self.free_events[i] = ev
        handle = C.c_void_p(cur.cuda_stream)
        base = room.__array_interface__["data"][0]
        for a, b in ((first, front), (front + got, n)):
            if b > a:
                self.check(lib.bnpk_copy_h2d_async(C.c_void_p(t.data_ptr() + (a - first)), C.c_void_p(base + a), b - a, handle))
        owner = owner_of(room)
        if owner is not None:
            owner.in_flight, owner.stream = True, handle                # (the reader syncs this stream before it reuses the room)
        return HArray(dev=t)


class _LinesBefore:
    """the lines of the file in front of a chunk — those this reader has read so far (known) plus those in front of the part
    of the file it reads (counted, once, if anybody ever adds this to a line number: only exceptions do)"""

    def __init__(self, local, reader):
        self._local, self._reader = int(local), reader

    def __int__(self):
        return self._local + self._reader.lines_before_range()

    __index__ = __int__

    def __radd__(self, other):
        return other + int(self)

    __add__ = __radd__


class NumpyFileReader:
    _start, _stop, _chunk_modulo, _lines_before, _lines_before_value = 0, None, None, None, None     # (a reader of a whole file)
    _chunk_index = 0                      # chunks read through read_chunk() so far (chunk-modulo shards)

    def __init__(self, file_obj, buffer_type, has_header=False, byte_range=None, chunk_modulo=None, lines_before=None):
        """byte_range: (start, stop) — the reader's file is bytes [start, stop) of the plain file ``file_obj`` (a rank's part
        of a sharded file, cut at record starts: io/sharding.py); behind ``stop`` the file is over for this reader.
        chunk_modulo: (r, n) — ``read_chunks`` yields only the chunks i with i % n == r (the shard of a stream that cannot
        be entered in the middle: a single-member gzip); the others are read and skipped.
        lines_before: () -> lines of the file in front of this reader's part, for readers of other things than plain byte
        ranges (BGZF shards); evaluated at most once, and only if a line number is asked for."""
        self._file_obj = file_obj
        self._buffer_type = buffer_type
        self._has_header = has_header
        self._f_name = self._file_obj.name if hasattr(self._file_obj, "name") else str(self._file_obj)
        self._header_data = self._buffer_type.read_header(self._file_obj)
        self._buffer_type = self._buffer_type.modify_class_with_header_data(self._header_data)
        self._marker = getattr(self._buffer_type, "_new_entry_marker", None)
        self._is_finished = False
        self._stream_mode = False          # gzip: every batch takes min_chunk_size NEW bytes (parser.py:164-165)
        self._left_over = None             # bytes behind the last complete entry of the previous batch (a staging view)
        self._staging = None
        self._early = None                 # device side of the read-ahead (_EarlyUpload)
        self._ahead_threads = []
        self._window_held = 0              # windowed read_chunks: bytes the reference's reader would hold behind the last chunk
        self.n_bytes_read = 0
        self._n_lines_local = 0            # lines of the chunks this reader has handed out
        self._start, self._stop = 0, None
        self._chunk_modulo = chunk_modulo
        self._lines_before = lines_before
        self._lines_before_value = None
        if byte_range is not None:
            self._start, self._stop = int(byte_range[0]), int(byte_range[1])
            self._file_obj.seek(self._start)

    # -- the part of the file this reader reads (io/sharding.py) ---------------------------------------------------------
    def lines_before_range(self):
        """lines of the file in front of this reader's first byte (0 for a reader of the whole file); counted on first use"""
        if self._lines_before_value is None:
            if self._lines_before is not None:
                self._lines_before_value = int(self._lines_before())
            elif self._start > 0:
                import ctypes as C
                from .._native import lib, check
                found = C.c_int64(0)
                check(lib.bnpk_count_byte_file(self._file_obj.fileno(), 0, self._start, NEWLINE, max(1, _READ_THREADS), C.byref(found)))
                self._lines_before_value = int(found.value)
            else:
                self._lines_before_value = 0
        return self._lines_before_value

    @property
    def n_lines_read(self):
        """lines of the file up to the end of the last chunk handed out — counted from the start of the FILE, also by a reader
        of a part of it (parser.py:141-143: what FormatException.line_number is relative to)"""
        return self._n_lines_local + self.lines_before_range()

    def lines_before_chunk(self, chunk_lines=0):
        """the line number of the first line of the chunk handed out last (``chunk_lines`` = its lines), as a number that is
        only worked out when somebody adds it to something"""
        if self._start == 0 and self._lines_before is None:
            return self._n_lines_local - chunk_lines
        return _LinesBefore(self._n_lines_local - chunk_lines, self)

    def _end_of_data(self):
        """first byte this reader does not read (plain files)"""
        size = os.fstat(self._file_obj.fileno()).st_size
        return size if self._stop is None else min(size, self._stop)

    # -- the reference's surface ---------------------------------------------------------------------------------
    def __enter__(self):
        return self

    def __exit__(self, *args):
        self.close()

    def __iter__(self):
        return self.read_chunks()

    def set_prepend_mode(self):
        self._stream_mode = True

    def close(self):
        for t in list(self._ahead_threads):                  # read-aheads still writing into staging buffers
            t.join()
        self._ahead_threads = []
        self._file_obj.close()
        if self._early is not None:
            self._release_early()
        if self._staging is not None:
            self._staging.release()
            self._staging = None
        self._left_over = None

    def _release_early(self):
        self._early.stream.synchronize()                     # nothing may still be copying into its buffers when they change hands
        # the next reader takes it over — its copy stream and its staging in HBM, as long as that is small: what a reader of
        # gigabyte batches staged (three buffers of a batch each) goes back to the allocator instead of being kept from the
        # counting kernels for the life of the process
        held = sum(b.numel() for b in self._early.buffers if b is not None)
        if held > _EARLY_KEEP_BYTES:
            self._early.buffers = [None] * len(self._early.buffers)
            self._early.free_events = [None] * len(self._early.free_events)
        if len(_early_cache) < 2:
            _early_cache.append(self._early)
        self._early = None

    def read(self):
        """the whole file as one buffer (parser.py:89-94)"""
        if self._chunk_modulo is not None:
            # one buffer cannot hold every n-th chunk of a stream; NpDataclassReader.read() joins the shard's chunks instead
            raise ValueError("read() of a chunk-modulo shard (a gzip stream that cannot be entered in the middle): the part of "
                             "rank %d of %d is a set of chunks — use read_chunks(), or bnp.open(...).read()" % self._chunk_modulo)
        raw = self._file_obj.read() if self._stop is None else self._file_obj.read(max(0, self._stop - self._file_obj.tell()))
        if len(raw) == 0:
            return None
        batch = np.empty(len(raw) + 2, dtype=np.uint8)
        batch[:len(raw)] = np.frombuffer(raw, dtype=np.uint8)
        n = self._terminate(batch, len(raw))
        return self._buffer_type.from_raw_buffer(batch[:n], header_data=self._header_data)

    def read_chunks(self, min_chunk_size=5000000, max_chunk_size=None):
        if self._chunk_modulo is not None:
            r, n = self._chunk_modulo
            for chunk in self._read_chunks(min_chunk_size, max_chunk_size):
                i, self._chunk_index = self._chunk_index, self._chunk_index + 1
                if i % n == r:
                    yield chunk
            return
        yield from self._read_chunks(min_chunk_size, max_chunk_size)

    def _read_chunks(self, min_chunk_size=5000000, max_chunk_size=None):
        if _READ_AHEAD and min_chunk_size >= _BIG and self._plain_file() and not self._stream_mode:
            yield from self._read_chunks_ahead(min_chunk_size, max_chunk_size)
            return
        if (_READ_AHEAD and _WINDOWED and _WINDOW_MIN <= min_chunk_size < _BIG and self._plain_file() and not self._stream_mode
                and self._marker is None and hasattr(self._buffer_type, "n_lines_per_entry")
                and not getattr(get_ops(), "host_only", False)):
            batch = max(min(_WINDOW_BATCH, _WINDOW_CUTS * min_chunk_size), 2 * min_chunk_size)
            yield from self._read_chunks_ahead(batch, max_chunk_size, window=min_chunk_size)
            return
        while not self._is_finished:
            chunk = self._read_chunk(min_chunk_size, max_chunk_size)
            if chunk is None:
                break
            yield chunk

    def _plain_file(self):
        f = self._file_obj
        if not isinstance(f, (io.BufferedReader, io.FileIO)):
            return False
        try:
            f.fileno()
            return f.seekable()
        except (OSError, ValueError):
            return False

    def _read_chunks_ahead(self, min_chunk_size, max_chunk_size, window=None):
        """read_chunks for big batches of a plain file: batch i + 1's new bytes are read (behind _FRONT bytes of room) while
        batch i is uploaded, scanned and used; its left-over is copied in front of them afterwards.
        window: the caller asked for chunks of this (small) size — the batches are a transport, and what is yielded are the
        chunks a reader with windows of ``window`` bytes would cut (``_cut_windows``)."""
        if self._staging is None:
            self._staging = _Staging()
        _FRONT = globals()["_FRONT"] if window is None else (max(globals()["_FRONT"], 2 * window + (1 << 20)) + 4095) & ~4095
        cutter = self._cut_whole if window is None else (lambda b, d: self._cut_windows(b, d, window, max_chunk_size))

        early = None
        if _EARLY_UPLOAD and not getattr(get_ops(), "host_only", False):
            if self._early is None:                          # (a finished reader's staging — stream, two HBM buffers — is kept for the next)
                self._early = _early_cache.pop() if _early_cache else _EarlyUpload()
            early = self._early

        # Two batches are read ahead where the reads can be given their places in the file beforehand (a plain file read by
        # bnpk_pread_parallel): while the caller works on batch i, batch i + 1 is (usually) in HBM already and batch i + 2 is
        # being read and copied — the copy engine is busy without a gap.  One batch ahead otherwise.
        placed = _READ_THREADS >= 2
        depth = min(2, _READ_DEPTH) if placed else 1
        f = self._file_obj
        next_pos = f.tell()
        file_size = self._end_of_data() if placed else 0

        def start():
            # the staging buffer is taken here, by the caller's thread: making sure that no copy out of it is in flight
            # means waiting for the stream, and the background thread would wait behind the upload of the current batch
            nonlocal next_pos
            room = self._staging.room(_FRONT + min_chunk_size + 2)
            slot = early.take(_FRONT + min_chunk_size + 2) if early is not None else None
            box = {}
            if placed:
                pos, want = next_pos, max(0, min(min_chunk_size, file_size - next_pos))
                next_pos += want

            def work():
                try:
                    new = room[_FRONT:_FRONT + (want if placed else min_chunk_size)]
                    target = early.target(slot, _FRONT) if early is not None else None
                    if early is not None:
                        early.begin(slot)
                    if placed:
                        got = self._fill_at(new, pos, target) if want else 0
                        sent = True
                    else:
                        got = self._fill(new, target)
                        sent = got < 0                                       # the reading threads sent the pieces themselves
                        got = abs(got)
                    if early is not None and got > 0:
                        if not sent:                                         # (a serial read: one copy behind it)
                            early.upload(slot, new[:got], _FRONT)
                        from .pinned import owner_of
                        box["uploaded"] = (slot, early.finish(slot, owner_of(room)))
                    box["result"] = (room, got)
                except BaseException as e:                   # noqa: BLE001  (re-raised where the bytes are taken)
                    box["error"] = e
            box["thread"] = threading.Thread(target=work, name="bnpk-read-ahead", daemon=True)
            box["thread"].start()
            self._ahead_threads.append(box["thread"])        # (close() waits for them before the staging buffers go back)
            return box

        def take(box):
            box["thread"].join()
            self._ahead_threads.remove(box["thread"])
            if "error" in box:
                raise box["error"]
            return box["result"]

        held = self._left_over if self._left_over is not None else np.zeros(0, dtype=np.uint8)
        self._left_over = None
        pending = collections.deque([start()])
        completed = False
        try:
            while not self._is_finished:
                ahead = pending.popleft()
                room, got = take(ahead)
                uploaded = ahead.get("uploaded")
                self._is_finished = got < min_chunk_size
                if got == 0 and held.size == 0:
                    break
                # (got == 0 with bytes held: the batch before ended exactly at the end of the file, so what it left over was
                # never terminated — the reference reads that tail again, finds it short, terminates and parses it,
                # parser.py:183-200 — it gets its newline / marker here and is parsed as the last batch)
                if held.size <= _FRONT:
                    room[_FRONT - held.size:_FRONT] = held
                    first, n = _FRONT - held.size, _FRONT + got
                else:                                        # an entry longer than the room in front: gather it anew
                    big = np.empty(held.size + got + 2, dtype=np.uint8)
                    big[:held.size] = held
                    big[held.size:held.size + got] = room[_FRONT:_FRONT + got]
                    room, first, n, uploaded = big, 0, held.size + got, None
                if self._is_finished:
                    n = self._terminate(room, n)
                batch = room[first:n]
                while not self._is_finished and len(pending) < depth:
                    pending.append(start())                  # the next batches' bytes, while this one is parsed and used
                if window is None and max_chunk_size is not None and batch.size > max_chunk_size:
                    raise Exception("No complete entry found")
                on_device = early.assemble(uploaded[0], uploaded[1], room, first, n, _FRONT, got) if uploaded is not None else None
                pieces = cutter(batch, on_device)
                del on_device
                held = batch                                 # (no complete entry: the whole batch is carried over)
                for buff, end in pieces:
                    # what lies behind the piece is the left-over from now on: a caller who stops here finds it again
                    held = batch[end:]
                    self.n_bytes_read += buff.size
                    self._n_lines_local += buff.n_lines
                    yield buff
                if self._is_finished:                        # (a finished file's tail without a complete entry is dropped)
                    held = np.zeros(0, dtype=np.uint8)
            completed = True
        finally:
            # abandoned mid-file (the caller stopped iterating, or an exception went through): what was read stays available —
            # the bytes behind the last yielded entry, and what the read-aheads brought in since, are the left-over of the
            # next read, and the file is NOT finished for the reader even if a read-ahead met its end (that read, of 0 new
            # bytes, terminates and parses the tail: parser.py:183-200)
            while pending:
                room, got = take(pending.popleft())
                if got == 0:                                 # (a read started behind the end of the file: nothing)
                    continue
                rest = np.empty(held.size + got, dtype=np.uint8)
                rest[:held.size] = held
                rest[held.size:] = room[_FRONT:_FRONT + got]
                held = rest
            if placed:
                f.seek(next_pos)                             # (the reads were placed by hand: the file object follows)
            if not completed and held.size:
                self._left_over, self._is_finished = np.array(held), False      # (a copy: the staging buffers go back)
            else:
                self._left_over = None
            if self._is_finished and self._early is not None:    # the file is through: its device staging goes to the next reader
                self._release_early()

    def _cut_whole(self, batch, on_device):
        """one buffer over all complete entries of the batch -> [(buffer, its size)]"""
        buff = self._parse(batch if on_device is None else on_device)
        return [] if buff is None else [(buff, buff.size)]

    def _cut_windows(self, batch, on_device, window, max_chunk_size):
        """the chunks a reader with windows of ``window`` bytes cuts out of the batch (a generator of (buffer, end offset)):
        one scan of the whole batch, one kernel that walks the windows over its newline table, one download.  A chunk is a
        VIEW of the batch (text and newline table), so the fixed cost per chunk is a few host objects."""
        from ..ops import LineScan
        from ..device import HArray
        cls, ops = self._buffer_type, get_ops()
        lpe = cls.n_lines_per_entry
        try:
            big = cls.from_raw_buffer(batch if on_device is None else on_device, header_data=self._header_data)
        except IncompleteEntryException:
            return
        except FormatException:
            # a malformed entry somewhere in the batch: the reference yields the chunks in front of it before it raises, so
            # this batch is cut the slow way, window by window, each scanned on its own
            yield from self._cut_windows_slowly(batch, window, max_chunk_size)
            return
        cuts, rebased = ops.window_cuts(big._data, big._scan, lpe, window, batch.size, self._is_finished, self._window_held,
                                        max_chunk_size, 256)
        text = big._data.dev()
        n_cut = int(cuts[0])
        # what the chunks of this batch share (the sequence column encoded once for all of them) — if every chunk strips
        # carriage returns the way the batch as a whole does (the reference decides that per chunk)
        share = None
        if n_cut and _SHARE and all(bool(cuts[6 + 4 * i]) == big._scan.has_cr for i in range(n_cut)):
            from .buffers import BatchShare
            share = BatchShare(big, [0] + [int(cuts[4 + 4 * i]) for i in range(n_cut)])
        j0 = s = 0
        for i in range(n_cut):
            j1, e, has_cr, w_end = (int(v) for v in cuts[4 + 4 * i:8 + 4 * i])
            scan = LineScan(e - s, (j1 - j0) * lpe, j1 - j0, HArray(dev=rebased[j0 * lpe:j1 * lpe]), bool(has_cr))
            self._window_held = w_end - e
            buff = cls(HArray(dev=text[s:e]), scan)
            if share is not None:
                buff._share = (share, j0, j1)
            yield buff, e
            j0, s = j1, e
        if cuts[1]:                                          # (behind the chunks in front of it, as in the plain loop)
            raise Exception("No complete entry found")

    def _cut_windows_slowly(self, batch, window, max_chunk_size):
        """_cut_windows by the book (parser.py:96-171 on the bytes of the batch): every window scanned on its own, so that
        the chunks in front of a malformed entry are yielded and the exception carries the reference's line number"""
        s, held = 0, self._window_held
        while s < batch.size:
            w_end = s + (held + window if held >= window else window)
            while True:
                if w_end > batch.size:
                    if not self._is_finished:
                        return                               # the window reaches into the next batch
                    w_end = batch.size
                if max_chunk_size is not None and w_end - s > max_chunk_size:
                    raise Exception("No complete entry found")
                buff = self._parse(np.array(batch[s:w_end]))
                if buff is not None or w_end >= batch.size:
                    break
                w_end += window
            if buff is None:
                return
            self._window_held = held = w_end - (s + buff.size)
            s += buff.size
            yield buff, s

    def read_chunk(self, min_chunk_size=5000000, max_chunk_size=None):
        """the next chunk of THIS reader's part of the file: of a chunk-modulo shard (a gzip stream every rank inflates,
        io/sharding.py) the next chunk i with i % n == r — the others are read and dropped, as ``read_chunks`` drops them"""
        if self._chunk_modulo is None:
            return self._read_chunk(min_chunk_size, max_chunk_size)
        r, n = self._chunk_modulo
        while True:
            chunk = self._read_chunk(min_chunk_size, max_chunk_size)
            i, self._chunk_index = self._chunk_index, self._chunk_index + 1
            if chunk is None or i % n == r:
                return chunk

    def _read_chunk(self, min_chunk_size=5000000, max_chunk_size=None):
        """the next buffer of complete entries, or None at the end of the file (parser.py:96-171)"""
        if self._staging is None:
            self._staging = _Staging()
        held = self._left_over if self._left_over is not None else np.zeros(0, dtype=np.uint8)
        self._left_over = None
        # a plain file is read in windows of min_chunk_size bytes from the first unconsumed byte (what the reference's
        # seek-back amounts to); a stream, or a window without a complete entry, takes min_chunk_size more
        want = min_chunk_size if (self._stream_mode or held.size >= min_chunk_size) else min_chunk_size - held.size
        while True:
            batch, n_new = self._extend(held, want)
            if batch.size == 0:
                return None
            if max_chunk_size is not None and batch.size > max_chunk_size:
                raise Exception("No complete entry found")
            buff = self._parse(batch)
            if buff is not None:
                break
            if n_new == 0:
                return None                                  # (as in the reference, an incomplete tail is dropped)
            held, want = batch, min_chunk_size
        if not self._is_finished:
            self._left_over = batch[buff.size:]
        self.n_bytes_read += buff.size
        self._n_lines_local += buff.n_lines
        return buff

    # -- mechanism ---------------------------------------------------------------------------------------------------
    def _terminate(self, array, n):
        """end of file: a final newline if the file lacks one, and the entry marker for multi-line FASTA (parser.py:183-190),
        written in place behind the n bytes of data; returns the new length"""
        if array[n - 1] != NEWLINE:
            array[n] = NEWLINE
            n += 1
        if self._marker is not None:
            array[n] = ord(self._marker)
            n += 1
        return n

    def _extend(self, held, want):
        """held bytes + up to ``want`` new bytes of the file, contiguous in a fresh staging buffer"""
        room = self._staging.room(held.size + want + 2)
        room[:held.size] = held
        got = self._fill(room[held.size:held.size + want])
        self._is_finished = got < want
        n = held.size + got
        # the end of the file — also when the read before ended exactly there and left an unterminated tail behind (the
        # reference reads that tail again, finds it short, terminates and parses it: parser.py:183-200)
        if n and self._is_finished:
            n = self._terminate(room, n)
        return room[:n], got

    def _fill_parallel(self, target, upload=None):
        """a plain file on disk / in the page cache: disjoint slices read by a few threads straight into the (page-locked)
        target (bnpk_pread_parallel: native threads, no interpreter lock between them; with ``upload`` = (device pointer,
        stream) every piece goes on to the device as soon as it is read).  Returns None when this is not a big read of a
        plain seekable file (the serial path)."""
        if _READ_THREADS < 2 or target.size < _BIG or self._stream_mode or not self._plain_file():
            return None
        f = self._file_obj
        fd, pos = f.fileno(), f.tell()
        want = min(target.size, self._end_of_data() - pos)
        if want <= 0:
            return 0
        import ctypes as C
        from .._native import lib, check
        ctx = None
        if upload is not None:
            from ..device import Device
            ctx = Device.get().ctx
        got = C.c_int64(0)
        d_dst, stream = upload if upload is not None else (None, None)
        check(lib.bnpk_pread_parallel(ctx, fd, pos, C.c_void_p(target.__array_interface__["data"][0]), want,
                                      _READ_THREADS, _PIECE, d_dst, stream, C.byref(got)))
        if got.value < want:
            raise OSError("short read of %s" % self._f_name)
        f.seek(pos + want)
        return want

    def _fill_at(self, target, pos, upload=None):
        """target.size bytes of the (plain) file from byte ``pos`` on, whatever the file object's position is: reads that
        run side by side take their places in the file when they are started"""
        import ctypes as C
        from .._native import lib, check
        ctx = None
        if upload is not None:
            from ..device import Device
            ctx = Device.get().ctx
        got = C.c_int64(0)
        d_dst, stream = upload if upload is not None else (None, None)
        check(lib.bnpk_pread_parallel(ctx, self._file_obj.fileno(), pos, C.c_void_p(target.__array_interface__["data"][0]),
                                      target.size, _READ_THREADS, _PIECE, d_dst, stream, C.byref(got)))
        if got.value < target.size:
            raise OSError("short read of %s" % self._f_name)
        return target.size

    def _fill(self, target, upload=None):
        """file.readinto(target) until it is full or the file ends (buffered / gzip readers return short reads).
        upload: (device pointer, stream) — a big read of a plain file sends its pieces there as they arrive and returns
        -got instead of got (a negative count: "these bytes are on their way already")"""
        got = self._fill_parallel(target, upload)
        if got is not None:
            return -got if upload is not None else got
        if self._stop is not None:                           # (a part of a plain file: it ends at ``stop``)
            target = target[:max(0, min(target.size, self._stop - self._file_obj.tell()))]
        if not hasattr(self._file_obj, "readinto"):
            raw = self._file_obj.read(target.size)
            target[:len(raw)] = np.frombuffer(raw, dtype=np.uint8)
            return len(raw)
        view, got = memoryview(target), 0
        while got < target.size:
            n = self._file_obj.readinto(view[got:])
            if not n:
                break
            got += n
        return got

    def _parse(self, batch):
        """the buffer over the complete entries of the batch, or None if it does not hold one yet"""
        try:
            return self._buffer_type.from_raw_buffer(batch, header_data=self._header_data)
        except IncompleteEntryException:
            return None
        except FormatException as e:
            e.line_number += self.n_lines_read               # (counted from the start of the file, also for a part of it)
            raise
