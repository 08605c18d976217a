"""NumpyFileReader: from a file object to device buffers of complete entries.

What a caller of the reference's reader (bionumpy/io/parser.py:36-206) can observe is kept — ``read`` / ``read_chunk``
/ ``read_chunks`` / iteration, which entries a chunk of ``min_chunk_size`` bytes holds (a plain file is read in
windows of ``min_chunk_size`` bytes that start at the first unconsumed byte, a gzip stream gets ``min_chunk_size``
new bytes on top of what was left over), the newline (and '>' for multi-line FASTA) appended at the end of the file,
``Exception("No complete entry found")`` beyond ``max_chunk_size``, ``n_bytes_read`` / ``n_lines_read`` and the
``FormatException.line_number`` that counts from the start of the file — but the mechanism is built for batches of
gigabytes going to a device, not for 5 MB numpy chunks:

* the bytes are read straight into page-locked staging buffers, two of them: a plain file by ``os.preadv`` of disjoint
  slices from a few threads (one thread copies out of the page cache at ~10 GB/s), anything else by ``readinto``; the
  end-of-file newline / marker is written in place behind the data;
* under ``read_chunks`` with batches of 32 MB and more of a plain file, a background thread reads the next batch's new
  bytes while the current batch is uploaded, scanned and worked on by the caller (they are read before the current
  batch's left-over is known, so such a batch is its left-over plus ``min_chunk_size`` new bytes; smaller batches keep
  the reference's window arithmetic exactly);
* nothing is ever seeked back or re-read, for plain files and gzip streams alike: the bytes behind the last complete
  entry (less than one entry) are carried over, by offset, to the front of the other staging buffer;
* whether a batch holds a complete entry, and where its last one ends, is decided by the device scan of the buffer
  class (``from_raw_buffer``) on the uploaded bytes — the host never looks at them (the reference counts the
  newlines of every chunk with numpy before it parses it).
"""
import io
import os
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from ..exceptions import FormatException, IncompleteEntryException
from ..ops import get_ops

NEWLINE = 10
_BIG = 32 << 20                    # batches from this size on: parallel file reads, read-ahead under read_chunks
_READ_THREADS = int(os.environ.get("BNPK_READ_THREADS", min(16, os.cpu_count() or 1)))    # 1 = the calling thread only
_READ_AHEAD = os.environ.get("BNPK_READ_AHEAD", "1") != "0"
_FRONT = 4 << 20                   # room kept in front of a read-ahead for the bytes left over by the batch before it


class _Staging:
    """where the bytes of a batch are gathered: page-locked buffers on the GPU path (io/pinned.py), plain numpy arrays
    under the host-only test backend.  ``room(n)`` returns a writable uint8 array of at least n bytes that does not
    alias the array returned by the previous call."""

    def __init__(self):
        self._pool = None
        if not getattr(get_ops(), "host_only", False):
            from .pinned import PinnedPool
            self._pool = PinnedPool(2)

    def room(self, n):
        if self._pool is not None:
            return self._pool.acquire(n).array
        return np.empty(n, dtype=np.uint8)

    def release(self):
        if self._pool is not None:
            self._pool.release()


class NumpyFileReader:
    def __init__(self, file_obj, buffer_type, has_header=False):
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
        self._ahead_thread = None
        self.n_bytes_read = 0
        self.n_lines_read = 0

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
        if self._ahead_thread is not None:                   # a read-ahead still writing into a staging buffer
            self._ahead_thread.join()
            self._ahead_thread = None
        self._file_obj.close()
        if self._staging is not None:
            self._staging.release()
            self._staging = None
        self._left_over = None

    def read(self):
        """the whole file as one buffer (parser.py:89-94)"""
        raw = self._file_obj.read()
        if len(raw) == 0:
            return None
        batch = np.empty(len(raw) + 2, dtype=np.uint8)
        batch[:len(raw)] = np.frombuffer(raw, dtype=np.uint8)
        n = self._terminate(batch, len(raw))
        return self._buffer_type.from_raw_buffer(batch[:n], header_data=self._header_data)

    def read_chunks(self, min_chunk_size=5000000, max_chunk_size=None):
        if _READ_AHEAD and min_chunk_size >= _BIG and self._plain_file() and not self._stream_mode:
            yield from self._read_chunks_ahead(min_chunk_size, max_chunk_size)
            return
        while not self._is_finished:
            chunk = self.read_chunk(min_chunk_size, max_chunk_size)
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

    def _read_chunks_ahead(self, min_chunk_size, max_chunk_size):
        """read_chunks for big batches of a plain file: batch i + 1's new bytes are read (behind _FRONT bytes of room) while
        batch i is uploaded, scanned and used; its left-over is copied in front of them afterwards."""
        if self._staging is None:
            self._staging = _Staging()

        def start():
            # the staging buffer is taken here, by the caller's thread: making sure that no copy out of it is in flight
            # means waiting for the stream, and the background thread would wait behind the upload of the current batch
            room = self._staging.room(_FRONT + min_chunk_size + 2)
            box = {}

            def work():
                try:
                    box["result"] = (room, self._fill(room[_FRONT:_FRONT + min_chunk_size]))
                except BaseException as e:                   # noqa: BLE001  (re-raised where the bytes are taken)
                    box["error"] = e
            box["thread"] = threading.Thread(target=work, name="bnpk-read-ahead", daemon=True)
            box["thread"].start()
            self._ahead_thread = box["thread"]               # (close() waits for it before the staging buffers go back)
            return box

        def take(box):
            box["thread"].join()
            self._ahead_thread = None
            if "error" in box:
                raise box["error"]
            return box["result"]

        held = self._left_over if self._left_over is not None else np.zeros(0, dtype=np.uint8)
        self._left_over = None
        ahead = start()
        try:
            while not self._is_finished:
                room, got = take(ahead)
                ahead = None
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
                    room, first, n = big, 0, held.size + got
                if self._is_finished:
                    n = self._terminate(room, n)
                batch = room[first:n]
                if not self._is_finished:
                    ahead = start()                          # the next batch's bytes, while this one is parsed and used
                if max_chunk_size is not None and batch.size > max_chunk_size:
                    raise Exception("No complete entry found")
                buff = self._parse(batch)
                if buff is None:                             # no complete entry yet: the whole batch is carried over
                    held = batch
                    continue
                held = batch[buff.size:] if not self._is_finished else np.zeros(0, dtype=np.uint8)
                self.n_bytes_read += buff.size
                self.n_lines_read += buff.n_lines
                yield buff
        finally:
            if ahead is not None:                            # abandoned mid-file: what was read stays available
                room, got = take(ahead)
                rest = np.empty(held.size + got, dtype=np.uint8)
                rest[:held.size] = held
                rest[held.size:] = room[_FRONT:_FRONT + got]
                held, self._is_finished = rest, got < min_chunk_size
            self._left_over = held if held.size and not self._is_finished else None
            if self._is_finished and held.size:              # (a finished file's tail without a complete entry is dropped)
                self._left_over = None

    def read_chunk(self, min_chunk_size=5000000, max_chunk_size=None):
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
        self.n_lines_read += buff.n_lines
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

    def _fill_parallel(self, target):
        """a plain file on disk / in the page cache: ``os.preadv`` of disjoint slices from a few threads straight into the
        (page-locked) target.  Returns None when this is not a big read of a plain seekable file (the serial path)."""
        if _READ_THREADS < 2 or target.size < _BIG or self._stream_mode or not self._plain_file():
            return None
        f = self._file_obj
        fd, pos = f.fileno(), f.tell()
        want = min(target.size, os.fstat(fd).st_size - pos)
        if want <= 0:
            return 0
        view = memoryview(target)
        n_threads = max(1, min(_READ_THREADS, want // (_BIG // 4)))
        step = -(-want // n_threads)

        def read_slice(i):
            a, b = i * step, min((i + 1) * step, want)
            while a < b:
                n = os.preadv(fd, [view[a:b]], pos + a)
                if n <= 0:
                    raise OSError("short read of %s" % self._f_name)
                a += n
        with ThreadPoolExecutor(max_workers=n_threads) as pool:
            list(pool.map(read_slice, range(n_threads)))
        f.seek(pos + want)
        return want

    def _fill(self, target):
        """file.readinto(target) until it is full or the file ends (buffered / gzip readers return short reads)"""
        got = self._fill_parallel(target)
        if got is not None:
            return got
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
            e.line_number += self.n_lines_read
            raise
