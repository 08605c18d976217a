"""On-disk cache of encoded reads (bionumpy/streams/memory_mapping.py:10-90).

Same three files as the reference — ``<basename>_data.dat`` (uint8, one code per symbol), ``<basename>_lengths.dat``
(int32 row lengths), ``<basename>_encoding.pkl`` — so that a data set decoded once can be loaded without touching the
text again.  ``load`` maps the files and hands them to the device in one upload; 2-bit DNA is packed there
(``bnpk_pack_codes``) the first time it is needed.  The pickled encoding is this package's encoding object.
"""
import pickle
import warnings

import numpy as np

from .device import HArray
from .encoded_array import EncodedArray, EncodedRaggedArray


class MemMapEncodedRaggedArray:

    @classmethod
    def load(cls, basename):
        """read-only memory maps -> EncodedRaggedArray (memory_mapping.py:12-31)"""
        import os
        with open("%s_encoding.pkl" % basename, "rb") as f:
            encoding = pickle.load(f)

        def mapped(path, dtype):                       # (np.memmap cannot map an empty file)
            return np.memmap(path, dtype=dtype, mode="r") if os.path.getsize(path) else np.zeros(0, dtype=dtype)
        data = mapped("%s_data.dat" % basename, np.uint8)
        lengths = mapped("%s_lengths.dat" % basename, np.int32)
        assert int(np.sum(lengths, dtype=np.int64)) == data.size, "lengths do not add up to the data file"
        return EncodedRaggedArray(EncodedArray(HArray(host=np.ascontiguousarray(data)), encoding),
                                  np.asarray(lengths, dtype=np.int64))

    @classmethod
    def create(cls, loader_creator, basename):
        """the three files of the reference's cache (memory_mapping.py:33-90) from the chunks of ``loader_creator()``.

        The reference walks the loader twice (sizes, then data) because it writes through an ``np.memmap`` that has to be
        sized first; here the data file is written front to back as the chunks come — ONE pass over the source — and the lengths
        at the end: same files, half the decoding.  A chunk's codes go from HBM to a page-locked buffer (``bnpk_copy_d2h_async``,
        64 MB pieces) and from there to the file by several threads at once (``os.pwrite`` releases the GIL): round 5 assigned
        into a write-mapped file, page fault by page fault — 0.55 s per GB, twenty times the decode that fed it."""
        import os
        from concurrent.futures import ThreadPoolExecutor
        warnings.warn("%s is in an experimental stage and may change in the future." % cls.__name__,
                      category=FutureWarning, stacklevel=2)
        encoding, all_lengths, at = None, [], 0
        piece, n_threads = 64 << 20, 8
        pinned = None
        fd = os.open("%s_data.dat" % basename, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        try:
            with ThreadPoolExecutor(n_threads) as pool:
                def write_out(view, offset):
                    """``view`` (uint8, contiguous) to the file at ``offset``, cut into one slice per thread"""
                    n = view.size
                    step = max(1 << 20, -(-n // n_threads))
                    jobs = [pool.submit(_pwrite_all, fd, view[o:o + step], offset + o) for o in range(0, n, step)]
                    for j in jobs:
                        j.result()
                for sequences in loader_creator():
                    if encoding is None:
                        encoding = sequences.encoding
                    else:
                        assert encoding == sequences.encoding, "Expected %s but got %s" % (encoding, sequences.encoding)
                    assert int(np.max(sequences.lengths, initial=0)) < 2 ** 31, "row longer than the int32 lengths file can say"
                    all_lengths.append(np.asarray(sequences.lengths, dtype=np.int32))
                    sequences._compact()
                    flat = sequences._flat_data()
                    flat = flat._unpacked() if hasattr(flat, "_unpacked") else flat
                    total = int(sequences.total())
                    if getattr(flat, "on_device", False) and total:
                        from ._native import lib, check
                        from .device import Device, ptr
                        from .io.pinned import PinnedBuffer
                        if pinned is None:
                            pinned = PinnedBuffer(piece)
                        t = flat.dev()
                        stream = Device.get().stream()
                        for o in range(0, total, piece):
                            m = min(piece, total - o)
                            check(lib.bnpk_copy_d2h_async(pinned.ptr, ptr(t[o:o + m]), m, stream))
                            check(lib.bnpk_stream_sync(stream))
                            write_out(pinned.array[:m], at + o)
                    elif total:
                        write_out(np.ascontiguousarray(flat.host()[:total]).view(np.uint8), at)
                    at += total
        finally:
            os.close(fd)
            if pinned is not None:
                pinned.free()
        with open("%s_encoding.pkl" % basename, "wb") as f:
            pickle.dump(encoding, f)
        with open("%s_lengths.dat" % basename, "wb") as f:
            if all_lengths:
                np.concatenate(all_lengths).astype(np.int32).tofile(f)
        return cls.load(basename)


def _pwrite_all(fd, view, offset):
    import os
    mv = memoryview(view)
    done = 0
    while done < len(mv):
        done += os.pwrite(fd, mv[done:], offset + done)
