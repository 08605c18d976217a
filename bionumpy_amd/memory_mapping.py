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
        """two passes over ``loader_creator()`` (sizes, then data), as the reference (memory_mapping.py:33-90)"""
        warnings.warn("%s is in an experimental stage and may change in the future." % cls.__name__,
                      category=FutureWarning, stacklevel=2)
        total, n_rows, encoding = 0, 0, None
        for sequences in loader_creator():
            n_rows += len(sequences)
            total += sequences.size
            if encoding is None:
                encoding = sequences.encoding
            else:
                assert encoding == sequences.encoding, "Expected %s but got %s" % (encoding, sequences.encoding)
        with open("%s_encoding.pkl" % basename, "wb") as f:
            pickle.dump(encoding, f)
        def created(path, dtype, n):                   # an empty data set is an empty file, not a phantom element
            if n:
                return np.memmap(path, dtype=dtype, mode="w+", shape=n)
            open(path, "wb").close()
            return np.zeros(0, dtype=dtype)
        data = created("%s_data.dat" % basename, np.uint8, total)
        lengths = created("%s_lengths.dat" % basename, np.int32, n_rows)
        d0 = r0 = 0
        for sequences in loader_creator():
            flat = np.asarray(sequences.raw().ravel())          # compacted on the device, one download per chunk
            data[d0:d0 + flat.size] = flat
            d0 += flat.size
            assert int(np.max(sequences.lengths, initial=0)) < 2 ** 31, "row longer than the int32 lengths file can say"
            lengths[r0:r0 + len(sequences)] = sequences.lengths
            r0 += len(sequences)
        if total:
            data.flush()
        if n_rows:
            lengths.flush()
        return cls.load(basename)
