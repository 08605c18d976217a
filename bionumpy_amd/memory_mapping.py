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
        data = np.memmap("%s_data.dat" % basename, dtype=np.uint8, mode="r")
        lengths = np.memmap("%s_lengths.dat" % basename, dtype=np.int32, mode="r")
        with open("%s_encoding.pkl" % basename, "rb") as f:
            encoding = pickle.load(f)
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
        data = np.memmap("%s_data.dat" % basename, dtype=np.uint8, mode="w+", shape=max(total, 1))[:total]
        lengths = np.memmap("%s_lengths.dat" % basename, dtype=np.int32, mode="w+", shape=max(n_rows, 1))[:n_rows]
        d0 = r0 = 0
        for sequences in loader_creator():
            flat = np.asarray(sequences.raw().ravel())          # compacted on the device, one download per chunk
            data[d0:d0 + flat.size] = flat
            d0 += flat.size
            lengths[r0:r0 + len(sequences)] = sequences.lengths
            r0 += len(sequences)
        data.flush()
        lengths.flush()
        return cls.load(basename)
