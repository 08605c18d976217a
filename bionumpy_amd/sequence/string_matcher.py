"""match_string on the MI355X path (bionumpy/sequence/string_matcher.py:16-55).

A boolean ragged array with one entry per window of ``len(matching_sequence)`` symbols of every row (rows shorter than
the pattern give empty rows), True where the window equals the pattern.  2-bit DNA is matched as a k-mer hash
comparison on the packed form, every other one-to-one encoding byte by byte (``bnpk_match_windows_*``).
"""
import numpy as np

from ..encoded_array import (EncodedArray, EncodedRaggedArray, AlphabetEncoding, as_encoded_array, packed_words,
                             _PackedDna)
from ..ops import get_ops
from ..ragged import RaggedArray


def _match_long(ops, ragged, offsets, n_rows, total, codes, n_out):
    """a pattern of more than 64 symbols (what one launch of the byte kernel compares): its pieces of at most 64 symbols are
    matched on their own and a window matches iff every piece matches at its offset inside the window — the flags of piece j
    for row r start at (windows of piece j before row r) + offset of the piece"""
    m = codes.size
    lens = ragged.lengths
    n_win = np.maximum(lens - (m - 1), 0)                      # windows of the whole pattern per row
    row_of = np.repeat(np.arange(n_rows), n_win)
    within = np.arange(int(n_win.sum())) - np.repeat(np.cumsum(n_win) - n_win, n_win)
    flags = np.ones(int(n_win.sum()), dtype=bool)
    for o in range(0, m, 64):
        piece = codes[o:o + 64]
        _, n_piece = ops.row_offsets(ragged._lens, piece.size)
        hit = ops.match_windows(ragged._flat_data(), offsets, n_rows, total, n_piece, piece, False).host().astype(bool)
        per_row = np.maximum(lens - (piece.size - 1), 0)
        starts = np.cumsum(per_row) - per_row
        flags &= hit[starts[row_of] + within + o]
    assert flags.size == n_out
    return flags


class _DeferredMatches(RaggedArray):
    """match_string's flags over 2-bit DNA, not written yet.  What the reference's callers do with them — ``.any(axis=-1)``,
    ``.sum(axis=-1)``, ``np.sum(flags)`` — is answered by one kernel that counts every row's matching windows from the packed
    words (``bnpk_match_rows_packed``: 1.9 GB read per 50 M reads instead of 7.2 GB of flags written and read back); anything
    else that looks at the flags has them written first, transparently: ``_data`` is a property (ragged._DeferredRows is the
    same idea for a quality column)."""

    @classmethod
    def _defer(cls, words, in_offsets, total_in, codes, lens, offsets, n_rows, total):
        obj = cls.__new__(cls)
        obj._pending = (words, in_offsets, int(total_in), codes)
        obj._real = None
        obj._init(None, None, lens, offsets, n_rows, total)
        return obj

    @property
    def _data(self):
        if self._pending is not None:
            from ..device import as_bool
            words, in_offsets, total_in, codes = self._pending
            self._real = as_bool(get_ops().match_windows(words, in_offsets, self._n_rows, total_in, self._total, codes, True))
            self._pending = None
        return self._real

    @_data.setter
    def _data(self, value):
        self._real = value
        if value is not None:
            self._pending = None

    @property
    def dtype(self):
        return np.dtype(np.bool_)

    def _row_counts(self):
        words, in_offsets, total_in, codes = self._pending
        return get_ops().match_rows(words, in_offsets, self._n_rows, total_in, codes)

    def _row_reduce(self, what, as_float=False):
        if self._pending is None or what != "sum" or as_float:
            return RaggedArray._row_reduce(self, what, as_float)
        from ..device_vector import DeviceVector
        return DeviceVector(self._row_counts())

    def _n_true(self):
        if self._pending is None or not self._n_rows:
            return RaggedArray._n_true(self)
        counts = self._row_counts()
        return int(counts.dev().sum().item()) if counts.on_device else int(counts.host().sum())


def match_string(sequence, matching_sequence):
    sequence = as_encoded_array(sequence)
    pattern = as_encoded_array(matching_sequence, sequence.encoding)
    assert isinstance(pattern, EncodedArray) and pattern.ndim == 1, "the pattern is a single sequence"
    codes = np.asarray(pattern.raw()).astype(np.uint8)
    m = codes.size
    if m == 0:
        raise ValueError("empty matching sequence")
    single = isinstance(sequence, EncodedArray)
    ragged = EncodedRaggedArray(sequence.ravel(), [sequence.size]) if single else sequence
    ragged._compact()
    ops = get_ops()
    n_rows, total, offsets = len(ragged), ragged.total(), ragged.offsets()
    out_off, n_out = ops.row_offsets(ragged._lens, m)
    enc = sequence.encoding
    packed = isinstance(enc, AlphabetEncoding) and enc.alphabet_size == 4 and m <= 31 and \
        (isinstance(ragged._data, _PackedDna) or total >= 4096)
    from ..device import HArray, as_bool
    from .kmers import _LazyLens
    if packed and not single:
        # the flags are written only if somebody looks at them: sums / any / all per row come straight from the 2-bit words
        return _DeferredMatches._defer(packed_words(ragged._data), offsets, total, codes, _LazyLens(out_off) if m > 1 else ragged._lens,
                                       out_off, n_rows, n_out)
    if packed:
        flags = as_bool(ops.match_windows(packed_words(ragged._data), offsets, n_rows, total, n_out, codes, True))
    elif m > 64:
        flags = HArray(host=_match_long(ops, ragged, offsets, n_rows, total, codes, n_out))
    else:
        flags = as_bool(ops.match_windows(ragged._flat_data(), offsets, n_rows, total, n_out, codes, False))
    if single:
        return flags.host()
    # the flags stay where the kernel wrote them (bool over the 0/1 bytes), the row lengths come from the trimmed offsets
    # only if somebody asks: ``match_string(reads, "GATTACA").any(axis=-1)`` is two kernels and no copy (string_matcher.py:16-55
    # of the reference returns a ragged array the caller reduces)
    return RaggedArray._from_parts(flags, None, _LazyLens(out_off) if m > 1 else ragged._lens, out_off, n_rows, n_out)
