"""get_minimizers on the MI355X path (bionumpy/sequence/minimizers.py:8-54).

For every window of ``window_size`` bases the minimum raw (LSB-first) hash among its
window_size - k + 1 k-mers; one value per window position, no dedup; row r keeps
max(0, L_r - window_size + 1) values.  One kernel (``bnpk_minimizers``) reads the packed reads and
writes the minimizers; the reference's N*w*k intermediate never exists.
"""
from ..encoded_array import EncodedArray, EncodedRaggedArray, AlphabetEncoding
from ..encodings.kmer_encodings import KmerEncoding
from ..ops import get_ops
from .kmers import _rolling, _trimmed_lens


def _get_minimizers_generic(sequence, k, window_size):
    """Minimizers(window_size - k + 1, KmerEncoder(k, encoding)).rolling_window(sequence) for alphabets that are not 4 letters
    wide: the smallest hash codes . alphabet_size ** arange(k) of every window, compared as numpy compares int64"""
    ops = get_ops()
    single = isinstance(sequence, EncodedArray)
    if single:
        sequence = EncodedRaggedArray(sequence.ravel(), [sequence.size])
    sequence._compact()
    lens, n_rows = sequence._lens, len(sequence)
    out_off, n_out = ops.row_offsets(lens, window_size)
    values = ops.minimizers_generic(sequence._data, sequence.offsets(), out_off, n_rows, n_out, k, window_size,
                                    sequence.encoding.alphabet_size)
    encoding = KmerEncoding(sequence.encoding, k)
    if single:
        return EncodedArray(values, encoding)
    return EncodedRaggedArray._from_parts(values, None, _trimmed_lens(out_off, lens, window_size), out_off, n_rows, n_out,
                                          encoding)


def get_minimizers(sequence, k, window_size):
    assert isinstance(sequence.encoding, AlphabetEncoding), \
        "Sequence needs to be encoded with an AlphabetEncoding, e.g. DNAEncoding"
    assert k <= window_size, "kmer size must be smaller than window size"
    assert 0 < k < 32, "k must be larger than 0 and smaller than 32"
    if sequence.encoding.alphabet_size != 4:             # any AlphabetEncoding (minimizers.py:48-52): the generic hashes
        return _get_minimizers_generic(sequence, k, window_size)
    values, out_off, lens, n_rows, n_out, single = _rolling(
        sequence, window_size, k, lambda ops, p, i, o, n, m, t: ops.minimizers(p, i, o, n, m, k, window_size))
    encoding = KmerEncoding(sequence.encoding, k)
    if single:
        return EncodedArray(values, encoding)
    return EncodedRaggedArray._from_parts(values, None, _trimmed_lens(out_off, lens, window_size), out_off, n_rows,
                                          n_out, encoding)


class Minimizers:
    """Minimizers (sequence/minimizers.py:8-17): the rollable form — ``Minimizers(n_kmers, KmerEncoder(k, encoding))`` is the
    minimum over windows of n_kmers consecutive k-mers; ``rolling_window`` gives what ``get_minimizers(sequence, k,
    n_kmers + k - 1)`` gives, a call the minimizer of ONE window of window_size letters (tests/test_minimizers.py:43-46)."""

    def __init__(self, n_kmers, kmer_encoding):
        self._n_kmers = n_kmers
        self._kmer_encoding = kmer_encoding
        self.window_size = n_kmers + kmer_encoding.window_size - 1
        self._encoding = kmer_encoding._encoding

    def rolling_window(self, sequence):
        from ..encoded_array import as_encoded_array
        return get_minimizers(as_encoded_array(sequence, self._encoding), self._kmer_encoding.window_size, self.window_size)

    def __call__(self, sequence):
        from ..encoded_array import as_encoded_array
        sequence = as_encoded_array(sequence, self._encoding)
        assert sequence.size == self.window_size, (sequence.size, self.window_size)
        return self.rolling_window(sequence)

