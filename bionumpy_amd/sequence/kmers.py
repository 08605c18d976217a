"""get_kmers / count_kmers on the MI355X path (bionumpy/sequence/kmers.py:36-145).

``get_kmers`` keeps the reference's signature and result type: an ``EncodedRaggedArray`` of int64
hashes tagged ``KmerEncoding(DNAEncoding, k)``; row r holds max(0, L_r - k + 1) hashes with the first
base in the least significant 2 bits.  The hashes are produced by the rolling 2-bit kernel
(``bnpk_kmers``) straight from the packed reads in HBM and stay there until ``.raw()`` is asked for.
"""
import logging

from ..encoded_array import (EncodedArray, EncodedRaggedArray, BaseEncoding, DNAEncoding, AlphabetEncoding,
                             change_encoding, as_encoded_array, packed_words)
from ..encodings.kmer_encodings import KmerEncoding
from ..exceptions import EncodingError
from ..ops import get_ops
from ..streams import streamable
from .count_encoded import count_encoded

logger = logging.getLogger(__name__)


def _as_dna_ragged(sequence):
    """(packed words, in_offsets, lens, n_rows, total) of a DNA-encoded array, all HBM-resident"""
    if isinstance(sequence, EncodedArray):                   # a single sequence == one row
        sequence = EncodedRaggedArray(sequence.ravel(), [sequence.size])
    sequence._compact()
    return packed_words(sequence._data), sequence.offsets(), sequence._lens, len(sequence), sequence.total()


def _rolling(sequence, window, k, kernel):
    """shared skeleton of get_kmers / get_minimizers: trimmed output offsets + one kernel launch"""
    ops = get_ops()
    single = isinstance(sequence, EncodedArray)
    source = None if single else getattr(sequence, "_trim_source", None)
    row_ends = None if single else getattr(sequence, "_row_ends", None)
    if source is not None and hasattr(ops, "windows_counted"):
        # rows of a reader's batch (io/buffers.py: BatchShare): their values are a part of the batch's, computed once
        shared = source[0].windows(source[1], k, window, source[2], source[3])
        if shared is not None:
            n_out, out_off = source[0].trimmed(source[1], window, source[2], source[3])
            return shared, out_off, sequence._lens, len(sequence), n_out, single
    packed, in_off, lens, n_rows, total = _as_dna_ragged(sequence)
    if source is not None and window >= 1:                   # (... or at least their number is known: BatchShare.trimmed)
        n_out, out_off = source[0].trimmed(source[1], window, source[2], source[3])
    elif hasattr(ops, "windows_counted") and ops.windows_counted is not None:
        # the number of windows comes back with the start mask; the trimmed row offsets are only scanned if somebody asks
        counted = ops.windows_counted(packed, in_off, n_rows, k, window, total, row_ends)
        if counted is not None:
            from ..device import LazyHArray
            out_off = LazyHArray(n_rows + 1, lambda: ops.row_offsets(lens, window)[0])
            return counted[0], out_off, lens, n_rows, counted[1], single
        out_off, n_out = ops.row_offsets(lens, window)
    else:
        out_off, n_out = ops.row_offsets(lens, window)
    values = kernel(ops, packed, in_off, out_off, n_rows, n_out, total)
    return values, out_off, lens, n_rows, n_out, single


def _as_four_letter(sequence):
    """the encoding checks get_kmers makes (kmers.py:66-81): base-encoded input becomes DNA, anything else must carry an
    AlphabetEncoding"""
    if sequence.encoding == BaseEncoding:
        try:
            sequence = change_encoding(sequence, DNAEncoding)
        except EncodingError:
            logging.error("Tried to change encoding of sequences to DNAEncoding, but failed. "
                          "Make sure your sequences are valid DNA, only containing A, C, G, and T")
            raise
    assert isinstance(sequence.encoding, AlphabetEncoding), \
        "Sequence needs to be encoded with an AlphabetEncoding, e.g. DNAEncoding. " \
        "Change encoding of your sequences by using e.g. bnp.change_encoding(sequences, bnp.DNAEncoding)"
    return sequence


def _get_kmers_generic(sequence, k):
    """KmerEncoder(k, encoding).rolling_window(sequence) (sequence/kmers.py:17-27,87; rollable.py:29-69): the path of
    alphabets that are not 4 letters wide — hash = codes . alphabet_size ** arange(k), trimmed per row."""
    ops = get_ops()
    single = isinstance(sequence, EncodedArray)
    if single:
        sequence = EncodedRaggedArray(sequence.ravel(), [sequence.size])
    sequence._compact()
    lens, n_rows = sequence._lens, len(sequence)
    out_off, n_out = ops.row_offsets(lens, k)
    hashes = ops.kmers_generic(sequence._data, sequence.offsets(), out_off, n_rows, n_out, k,
                               sequence.encoding.alphabet_size)
    encoding = KmerEncoding(sequence.encoding, k)
    if single:
        return EncodedArray(hashes, encoding)
    return EncodedRaggedArray._from_parts(hashes, None, _trimmed_lens(out_off, lens, k), out_off, n_rows, n_out, encoding)


class KmerEncoder:
    """KmerEncoder (sequence/kmers.py:17-33): the generic rolling k-mer hash of any AlphabetEncoding.  get_kmers uses it
    for alphabets that are not 4 letters wide; for 4 letters it gives the same hashes as the 2-bit path
    (tests/test_kmer.py:27-30)."""

    def __init__(self, k, alphabet_encoding):
        assert isinstance(alphabet_encoding, AlphabetEncoding), alphabet_encoding
        self.window_size = k
        self._k = k
        self._encoding = alphabet_encoding

    def rolling_window(self, sequence):
        return _get_kmers_generic(as_encoded_array(sequence, self._encoding), self._k)

    def __call__(self, sequence):
        """the hash of one window of exactly k letters"""
        sequence = as_encoded_array(sequence, self._encoding)
        assert sequence.size == self._k
        return _get_kmers_generic(sequence, self._k)


def get_kmers(sequence, k, canonical=False):
    """k-mer hashes of every position of every sequence (sequence/kmers.py:36-87).

    canonical=True (extension): every hash is replaced by the smaller of itself and the hash of its
    reverse complement k-mer, so that both strands of a sequence give the same k-mers."""
    assert 0 < k < 32, "k must be larger than 0 and smaller than 32"
    sequence = _as_four_letter(sequence)
    if canonical and "".join(sequence.encoding.get_alphabet()).upper() != "ACGT":
        raise NotImplementedError("canonical k-mers need the ACGT alphabet (complement = 3 - code)")
    if sequence.encoding.alphabet_size != 4:                 # (kmers.py:82-87: only 4-letter alphabets take the 2-bit path)
        return _get_kmers_generic(sequence, k)
    hashes, out_off, lens, n_rows, n_out, single = _rolling(
        sequence, k, k, lambda ops, p, i, o, n, m, t: ops.kmers(p, i, o, n, m, k, total=t))
    if canonical and n_out:
        from ..device import SharedSlice, HArray
        if isinstance(hashes, SharedSlice):                  # (canonical_kmers overwrites its argument)
            hashes = HArray(dev=hashes.dev().clone())
        hashes = get_ops().canonical_kmers(hashes, k)
    encoding = KmerEncoding(sequence.encoding, k)
    if single:
        return EncodedArray(hashes, encoding)
    new_lens = _trimmed_lens(out_off, lens, k)
    return EncodedRaggedArray._from_parts(hashes, None, new_lens, out_off, n_rows, n_out, encoding)


def _trimmed_lens(out_off, lens, window):
    """row lengths after the trim, derived lazily from the offsets"""
    return _LazyLens(out_off) if window > 1 else lens


class _LazyLens:
    """int64 row lengths = diff(offsets); only materialised (on the host) when somebody asks"""

    def __init__(self, offsets):
        self._offsets = offsets
        self._np = None

    @property
    def size(self):
        return self._offsets.size - 1

    def host(self):
        if self._np is None:
            import numpy as np
            self._np = np.diff(self._offsets.host())
        return self._np

    def dev(self):
        off = self._offsets
        if self._np is None and off.on_device:               # the difference of neighbouring offsets, taken where they are
            d = off.dev()
            return d[1:] - d[:-1]
        from ..device import HArray
        return HArray(host=self.host()).dev()

    @property
    def on_device(self):
        return self._offsets.on_device

    @property
    def dtype(self):
        import numpy as np
        return np.dtype(np.int64)


def _pending_weight(n_bases):
    """what uncounted reads weigh toward SparseKmerCounts.PENDING_LIMIT (a bound in 8-byte hashes): their words"""
    return n_bases // 32 + n_bases // 64 + 4


_DENSE_MAX_K = 13            # 4^13 int64 bins = 512 MiB (pipeline.DENSE_MAX_K)


@streamable(sum, coalesce=True)
def count_kmers(sequence, k, axis=None, canonical=False):
    """count every k-mer (sequence/kmers.py:129-145); k <= 8 gives the reference's dense EncodedCounts,
    larger k the sparse (sorted unique keys, counts) extension — see count_encoded.

    For the flattened sparse histogram the hashes are never laid out row by row: they are generated straight
    into the first radix level of the counting sort (bnpk_kmers_partition), as in pipeline.py."""
    assert 0 < k < 32, "k must be larger than 0 and smaller than 32"
    if axis is None and k > 8:
        sequence = _as_four_letter(sequence)                 # (once: on base-encoded input this is the gather + 2-bit encode)
    if axis is None and 8 < k <= _DENSE_MAX_K and not canonical and sequence.encoding.alphabet_size == 4:
        # 4^k bins still fit the device (as in pipeline.py): one dense counting pass and a compaction of the non-zero bins
        # give the same sorted (key, count) pairs as the radix path, which would spend its levels on a 2k <= 26 bit key
        kmers = get_kmers(sequence, k)
        kmers._compact()
        if kmers.total() * 16 >= 4 ** k:                     # (a small input is not worth zeroing 4^k bins)
            from .count_encoded import SparseKmerCounts
            ops = get_ops()
            keys, counts = ops.dense_to_sparse(ops.count_dense(kmers._flat_data(), 4 ** k))
            return SparseKmerCounts(kmers.encoding, keys, counts)
    if axis is None and k > 8 and sequence.encoding.alphabet_size == 4:
        from .count_encoded import SparseKmerCounts
        if canonical and "".join(sequence.encoding.get_alphabet()).upper() != "ACGT":
            raise NotImplementedError("canonical k-mers need the ACGT alphabet (complement = 3 - code)")
        ops = get_ops()
        source = None if isinstance(sequence, EncodedArray) else getattr(sequence, "_trim_source", None)
        if source is not None and not canonical and hasattr(ops, "windows_counted"):
            # rows of a reader's batch (io/buffers.py: BatchShare — the reference's loop at its 5 MB chunks): the chunk's k-mers
            # are a part of the batch's, computed once per batch; the histogram keeps that part and nothing is launched, packed or
            # asked of the device for the chunk (the form below: a row-slice kernel, an offsets scan with its answer, two mask
            # kernels per chunk — 0.19 ms per 5 MB chunk against 0.07 for this one)
            shared = source[0].windows(source[1], k, k, source[2], source[3])
            if shared is not None and 0 < shared.size <= SparseKmerCounts.LAZY_MAX:
                return SparseKmerCounts(KmerEncoding(sequence.encoding, k), pending=[shared], key_bits=2 * k)
        packed, in_off, lens, n_rows, total = _as_dna_ragged(sequence)
        out_off, n_out = ops.row_offsets(lens, k)
        if 0 < n_out <= SparseKmerCounts._reads_limit() // 4 and not canonical:
            # a chunk of a file stream (or anything else of moderate size): nothing is counted yet — the histogram keeps the
            # READS, and counts them together with those of the chunks it is added to as soon as somebody looks at it
            # (see SparseKmerCounts / PendingReads: one pass of the fused generator over all of them)
            from .count_encoded import PendingReads
            mask = ops.kmer_start_mask(in_off, n_rows, total, k)
            return SparseKmerCounts(KmerEncoding(sequence.encoding, k), pending=[PendingReads(packed, mask, total, n_out, k)],
                                    key_bits=2 * k, n_pending=_pending_weight(total))
        if n_out > 0:
            mask = ops.kmer_start_mask(in_off, n_rows, total, k)
            skew = 2.0 if canonical else 1.0          # min(h, rc(h)) has density 2(1 - x) over the key range
            levels = ops.radix_plan(int(n_out * skew), 2 * k)
            bits = levels[0] if levels else 0
            hashes, cuts = ops.kmers_partitioned(packed, mask, total, n_out, k, bits, canonical=canonical)
            del mask
            keys, counts = ops.count_sparse(hashes, key_bits=2 * k, consume=True,
                                            partition=(cuts, bits) if bits else None, skew=skew)
            return SparseKmerCounts(KmerEncoding(sequence.encoding, k), keys, counts)
    kmers = get_kmers(sequence, k, canonical=canonical)
    return count_encoded(kmers, axis=axis)
