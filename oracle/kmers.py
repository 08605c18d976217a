"""2-bit k-mer hashing, minimizers, counting and k-mer index (test infrastructure — see oracle/__init__.py).

Follows:
  bionumpy/sequence/kmers.py:36-126            get_kmers / convolution / _get_dna_kmers
  npstructures BitArray.pack / sliding_window   (un-vendored, >=0.2.15; semantics restated from
                                                SURVEY.md Appendix A and pinned by
                                                docs_source/topics/kmers.rst:73-78)
  bionumpy/sequence/kmers.py:17-27             KmerEncoder.__call__ (generic dot-product path)
  bionumpy/sequence/rollable.py:29-69          RollableFunction.rolling_window
  bionumpy/sequence/minimizers.py:8-54         Minimizers / get_minimizers
  bionumpy/encodings/kmer_encodings.py:55-74   KmerEncoding.to_string / get_labels
  bionumpy/sequence/count_encoded.py:150-188   count_encoded
  bionumpy/sequence/indexing/kmer_indexing.py:24-55  KmerIndex.create_index / get_indices

k = 31 counting has NO reference implementation (get_labels asserts k <= 8); the
sparse histogram (sorted unique int64 keys + int64 counts == np.unique) is this
project's stated extension.
"""
import numpy as np

from .ragged import row_starts, flat_indices

_ALPHABET = "ACGT"


def pack_2bit(codes):
    """BitArray.pack(codes, bit_stride=2): 32 codes per uint64, code i at bits 2*(i%32) of word i//32.

    One zero pad word is appended so that ``words[i//32 + 1]`` always exists.
    """
    codes = np.asarray(codes, dtype=np.uint8)
    n = codes.size
    n_words = (n + 31) // 32
    padded = np.zeros((n_words + 1) * 32, dtype=np.uint64)
    padded[:n] = codes
    shifts = (2 * np.arange(32, dtype=np.uint64))
    return np.bitwise_or.reduce(padded.reshape(-1, 32) << shifts, axis=-1)


def sliding_window_2bit(words, n, k):
    """BitArray.sliding_window(k) over n packed codes -> uint64[n-k+1].

    h[i] = ((w[i/32] >> 2(i%32)) | (w[i/32+1] << (64 - 2(i%32)))) & (4^k - 1)
    """
    m = n - k + 1
    if m <= 0:
        return np.zeros(0, dtype=np.uint64)
    i = np.arange(m, dtype=np.int64)
    word = i >> 5
    sh = (2 * (i & 31)).astype(np.uint64)
    lo = words[word] >> sh
    hi = np.where(sh == 0, np.uint64(0), words[word + 1] << ((np.uint64(64) - sh) & np.uint64(63)))
    mask = np.uint64((1 << (2 * k)) - 1)
    return (lo | hi) & mask


def kmer_hashes_flat(codes, k):
    """_get_dna_kmers body (sequence/kmers.py:121-126): flat int64 hashes incl. windows straddling rows."""
    assert 0 < k < 32, "k must be larger than 0 and smaller than 32"
    codes = np.asarray(codes, dtype=np.uint8)
    return sliding_window_2bit(pack_2bit(codes), codes.size, k).view(np.int64)


def _trim_rows(flat_values, lengths, window):
    """EncodedRaggedArray(values, shape, safe_mode=False)[..., :-(window-1)] then compacted.

    Rows keep their ORIGINAL starts in the flat value array; row r keeps
    max(0, L_r - window + 1) values (sequence/kmers.py:97-100, rollable.py:56-66).
    """
    lengths = np.asarray(lengths, dtype=np.int64)
    starts = row_starts(lengths)
    new_lengths = np.maximum(lengths - (window - 1), 0) if window > 1 else lengths.copy()
    return flat_values[flat_indices(starts, new_lengths)], new_lengths


def get_kmers(codes, lengths, k):
    """get_kmers for alphabet size 4 (sequence/kmers.py:36-87).  Returns (int64 flat, row lengths)."""
    assert 0 < k < 32, "k must be larger than 0 and smaller than 32"
    return _trim_rows(kmer_hashes_flat(codes, k), lengths, k)


def _generic_flat(codes, k, alphabet_size=4):
    codes = np.asarray(codes)
    if codes.size < k:
        return np.zeros(0, dtype=np.int64)
    windows = np.lib.stride_tricks.sliding_window_view(codes, k)
    return windows.astype(np.int64).dot(alphabet_size ** np.arange(k, dtype=np.int64))


def get_kmers_generic(codes, lengths, k, alphabet_size=4):
    """KmerEncoder(k).rolling_window (sequence/kmers.py:23-27 + rollable.py:29-69)."""
    return _trim_rows(_generic_flat(codes, k, alphabet_size), lengths, k)


def get_minimizers(codes, lengths, k, window_size, alphabet_size=4):
    """get_minimizers (sequence/minimizers.py:20-54) for any AlphabetEncoding (alphabet_size letters).

    For every window of ``window_size`` bases: the minimum raw hash among its
    window_size-k+1 k-mers (Minimizers.__call__, minimizers.py:15-17); one value per
    window position; row r keeps max(0, L_r - window_size + 1) values.
    """
    assert k <= window_size, "kmer size must be smaller than window size"
    codes = np.asarray(codes)
    n_kmers = window_size - k + 1
    hashes = _generic_flat(codes, k, alphabet_size)
    if hashes.size < n_kmers:
        mins = np.zeros(0, dtype=np.int64)
    else:
        mins = np.lib.stride_tricks.sliding_window_view(hashes, n_kmers).min(axis=-1)
    return _trim_rows(mins, lengths, window_size)


def kmer_to_string(kmer, k):
    """KmerEncoding.to_string for alphabet size 4 (kmer_encodings.py:55-70): first base = LSB."""
    kmer = int(kmer)
    return "".join(_ALPHABET[(kmer >> (2 * j)) & 3] for j in range(k))


def kmer_from_string(s):
    """KmerEncoding.encode(str) (kmer_encodings.py:40-44)."""
    lut = {c: i for i, c in enumerate(_ALPHABET)}
    return sum(lut[c.upper()] << (2 * j) for j, c in enumerate(s))


def kmer_labels(k):
    """KmerEncoding.get_labels (kmer_encodings.py:72-74)."""
    assert k <= 8, "Only supported for k <= 5"
    return [kmer_to_string(i, k) for i in range(4 ** k)]


def count_dense(hashes, k):
    """count_encoded(..., axis=None) counts vector (count_encoded.py:166-177): bincount, minlength 4^k."""
    hashes = np.asarray(hashes, dtype=np.int64).ravel()
    n_bins = 4 ** k
    max_size = 1000000
    counts = np.zeros(n_bins, dtype=np.int64)
    for i in range(hashes.size // max_size + 1):
        counts += np.bincount(hashes[i * max_size:(i + 1) * max_size], minlength=n_bins)
    return counts


def count_dense_rows(hashes, lengths, k):
    """count_encoded(..., axis=-1) on a ragged array: per-row bincount (count_encoded.py:178-182)."""
    starts = row_starts(lengths)
    n_bins = 4 ** k
    return np.array([np.bincount(hashes[s:s + l], minlength=n_bins)
                     for s, l in zip(starts, lengths)], dtype=np.int64).reshape(len(lengths), n_bins)


def count_weighted(values, weights, n_bins, axis=-1):
    """count_encoded(values, weights, axis) — the counts array of count_encoded.py:166-187, statement by statement:
    flat values and weights of less than two dimensions: np.bincount(values, weights=weights, minlength) (float64);
    axis == -1 and 2-D weights: one bincount of the flat values per row of the weights, made integer unless the weights are
    floating-point; axis == -1 otherwise: one bincount per row of the values under the same weights."""
    weights = np.asanyarray(weights)
    weights2d = weights.ndim == 2
    values = np.asarray(values)
    if axis is None:
        values = values.ravel()
    if values.ndim == 1 and not weights2d:
        return np.bincount(values, weights=weights, minlength=n_bins)
    assert axis == -1
    if not weights2d:
        return np.array([np.bincount(row, weights=weights, minlength=n_bins) for row in values])
    counts = np.array([np.bincount(values, weights=row, minlength=n_bins) for row in weights])
    if not np.issubdtype(counts.dtype, np.integer) and not np.issubdtype(weights.dtype, np.floating):
        counts = counts.astype(int)
    return counts


def count_sparse(hashes):
    """EXTENSION for k > 8: np.unique(hashes, return_counts=True) -> (sorted int64 keys, int64 counts)."""
    keys, counts = np.unique(np.asarray(hashes, dtype=np.int64).ravel(), return_counts=True)
    return keys, counts.astype(np.int64)


def merge_sparse(parts):
    """Sum of sparse histograms across chunks / shards (the k=31 analogue of EncodedCounts.__add__)."""
    parts = [p for p in parts if p[0].size]
    if not parts:
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    keys = np.concatenate([p[0] for p in parts])
    counts = np.concatenate([p[1] for p in parts])
    ukeys, inv = np.unique(keys, return_inverse=True)
    out = np.zeros(ukeys.size, dtype=np.int64)
    np.add.at(out, inv, counts)
    return ukeys, out


def kmer_index_pairs(codes, lengths, k):
    """the distinct (kmer, row) pairs behind KmerIndex.create_index (kmer_indexing.py:24-47), sorted by kmer then row:
    for every k-mer value the reference stores np.unique of the rows that contain it."""
    hashes, new_lengths = get_kmers(codes, lengths, k)
    rows = np.repeat(np.arange(len(new_lengths), dtype=np.int64), new_lengths)
    order = np.lexsort((rows, hashes))
    h, r = hashes[order], rows[order]
    keep = np.ones(h.size, dtype=bool)
    keep[1:] = (h[1:] != h[:-1]) | (r[1:] != r[:-1])
    return h[keep], r[keep]


def build_kmer_index(codes, lengths, k):
    """KmerIndex.create_index (kmer_indexing.py:24-47): {int kmer: sorted unique row ids containing it}."""
    h, r = kmer_index_pairs(codes, lengths, k)
    ukeys, first = np.unique(h, return_index=True)
    bounds = np.concatenate((first, [h.size]))
    return {int(key): r[bounds[i]:bounds[i + 1]] for i, key in enumerate(ukeys)}


def debruijn_neighbours(kmer_set, kmer, k, forward=True):
    """DeBruijnGraph.forward / backward (bionumpy/sequence/debruin.py:20-37): the int k-mers of ``kmer_set`` that
    follow (precede) ``kmer``, in the order the reference tries them (new base A, C, G, T)."""
    if forward:
        base = kmer >> 2
        candidates = [base + (i << (2 * (k - 1))) for i in range(4)]
    else:
        base = (kmer << 2) & (4 ** k - 1)
        candidates = [base + i for i in range(4)]
    members = set(int(x) for x in kmer_set)
    return [c for c in candidates if c in members]


def colored_debruijn(codes, lengths, k):
    """ColoredDeBruijnGraph.from_sequences (debruin.py:51-58): {int kmer: [row of every occurrence, in row order]}"""
    hashes, new_lengths = get_kmers(codes, lengths, k)
    rows = np.repeat(np.arange(len(new_lengths), dtype=np.int64), new_lengths)
    out = {}
    for h, r in zip(hashes.tolist(), rows.tolist()):
        out.setdefault(h, []).append(r)
    return out


# ---- reverse complement / canonical k-mers -------------------------------------------------------------
_ASCII_COMPLEMENT = np.zeros(128, dtype=np.uint8)                 # bionumpy/sequence/dna.py:10,29-33
for _a, _b in {"A": "T", "G": "C", "C": "G", "T": "A", "N": "N"}.items():
    _ASCII_COMPLEMENT[ord(_a)] = ord(_b)


def reverse_complement(flat, lengths, ascii_bytes=False):
    """get_reverse_complement = complement(sequence)[..., ::-1] (bionumpy/sequence/dna.py:36-65) on the flat data of a
    ragged array: codes of the alphabet "ACGT" are complemented as 3 - code (dna.py:22-27: the alphabet mapped through
    A<->T, C<->G), ASCII bytes through the 128-entry table (dna.py:29-33: every other byte becomes 0)."""
    flat = np.asarray(flat, dtype=np.uint8)
    lengths = np.asarray(lengths, dtype=np.int64)
    comp = _ASCII_COMPLEMENT[flat] if ascii_bytes else (3 - flat).astype(np.uint8)
    starts = np.cumsum(lengths) - lengths
    ends = starts + lengths
    row = np.repeat(np.arange(lengths.size), lengths)
    pos = np.arange(flat.size, dtype=np.int64)
    return comp[starts[row] + ends[row] - 1 - pos]


def reverse_complement_hash(hashes, k):
    """hash (first base = least significant 2 bits) of the reverse complement of the k-mer with hash h"""
    h = np.asarray(hashes, dtype=np.uint64)
    out = np.zeros_like(h)
    for j in range(k):                                            # code j of the k-mer becomes code k-1-j, complemented
        c = (h >> np.uint64(2 * j)) & np.uint64(3)
        out |= (np.uint64(3) - c) << np.uint64(2 * (k - 1 - j))
    return out.astype(np.int64)


def canonical_kmers(hashes, k):
    """min(h, rc(h)): the strand-independent representative of a k-mer (not in the reference)"""
    h = np.asarray(hashes, dtype=np.int64)
    return np.minimum(h, reverse_complement_hash(h, k))


def match_string(flat, lengths, pattern):
    """match_string (bionumpy/sequence/string_matcher.py:16-55): StringMatcher.rolling_window = for every window of
    len(pattern) symbols of the FLAT array np.all(window == pattern), then the ragged trim [..., :-(m-1)]
    (sequence/rollable.py:56-66).  Returns (uint8 0/1 per kept window, new row lengths)."""
    flat = np.asarray(flat)
    pattern = np.asarray(pattern, dtype=flat.dtype)
    m = pattern.size
    lengths = np.asarray(lengths, dtype=np.int64)
    if flat.size < m:
        return np.zeros(0, dtype=np.uint8), np.maximum(lengths - (m - 1), 0)
    windows = np.lib.stride_tricks.sliding_window_view(flat, m)
    hit = np.all(windows == pattern, axis=-1).astype(np.uint8)
    hit = np.concatenate([hit, np.zeros(m - 1, dtype=np.uint8)])     # (flat positions without a full window)
    return _trim_rows(hit, lengths, m)


def pwm_scores(codes, lengths, matrix):
    """get_motif_scores (bionumpy/sequence/position_weight_matrix.py:177-196): PWM.calculate_scores over the FLAT code
    array (:86-104: scores[:n-offset] += matrix[:, offset][codes[offset:]], offset = 0 .. W-1, float64) followed by
    the ragged trim [..., :-(W-1)].  matrix[code][position].  Returns (float64 scores, new row lengths)."""
    codes = np.asarray(codes, dtype=np.int64)
    matrix = np.asarray(matrix, dtype=float)
    scores = np.zeros(codes.size, dtype=float)
    for offset, row in enumerate(matrix.T.copy()):
        scores[:scores.size - offset] += row[codes[offset:]]
    return _trim_rows(scores, lengths, matrix.shape[1])
