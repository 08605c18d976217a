"""count_encoded / EncodedCounts (bionumpy/sequence/count_encoded.py:11-188) on the MI355X path.

Dense counts (finite alphabets: DNA letters, k-mers with k <= 8) are an LDS / global-atomic histogram
(``bnpk_count_dense``) and come back as the reference's ``EncodedCounts(alphabet, counts)``.
k > 8 has no reference implementation (``KmerEncoding.get_labels`` asserts k <= 8); for it
``count_encoded(..., axis=None)`` returns ``SparseKmerCounts`` = np.unique(hashes, return_counts=True)
computed by radix sort + run-length on the device — the stated extension of SURVEY.md §3.5.
"""
from numbers import Number

import numpy as np

from ..device import HArray
from ..encoded_array import EncodedArray, EncodedRaggedArray
from ..ops import get_ops


class EncodedCounts:
    """count_encoded.py:11-147"""

    def __init__(self, alphabet, counts, row_names=None):
        self.counts = counts
        self.alphabet = alphabet
        self.row_names = row_names

    def __str__(self):
        return "\n".join("%s: %s" % (c, n) for c, n in zip(self.alphabet, self.counts.T))

    def __repr__(self):
        return "EncodedCounts(alphabet=%r, counts=%r, row_names=%r)" % (self.alphabet, self.counts, self.row_names)

    def __eq__(self, other):
        if self.alphabet != other.alphabet:
            return False
        return bool(np.all(self.counts == other.counts))

    def __getitem__(self, idx):
        return self.counts[..., self.alphabet.index(idx)]

    def _other_counts(self, other):
        if isinstance(other, Number):
            return other
        assert self.alphabet == other.alphabet
        return other.counts

    def __add__(self, other):
        return self.__class__(self.alphabet, self.counts + self._other_counts(other))

    __radd__ = __add__

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if method != "__call__":
            return NotImplemented
        assert all(i.alphabet == self.alphabet for i in inputs if isinstance(i, EncodedCounts))
        arrays = [i.counts if isinstance(i, EncodedCounts) else i for i in inputs]
        kwargs = {k: v.counts if isinstance(v, EncodedCounts) else v for k, v in kwargs.items()}
        return self.__class__(self.alphabet, ufunc(*arrays, **kwargs))

    def _merge_descriptor(self):
        """what a rank that read nothing needs to join the merge with an empty result of this kind (streams._merged_over_ranks)"""
        return (self.__class__, list(self.alphabet), tuple(np.shape(self.counts)), str(np.asarray(self.counts).dtype))

    @classmethod
    def _empty_like_descriptor(cls, d):
        return cls(d[1], np.zeros(d[2], dtype=d[3]))

    def _merged_over_ranks(self, shard):
        """the sum of every rank's counts, on every rank (``+`` across the GPUs of a job whose ranks each read a part of the
        file: SURVEY §8e — one all-reduce of the bins over RCCL, bnpk_allreduce_hist).  Without a process group (a shard
        given by hand) the counts stay this rank's own."""
        from .. import parallel
        if not parallel.group_is_up():
            return self
        counts = np.ascontiguousarray(self.counts, dtype=np.int64)
        total = parallel.allreduce_dense(HArray(host=counts.reshape(-1).copy()), shard.group).host().reshape(counts.shape)
        return self.__class__(self.alphabet, total.astype(self.counts.dtype, copy=False), self.row_names)

    @property
    def proportions(self):
        s = self.counts.sum(axis=-1, keepdims=True)
        return np.where(s > 0, self.counts / np.where(s > 0, s, 1), 0)

    def get_count_for_label(self, label):
        return sum(self.counts[..., self.alphabet.index(l)] for l in label)

    @property
    def labels(self):
        return self.alphabet

    @classmethod
    def vstack(cls, counts):
        alphabet = counts[0].alphabet
        assert all(c.alphabet == alphabet for c in counts)
        ret = cls(alphabet, np.array([c.counts for c in counts], dtype="int"))
        if counts[0].row_names is not None:
            ret.row_names = [c.row_names for c in counts]
        return ret

    def most_common(self, n=None):
        args = np.argsort(self.counts)[::-1]
        if n is not None:
            args = args[:n]
        return self.__class__([self.alphabet[i] for i in args], self.counts[args])

    def as_dict(self):
        return dict(zip(self.alphabet, self.counts.T))


class SparseKmerCounts:
    """Histogram of k-mers for k > 8: sorted distinct int64 keys + int64 counts, HBM-resident.

    ``+`` merges two histograms (the k = 31 analogue of EncodedCounts.__add__, so that
    ``sum(count_kmers(chunk.sequence, 31) for chunk in reader)`` works like the reference's streams).

    The counting is LAZY for small inputs: a histogram may hold k-mer hashes that have not been counted yet (``pending``),
    and adding two histograms only joins those lists.  The reference's loops call count_encoded / count_kmers once per
    5 MB chunk and add the results up (scripts/kmer_counting_example.py:4-17, streams/decorators.py:78-108); the sum of the
    chunks' histograms is the histogram of all their k-mers, so the hashes of many chunks are counted TOGETHER — one
    partition + finishing pass per ~half a billion k-mers instead of one per chunk (a dozen launches and five host round
    trips each, and a merge of the running total with every chunk's histogram, which is quadratic in the number of
    chunks) — as soon as somebody looks at the keys or counts, or when PENDING_LIMIT hashes have piled up."""

    PENDING_LIMIT = 1 << 29            # uncounted hashes a histogram holds at most (4 GiB)
    LAZY_MAX = 1 << 26                 # inputs up to this many hashes are counted lazily (larger ones: at once)
    READS_LIMIT = 3 << 30              # k-mers of uncounted READS a histogram holds at most: what one counting pass takes
                                       # (8 B/k-mer partitioned twice + 16 B per distinct k-mer: ~130 GB of the 288 of an
                                       # MI355X; _reads_limit() scales it to the HBM of the device that is there)
    _reads_limit_value = None

    @classmethod
    def _reads_limit(cls):
        """READS_LIMIT for the device the process runs on: a counting pass takes ~44 bytes per pending k-mer; of a device
        with less HBM than 288 GB only the same share is claimed (the constant was hard-wired to one chip)"""
        if cls._reads_limit_value is None:
            limit = cls.READS_LIMIT
            try:
                total = int(get_ops().device_memory_bytes())
                limit = max(1 << 24, min(limit, int(limit * total / (288 << 30))))
            except Exception:                              # noqa: BLE001  (host-logic backend: no device to ask)
                pass
            cls._reads_limit_value = limit
        return cls._reads_limit_value

    def __init__(self, encoding, keys=None, counts=None, pending=None, key_bits=None, n_pending=None, key_range=None):
        self.encoding = encoding
        self.key_range = key_range         # (lo, hi): this object holds the keys of [lo, hi) only — one rank's part of a histogram
                                           # that is partitioned by key range over the ranks of a job (_merged_over_ranks)
        if key_bits is None:               # what the encoding's hashes need (2k for the 2-bit alphabets), not the widest key
            key_bits = _key_bits_of(encoding)
        as_h = lambda x: x if isinstance(x, HArray) else HArray(host=np.asarray(x, dtype=np.int64))
        self._k = None if keys is None else as_h(keys)
        self._c = None if counts is None else as_h(counts)
        # HArrays of int64 hashes, or PendingReads (the reads themselves: their hashes are generated when they are counted) —
        # shared between histograms, never written to
        self._pending = list(pending or [])
        self._n_pend = sum(p.size for p in self._pending) if n_pending is None else n_pending
        self._n_read_kmers = None          # k-mers of the PendingReads among them (summed once, then carried along by __add__)
        self._key_bits = key_bits

    def _n_pending(self):
        return self._n_pend

    def _pending_read_kmers(self):
        if self._n_read_kmers is None:
            self._n_read_kmers = sum(p.size for p in self._pending if isinstance(p, PendingReads))
        return self._n_read_kmers

    def _force(self):
        """count what is pending and merge it with what has been counted"""
        if self._pending:
            ops = get_ops()
            reads = [p for p in self._pending if isinstance(p, PendingReads)]
            hashes = [p for p in self._pending if not isinstance(p, PendingReads)]
            self._pending, self._n_pend, self._n_read_kmers = [], 0, 0
            done = [(self._k, self._c)] if self._k is not None and self._k.size else []
            if reads:
                done.append(_count_reads(ops, reads))
            del reads
            if len(hashes) == 1:
                done.append(ops.count_sparse(hashes[0], key_bits=self._key_bits))
            elif hashes:                                   # (the joined array is this call's own: the counting may use it up)
                done.append(ops.count_sparse(ops.concat(hashes), key_bits=self._key_bits, consume=True))
            while len(done) > 1:
                (k1, c1), (k2, c2) = done.pop(), done.pop()
                done.append(ops.merge_add(k2, c2, k1, c1) if k1.size and k2.size else ((k1, c1) if k1.size else (k2, c2)))
            self._k, self._c = done[0]
        elif self._k is None:
            self._k = self._c = HArray(host=np.zeros(0, dtype=np.int64))
        return self

    @property
    def _keys(self):
        return self._force()._k

    @property
    def _counts(self):
        return self._force()._c

    @property
    def keys(self):
        return self._keys.host()

    @property
    def counts(self):
        return self._counts.host()

    def __len__(self):
        return self._keys.size

    def __getitem__(self, kmer):
        if isinstance(kmer, str):
            kmer = int(self.encoding.encode(kmer).raw())
        keys = self.keys
        i = int(np.searchsorted(keys, kmer))
        return int(self.counts[i]) if i < keys.size and keys[i] == kmer else 0

    def __add__(self, other):
        if isinstance(other, Number):
            assert other == 0, "only 0 (the start value of sum) can be added to a sparse histogram"
            return self
        assert self.encoding == other.encoding
        if self._pending or other._pending:
            # nothing is counted here: the uncounted hashes of both go on one list (counted parts are merged now — both
            # key lists are sorted and distinct: one merge along the merge path, bnpk_merge_add)
            if self._k is not None and other._k is not None and self._k.size and other._k.size:
                k, c = get_ops().merge_add(self._k, self._c, other._k, other._c)
            else:
                k, c = (self._k, self._c) if (self._k is not None and self._k.size) else (other._k, other._c)
            # (the width of the keys comes from the operands that still have hashes to count)
            bits = max([x._key_bits for x in (self, other) if x._pending])
            out = SparseKmerCounts(self.encoding, k, c, self._pending + other._pending, bits, self._n_pend + other._n_pend)
            out._n_read_kmers = self._pending_read_kmers() + other._pending_read_kmers()
            if out._n_pending() >= self.PENDING_LIMIT or out._n_read_kmers >= self._reads_limit():
                out._force()
            return out
        keys, counts = get_ops().merge_add(self._keys, self._counts, other._keys, other._counts)
        return SparseKmerCounts(self.encoding, keys, counts)

    __radd__ = __add__

    def _merge_descriptor(self):
        return (self.__class__, self.encoding, self._key_bits)

    @classmethod
    def _empty_like_descriptor(cls, d):
        return cls(d[1], key_bits=d[2])

    def _merged_over_ranks(self, shard):
        """this rank's part of the histogram of ALL ranks' k-mers: the 2k-bit key space is cut into one range per rank and
        rank r returns the sorted distinct keys of range r with their counts summed over the job (``key_range`` says which) —
        the ranks' results, one behind the other, are the histogram of the whole file (the result stays
        range-partitioned; ``gathered()`` puts it together where somebody needs it in one place).  What crosses the links is
        what this rank counted (16 bytes per locally distinct key: parallel.exchange_counted), summed on arrival by a tree
        of merges.  Without a process group (a shard given by hand) nothing is exchanged."""
        from .. import parallel
        if not parallel.group_is_up():
            return self
        self._force()
        coll = parallel.collectives(shard.group)
        keys, counts = parallel.exchange_counted(self._k, self._c, self._key_bits, shard.group)
        return SparseKmerCounts(self.encoding, keys, counts, key_bits=self._key_bits,
                                key_range=parallel.key_range_of(coll.rank, coll.world, self._key_bits))

    def gathered(self, group=None):
        """the whole histogram on every rank, from the ranks' parts (an all-gather of the (key, count) runs in rank order:
        the ranges ascend with the ranks, so the concatenation is sorted)"""
        from .. import parallel
        if self.key_range is None or not parallel.group_is_up():
            return self
        keys, counts = parallel.allgather_runs(self._keys, self._counts, group)
        return SparseKmerCounts(self.encoding, keys, counts, key_bits=self._key_bits)

    def __eq__(self, other):
        return self.encoding == other.encoding and np.array_equal(self.keys, other.keys) \
            and np.array_equal(self.counts, other.counts)

    def most_common(self, n=None):
        args = np.argsort(self.counts, kind="stable")[::-1]
        if n is not None:
            args = args[:n]
        return SparseKmerCounts(self.encoding, self.keys[args], self.counts[args])

    def as_dict(self):
        return {self.encoding.to_string(k): int(c) for k, c in zip(self.keys, self.counts)}

    def __repr__(self):
        return "SparseKmerCounts(%s, %d distinct)" % (self.encoding, len(self))


class PendingReads:
    """2-bit reads whose k-mers a histogram has not counted yet: what ``count_kmers(chunk.sequence, k)`` of a file stream leaves
    behind per chunk (0.4 bytes per base against 8 per k-mer hash).  ``size`` = its k-mers; toward PENDING_LIMIT — a memory
    bound in hashes — it weighs its words."""

    def __init__(self, packed, start_mask, n_bases, n_kmers, k):
        self.packed, self.start_mask, self.n_bases, self.size, self.k = packed, start_mask, int(n_bases), int(n_kmers), int(k)


def _count_reads(ops, reads):
    """(keys, counts) of the k-mers of several chunks' reads, counted TOGETHER the way a resident batch is (pipeline.py): the
    chunks' packed words and start masks are put one behind the other — every chunk cut at a whole mask word = 64 bases; the
    bases between a chunk's last one and the cut start no k-mer, which is all the generator asks of them — the hashes are
    generated into the first radix level (bnpk_kmers_partition) and never laid out read by read."""
    k = reads[0].k
    assert all(r.k == k for r in reads)
    if len(reads) == 1:
        packed, mask, n_bases = reads[0].packed, reads[0].start_mask, reads[0].n_bases
    else:
        blocks = [-(-r.n_bases // 64) for r in reads]
        packed = ops.concat_words([(r.packed, 2 * b) for r, b in zip(reads, blocks)])
        mask = ops.concat_words([(r.start_mask, b) for r, b in zip(reads, blocks)])
        n_bases = 64 * sum(blocks)
    n_kmers = sum(r.size for r in reads)
    levels = ops.radix_plan(n_kmers, 2 * k)
    bits = levels[0] if levels else 0
    hashes, cuts = ops.kmers_partitioned(packed, mask, n_bases, n_kmers, k, bits)
    del packed, mask
    return ops.count_sparse(hashes, key_bits=2 * k, consume=True, partition=(cuts, bits) if bits else None)


def _key_bits_of(encoding):
    """bits a hash of ``encoding`` (a KmerEncoding) occupies: 2k for 4-letter alphabets; 62 where it cannot be told"""
    k = getattr(encoding, "k", None)
    letters = getattr(getattr(encoding, "_alphabet_encoding", None), "alphabet_size", None)
    if k is None or letters is None:
        return 62
    return min(62, 2 * k if letters == 4 else max(1, (letters ** k - 1).bit_length()))


def count_encoded(values, weights=None, axis=-1):
    """Count the occurrences of encoded entries (count_encoded.py:150-188).

    axis=None: flattened counts; axis=-1 on a ragged array: one histogram per row."""
    if weights is not None:
        return _count_weighted(values, weights, axis)
    ops = get_ops()
    encoding = values.encoding
    flat_request = axis is None or (isinstance(values, EncodedArray) and values.ndim == 1)
    k = getattr(encoding, "k", None)
    if flat_request and k is not None and k > 8:
        store = _flat_store(values)
        n_letters = encoding._alphabet_encoding.alphabet_size
        key_bits = 2 * k if n_letters == 4 else (n_letters ** k - 1).bit_length()
        if key_bits > 62:
            raise NotImplementedError("sparse counts need k-mer hashes below 2^62 (%d letters, k = %d)" % (n_letters, k))
        if 0 < store.size <= SparseKmerCounts.LAZY_MAX and store.on_device:
            return SparseKmerCounts(encoding, pending=[store], key_bits=key_bits)    # counted when somebody looks (see the class)
        keys, counts = ops.count_sparse(store, key_bits=key_bits)
        return SparseKmerCounts(encoding, keys, counts)
    alphabet = encoding.get_alphabet() if hasattr(encoding, "get_alphabet") else encoding.get_labels()
    n_bins = len(alphabet)
    if flat_request:
        return EncodedCounts(alphabet, _flat_histogram(values, n_bins).host().copy())
    assert axis == -1 and isinstance(values, EncodedRaggedArray)
    values._compact()
    store = values._flat_data()
    store = store._unpacked() if hasattr(store, "_unpacked") else store
    if store.dtype == np.uint8 and n_bins <= ops.COUNT_BYTES_ROWS_MAX_BINS:       # letters: counted where they lie, as bytes
        hist = ops.count_bytes_rows(store, values.offsets(), len(values), values.total(), n_bins)
    else:
        hist = ops.count_dense_rows(_as_int64(store), values.offsets(), len(values), n_bins)
    return EncodedCounts(alphabet, hist.host().reshape(len(values), n_bins).copy())


def _flat_histogram(values, n_bins):
    """the counts of all codes of an encoded (ragged) array, on the device: k-mer hashes (int64) through the dense histogram,
    LETTERS as they are stored — uint8 codes as bytes (bnpk_count_bytes), 2-bit packed DNA by popcounts over its words
    (bnpk_count_packed2) — no host copy, no widening (count_encoded.py:166-176 of the reference calls np.bincount on them)"""
    ops = get_ops()
    if isinstance(values, EncodedRaggedArray):
        values._compact()
        store, n = values._data, values.total()
    else:
        store, n = values._harray(), values.size
    if hasattr(store, "_unpacked"):                      # packed DNA (encoded_array._PackedDna)
        if n_bins == 4 and store._codes is None:
            return ops.count_packed(store.packed, n)
        store = store._unpacked()
    if store.dtype == np.uint8 and n_bins <= 256:
        return ops.count_bytes(store, n_bins)
    return ops.count_dense(_as_int64(store), n_bins)


def _count_weighted(values, weights, axis):
    """count_encoded(values, weights, axis) — bionumpy/sequence/count_encoded.py:166-187: np.bincount(values, weights=weights)
    over flat values (float64 counts, as np.bincount returns them), per row of a matrix of values under 1-D weights
    ("for row in values"), or per row of 2-D weights over flat values ("for row in weights"; integer counts unless the weights
    are floating-point).  The histograms are accumulated on the device (bnpk_count_weighted): integer and bool weights in
    int64 — exact — floating-point ones in float64 with atomic adds."""
    ops = get_ops()
    w = np.asanyarray(weights)
    if w.dtype == np.bool_ or np.issubdtype(w.dtype, np.integer):
        on_device = w.astype(np.int64)
    elif np.issubdtype(w.dtype, np.floating):
        on_device = w.astype(np.float64)
    else:
        raise TypeError("count_encoded: weights of dtype %s" % w.dtype)
    if axis is None:
        values = values.ravel()
    encoding = values.encoding
    alphabet = encoding.get_alphabet() if hasattr(encoding, "get_alphabet") else encoding.get_labels()
    n_bins = len(alphabet)
    flat = isinstance(values, EncodedArray) and values.ndim == 1
    if flat and w.ndim <= 1:
        n, n_rows, vstride, wstride = len(values), 1, 0, 0
        if w.size != n:
            raise ValueError("The weights and list don't have the same length.")
    elif axis == -1 and w.ndim == 2:
        assert flat, "2-D weights count flat values once per row of the weights"
        n, n_rows, vstride, wstride = len(values), w.shape[0], 0, w.shape[1]
        if w.shape[1] != n:
            raise ValueError("The weights and list don't have the same length.")
    elif axis == -1:
        lens = np.asarray(values.shape[1]) if isinstance(values, EncodedRaggedArray) else np.full(values.shape[0], values.shape[1])
        if np.any(lens != w.size):
            raise ValueError("The weights and list don't have the same length.")
        n, n_rows, vstride, wstride = w.size, len(values), w.size, 0
    else:
        raise ValueError("count_encoded: axis %r with these values and weights" % (axis,))
    store = _as_int64(_flat_store(values))
    hist = np.zeros((n_rows, n_bins), dtype=on_device.dtype)
    step = 32768                                             # rows per launch
    for r0 in range(0, n_rows, step):
        r1 = min(n_rows, r0 + step)
        v_part = store if vstride == 0 else HArray(host=store.host()[r0 * n:r1 * n]) if not store.on_device else \
            HArray(dev=store.dev()[r0 * n:r1 * n])
        w_part = HArray(host=np.ascontiguousarray(on_device.reshape(-1) if wstride == 0 else on_device[r0:r1].reshape(-1)))
        hist[r0:r1] = ops.count_weighted(v_part, w_part, n, r1 - r0, vstride, wstride, n_bins).host().reshape(r1 - r0, n_bins)
    if w.ndim == 2:
        counts = hist if np.issubdtype(w.dtype, np.floating) else hist.astype(int)
    else:
        counts = hist.astype(np.float64)                     # np.bincount(..., weights=...) returns float64
        if flat:
            counts = counts[0]
    return EncodedCounts(alphabet, counts)


def _flat_store(values):
    if isinstance(values, EncodedRaggedArray):
        values._compact()
        return values._flat_data()
    return values._harray()


def _as_int64(store):
    if store.dtype == np.int64:
        return store
    return HArray(host=store.host().astype(np.int64))     # letters (uint8 codes): tiny convenience inputs
