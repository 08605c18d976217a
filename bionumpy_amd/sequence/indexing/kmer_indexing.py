"""KmerIndex / KmerLookup (bionumpy/sequence/indexing/kmer_indexing.py:7-76) on the MI355X path.

The reference loops over every distinct k-mer with an O(U*N) numpy scan; here the index is the
sorted, de-duplicated (kmer, row) pair list, built by the sparse counting kernels (the pairs are the distinct
values of rank(kmer) * n_rows + row, ops.unique_pairs), and a lookup is a pair of binary searches
(``bnpk_search_sorted``) — same answers: the ascending row ids that contain
the k-mer, ``[]`` for an unseen k-mer.
"""
import numpy as np

from ...device import HArray
from ...encoded_array import as_encoded_array
from ...ops import get_ops
from ..kmers import get_kmers


class KmerIndex:
    def __init__(self, k, pair_keys, pair_rows, sequences_encoding, pair_counts=None):
        self._k = k
        self._keys = pair_keys          # HArray int64, sorted (kmer of every distinct (kmer,row) pair)
        self._rows = pair_rows          # HArray int64, row ids, ascending within a kmer
        self._counts = pair_counts      # HArray int64 or None: occurrences of the k-mer in the row
        self._sequences_encoding = sequences_encoding

    def __repr__(self):
        return "%d-merIndex of sequences with %s" % (self._k, self._sequences_encoding)

    @property
    def k(self):
        return self._k

    @classmethod
    def create_index(cls, sequences, k, multiplicities=False):
        """``multiplicities``: also keep how often every k-mer occurs in every row (the colored de Bruijn graph's lists)"""
        ops = get_ops()
        kmers = get_kmers(sequences, k)
        kmers._compact()
        rows = ops.row_ids(kmers.offsets(), len(kmers), kmers.total())
        res = ops.unique_pairs(kmers._flat_data(), rows, key_bits=2 * k, n_values=max(len(kmers), 1),
                               with_counts=multiplicities)
        return cls(k, res[0], res[1], sequences.encoding, res[2] if multiplicities else None)

    def _encode_query(self, kmer):
        if isinstance(kmer, str):
            assert len(kmer) == self._k
            return int(get_kmers(as_encoded_array(kmer, self._sequences_encoding), self._k).raw()[0])
        return int(kmer)

    def get_indices_batch(self, kmers):
        """(lo, hi) ranges into the pair list for many int k-mers at once (device binary searches)"""
        ops = get_ops()
        q = kmers if isinstance(kmers, HArray) else HArray(host=np.asarray(kmers, dtype=np.int64))
        lo = ops.search_sorted(self._keys, q, upper=False)
        hi = ops.search_sorted(self._keys, q, upper=True)
        return lo, hi

    def count_hits(self, kmers):
        """number of indexed sequences that contain each of many int k-mers (0 for an unseen one): hi - lo of
        ``get_indices_batch``, on the device for device queries.  With the index built on every rank from a file opened with
        ``shard=False`` and the queries taken from a file every rank reads its own part of (``bnp.open``'s default in a
        torchrun job), this is SURVEY §8e's "replicate the index, shard the lookups": no exchange at all."""
        lo, hi = self.get_indices_batch(kmers)
        if lo.on_device:
            return HArray(dev=hi.dev() - lo.dev())
        return hi.host() - lo.host()

    @staticmethod
    def _part(h, lo, hi):
        """h[lo:hi] on the host — only those elements cross the link"""
        return HArray(dev=h.dev()[lo:hi]).host() if h.on_device else h.host()[lo:hi]

    def get_indices(self, kmer):
        lo, hi = self.get_indices_batch(np.array([self._encode_query(kmer)], dtype=np.int64))
        lo, hi = int(lo.host()[0]), int(hi.host()[0])
        if hi == lo:
            return []
        return self._part(self._rows, lo, hi)

    def get_indices_with_repeats(self, kmer):
        """the row of every occurrence of the k-mer, in row order (needs ``multiplicities=True``)"""
        assert self._counts is not None, "create_index(..., multiplicities=True)"
        lo, hi = self.get_indices_batch(np.array([self._encode_query(kmer)], dtype=np.int64))
        lo, hi = int(lo.host()[0]), int(hi.host()[0])
        return np.repeat(self._part(self._rows, lo, hi), self._part(self._counts, lo, hi))


class KmerLookup:
    index_class = KmerIndex

    def __init__(self, kmer_index, sequences):
        self._kmer_index = kmer_index
        self._sequences = sequences

    def __repr__(self):
        return "Lookup on %d-merIndex of %d sequences" % (self._kmer_index.k, len(self._sequences))

    @classmethod
    def create_lookup(cls, sequences, *args, **kwargs):
        index = cls.index_class.create_index(sequences, *args, **kwargs)
        return cls(index, sequences)

    def get_sequences(self, kmer):
        idx = self._kmer_index.get_indices(kmer)
        return self._sequences[np.asarray(idx, dtype=np.int64)]
