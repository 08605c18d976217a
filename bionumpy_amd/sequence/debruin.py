"""De Bruijn graph lookups (bionumpy/sequence/debruin.py:8-62) on the MI355X path.

The reference keeps a Python ``set`` of the int k-mers (``DeBruijnGraph``) or a ``dict`` k-mer -> row list
(``ColoredDeBruijnGraph``) and answers one query at a time.  Here the set is the sorted distinct k-mers the sparse
counting kernels produce (bnpk_kmers_partition / bnpk_radix_partition / bnpk_finish_sorted), membership is a binary
search on the device (bnpk_search_sorted), and the coloured graph is the KmerIndex pair list with multiplicities.
Same answers: ``forward`` / ``backward`` list the neighbours in the order the reference tries them (the new base
A, C, G, T), ``graph[kmer]`` lists the row of every occurrence in row order.
"""
import numpy as np

from ..device import HArray
from ..encoded_array import as_encoded_array
from ..encodings import DNAEncoding
from ..encodings.kmer_encodings import KmerEncoding
from ..ops import get_ops
from .indexing.kmer_indexing import KmerIndex
from .kmers import get_kmers


class DeBruijnGraph:
    def __init__(self, kmer_set, k):
        self._kmer_set = kmer_set if isinstance(kmer_set, HArray) else \
            HArray(host=np.unique(np.asarray(sorted(kmer_set), dtype=np.int64)))   # sorted distinct int k-mers
        self._kmer_encoding = KmerEncoding(DNAEncoding, k)
        self._k = k

    @classmethod
    def from_sequences(cls, sequences, k=31):
        ops = get_ops()
        kmers = get_kmers(as_encoded_array(sequences, DNAEncoding), k)
        kmers._compact()
        keys, _ = ops.count_sparse(kmers._flat_data(), key_bits=2 * k)
        return cls(keys, k)

    def __len__(self):
        return self._kmer_set.size

    def _hash(self, kmer):
        if isinstance(kmer, str):
            assert len(kmer) == self._k
            return int(get_kmers(as_encoded_array(kmer, DNAEncoding), self._k).raw()[0])
        return int(kmer)

    def contains(self, kmers):
        """membership of many int k-mers at once (bool numpy array)"""
        ops = get_ops()
        q = HArray(host=np.ascontiguousarray(kmers, dtype=np.int64))
        if self._kmer_set.size == 0 or q.size == 0:
            return np.zeros(q.size, dtype=bool)
        lo = ops.search_sorted(self._kmer_set, q, upper=False).host()
        hi = ops.search_sorted(self._kmer_set, q, upper=True).host()
        return hi > lo

    def _neighbours(self, kmer, forward):
        """the four k-mers that can follow (precede) ``kmer``: its last k - 1 bases shifted down and every base on top, or its
        first k - 1 bases shifted up and every base below (debruin.py:20-27), as one int64 vector"""
        bases = np.arange(4, dtype=np.int64)
        if forward:
            return (int(kmer) >> 2) + (bases << (2 * (self._k - 1)))
        return ((int(kmer) << 2) & (4 ** self._k - 1)) + bases

    def _present(self, candidates):
        found = self.contains(candidates)
        return [self._kmer_encoding.to_string(int(c)) for c, f in zip(candidates, found) if f]

    def forward(self, kmer):
        return self._present(self._neighbours(self._hash(kmer), True))

    def backward(self, kmer):
        return self._present(self._neighbours(self._hash(kmer), False))


class ColoredDeBruijnGraph:
    def __init__(self, index, k):
        self._index = index
        self._kmer_encoding = KmerEncoding(DNAEncoding, k)

    @classmethod
    def from_sequences(cls, sequences, k):
        return cls(KmerIndex.create_index(as_encoded_array(sequences, DNAEncoding), k, multiplicities=True), k)

    def __getitem__(self, kmer):
        return [int(r) for r in self._index.get_indices_with_repeats(kmer)]
