"""KmerEncoding: the tag carried by hashed k-mers (bionumpy/encodings/kmer_encodings.py:11-86)."""
import numpy as np

from ..encoded_array import Encoding, AlphabetEncoding, EncodedArray, EncodedRaggedArray


class KmerEncoding(Encoding):
    def __init__(self, alphabet_encoding, k):
        assert isinstance(alphabet_encoding, AlphabetEncoding), alphabet_encoding
        self._alphabet_encoding = alphabet_encoding
        self._k = k

    @property
    def k(self):
        return self._k

    def encode(self, data):
        """str or list of str of length k -> hashed k-mer(s) (kmer_encodings.py:25-53)"""
        n = self._alphabet_encoding.alphabet_size
        weights = n ** np.arange(self._k, dtype=np.int64)
        if isinstance(data, str):
            assert len(data) == self.k
            letters = self._alphabet_encoding.encode(data).raw().astype(np.int64)
            return EncodedArray(letters.dot(weights), self)
        if isinstance(data, (list, EncodedRaggedArray)):
            assert all(len(row) == self.k for row in data)
            enc = self._alphabet_encoding.encode(data) if isinstance(data, list) else data
            letters = enc.ravel().raw().astype(np.int64).reshape(-1, self._k)
            return EncodedArray(letters.dot(weights), self)
        raise NotImplementedError

    def to_string(self, kmer):
        """first base is the least significant digit (kmer_encodings.py:55-70, tests/test_kmer.py:85-94)"""
        if np.asanyarray(kmer).ndim > 0:
            return ",".join(self.to_string(k) for k in kmer)
        n = self._alphabet_encoding.alphabet_size
        kmer = int(kmer)
        if n == 4:
            digits = [(kmer >> (2 * j)) & 3 for j in range(self._k)]
        else:
            digits = [(kmer // n ** j) % n for j in range(self._k)]
        alphabet = self._alphabet_encoding.get_alphabet()
        return "".join(alphabet[d] for d in digits)

    def get_labels(self):
        assert self._k <= 8, "Only supported for k <= 5"
        return [self.to_string(kmer) for kmer in range(self._alphabet_encoding.alphabet_size ** self._k)]

    def __str__(self):
        return "%dmerEncoding(%s)" % (self._k, self._alphabet_encoding)

    def __repr__(self):
        return "KmerEncoding(%s, %d)" % (self._alphabet_encoding, self._k)

    def __eq__(self, other):
        return isinstance(other, KmerEncoding) and self._k == other._k \
            and self._alphabet_encoding == other._alphabet_encoding

    def __hash__(self):
        return hash(repr(self))
