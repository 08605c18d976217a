import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, warnings
import bionumpy_amd as bnp
p = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "big.fq.gz")
base = "/tmp/mm_reads"
def loader():
    for chunk in bnp.open(p).read_chunks(100000):
        yield bnp.change_encoding(chunk.sequence, bnp.DNAEncoding)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    created = bnp.MemMapEncodedRaggedArray.create(loader, base)
whole = bnp.change_encoding(bnp.open(p).read().sequence, bnp.DNAEncoding)
loaded = bnp.MemMapEncodedRaggedArray.load(base)
a = bnp.sequence.count_kmers(loaded, 31); b = bnp.sequence.count_kmers(whole, 31)
print(a.encoding == b.encoding, len(a), len(b), a._key_bits, b._key_bits)
ka, kb = a.keys, b.keys
print(ka[:3], kb[:3], np.array_equal(ka, kb), np.array_equal(a.counts, b.counts))
h1 = np.asarray(bnp.get_kmers(loaded, 31).raw().ravel()); h2 = np.asarray(bnp.get_kmers(whole, 31).raw().ravel())
print(h1.size, h2.size, np.array_equal(h1, h2), np.flatnonzero(h1 != h2)[:5] if h1.size == h2.size else None)
