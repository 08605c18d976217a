"""API-level tests of bionumpy_amd, written to read like the reference's own tests for this path
(tests/test_kmer.py, test_minimizers.py, test_kmer_index.py, test_parsers.py, test_io.py,
test_io_exceptions.py, test_encodings.py, docs_source/topics/kmers.rst).  See tests/backends.py for the
two backends each test runs under."""
import io
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from backends import bnp  # noqa: E402,F401

import oracle  # noqa: E402

FASTQ_TEXT = "@headerishere\nCTTGTTGA\n+\n!!!!!!!!\n@anotherheader\nCGG\n+\n~~~\n"
FASTA_TEXT = ">header\nCTTGTTGA\n>header2\nCGG\n"
MULTILINE_TEXT = ">header\nCTTGCC\nGCCTCC\n>header2\nCCCCCC\nGGGCCC\nTTT\n"


def _reader(bnp, text, buffer_type, prepend=False):
    r = bnp.io.NumpyFileReader(io.BytesIO(text.encode("ascii") if isinstance(text, str) else text), buffer_type)
    if prepend:
        r.set_prepend_mode()
    return bnp.io.NpDataclassReader(r)


def _ragged_strings(ragged):
    return [str(r) for r in ragged]


# ------------------------------------------------------------------------------------ k-mers
def test_get_kmers_docstring(bnp):
    # bionumpy/sequence/kmers.py:56-61
    sequences = bnp.as_encoded_array(["ACTG", "AAA", "TTGGC"], bnp.DNAEncoding)
    kmers = bnp.sequence.get_kmers(sequences, 3)
    assert [[str(k) for k in row] for row in kmers] == [["ACT", "CTG"], ["AAA"], ["TTG", "TGG", "GGC"]]
    assert repr(kmers) == ("encoded_ragged_array([[ACT, CTG],\n"
                           "                      [AAA],\n"
                           "                      [TTG, TGG, GGC]], 3merEncoding(AlphabetEncoding('ACGT')))")


def test_named_alphabets_of_the_reference(bnp):
    # encodings/alphabet_encoding.py:102-111: the names a caller of the reference imports, with the reference's semantics
    assert bnp.as_encoded_array("acgtn", bnp.ACGTnEncoding).raw().tolist() == [0, 1, 2, 3, 4]
    assert bnp.as_encoded_array("ACTG", bnp.ACTGEncoding).raw().tolist() == [0, 1, 2, 3]
    assert bnp.as_encoded_array("acug", bnp.RNAENcoding).raw().tolist() == [0, 1, 2, 3] and bnp.RNAENcoding is bnp.ACUGEncoding
    assert bnp.as_encoded_array("0917", bnp.DigitEncoding).raw().tolist() == [0, 9, 1, 7]
    prot = bnp.as_encoded_array(["MKV*", "ACDW"], bnp.AminoAcidEncoding)
    labels = bnp.AminoAcidEncoding.get_labels()
    assert [[labels[c] for c in row] for row in prot.raw()] == [list("MKV*"), list("ACDW")] and prot.tolist() == ["MKV*", "ACDW"]
    assert bnp.as_encoded_array("=ACMN", bnp.BamEncoding).raw().tolist() == [0, 1, 2, 3, 15]
    with pytest.raises(bnp.EncodingError):
        bnp.as_encoded_array("ACGU", bnp.ACGTnEncoding)
    # k-mers over a five-letter alphabet: the generic kernel against the oracle
    reads = ["ACGTNNACGT", "NNN", "GATTACA"]
    seqs = bnp.as_encoded_array(reads, bnp.ACGTnEncoding)
    codes = np.concatenate([np.array(["ACGTN".index(c) for c in r], dtype=np.uint8) for r in reads])
    expect, lens = oracle.get_kmers_generic(codes, np.array([len(r) for r in reads]), 3, 5)
    kmers = bnp.sequence.get_kmers(seqs, 3)
    assert np.array_equal(np.concatenate([np.asarray(r) for r in kmers.raw()]), expect) and [len(r) for r in kmers] == lens.tolist()
    assert str(kmers[0][0]) == "ACG" and str(kmers[1][0]) == "NNN"


def test_generic_alphabets(bnp):
    # any AlphabetEncoding encodes (encodings/alphabet_encoding.py:19-46) and hashes (sequence/kmers.py:17-27,82-87)
    acgtn = bnp.AlphabetEncoding("ACGTN")
    enc = bnp.as_encoded_array("ACGTNacgtn", acgtn)
    assert enc.raw().tolist() == [0, 1, 2, 3, 4, 0, 1, 2, 3, 4] and str(enc) == "ACGTNACGTN"
    with pytest.raises(bnp.EncodingError) as e:
        bnp.as_encoded_array(["ACGT", "NNX", "A"], acgtn)
    assert e.value.offset == 6
    seqs = bnp.as_encoded_array(["ACGTN", "NN", "", "GATTACANNNGATTACA"], acgtn)
    kmers = bnp.sequence.get_kmers(seqs, 3)
    assert [[str(k) for k in row] for row in kmers][:3] == [["ACG", "CGT", "GTN"], [], []]
    codes = np.concatenate([r for r in seqs.raw()]).astype(np.uint8)
    lens = np.array([5, 2, 0, 17])
    expect, new_lens = oracle.get_kmers_generic(codes, lens, 3, 5)
    assert np.array_equal(np.concatenate([np.asarray(r) for r in kmers.raw()]), expect)
    assert [len(r) for r in kmers] == new_lens.tolist() == [3, 0, 0, 15]
    counts = bnp.count_encoded(kmers, axis=None)
    assert counts["NNN"] == 1 and counts["TAC"] == 2 and counts["AAA"] == 0
    # amino acids, k large enough for the int64 hash to wrap like numpy's dot
    protein = bnp.AlphabetEncoding("ACDEFGHIKLMNPQRSTVWY")
    text = "MKVLAAGIVGLHRHSWYWYWYMKVLAAGIVGLHRHSWY"
    one = bnp.as_encoded_array(text, protein)
    for k in (1, 5, 14, 20):
        got = bnp.sequence.get_kmers(one, k).raw()
        expect, _ = oracle.get_kmers_generic(one.raw().astype(np.uint8), np.array([len(text)]), k, 20)
        assert np.array_equal(got, expect), k
    # tests/test_kmer.py:27-30: the generic encoder agrees with the 2-bit path on DNA
    dna = bnp.as_encoded_array(["cgtt", "AacACtggatcggacTTATCTGACG", "G"], bnp.DNAEncoding)
    fast = bnp.sequence.get_kmers(dna, 3)
    generic = bnp.sequence.kmers.KmerEncoder(3, bnp.DNAEncoding).rolling_window(dna)
    assert fast.tolist() == generic.tolist() and fast.encoding == generic.encoding
    assert int(bnp.sequence.kmers.KmerEncoder(3, bnp.DNAEncoding)("ACG").raw()[0]) == 0 + 1 * 4 + 2 * 16


def test_numpy_functions_on_encoded_arrays(bnp, big_fq_gz):
    # encoded_array.py:454-486 (__array_function__) and np.concatenate(chunks) (lazybnpdataclass.py:178-196)
    a = bnp.as_encoded_array("ACGT", bnp.DNAEncoding)
    b = bnp.as_encoded_array("TTG", bnp.DNAEncoding)
    both = np.concatenate([a, b])
    assert isinstance(both, bnp.EncodedArray) and both.encoding == bnp.DNAEncoding and str(both) == "ACGTTTG"
    assert np.bincount(both, minlength=4).tolist() == [1, 1, 2, 3] and np.bincount(both).tolist() == [1, 1, 2, 3]
    assert np.argsort(both).tolist() == np.argsort(both.raw()).tolist()
    assert str(np.append(a, b)) == "ACGTTTG" and str(np.zeros_like(a)) == "AAAA"
    assert str(np.where(np.array([True, False, True, False]), a, bnp.as_encoded_array("TTTT", bnp.DNAEncoding))) == "ATGT"
    assert np.lexsort((a, np.array([1, 0, 1, 0]))).tolist() == [1, 3, 0, 2]
    ragged = [bnp.as_encoded_array(["ACG", "", "T"], bnp.DNAEncoding), bnp.as_encoded_array(["GG", "TTTT"], bnp.DNAEncoding)]
    joined = np.concatenate(ragged)
    assert isinstance(joined, bnp.EncodedRaggedArray) and joined.tolist() == ["ACG", "", "T", "GG", "TTTT"]
    kmers = np.concatenate([bnp.sequence.get_kmers(r, 2) for r in ragged])
    assert [[str(k) for k in row] for row in kmers] == [["AC", "CG"], [], [], ["GG"], ["TT", "TT", "TT"]]
    # chunks of a file, joined again: the same entries as one read of the whole file
    chunks = list(bnp.open(big_fq_gz).read_chunks(min_chunk_size=100_000))
    assert len(chunks) > 1
    whole = bnp.open(big_fq_gz).read()
    glued = np.concatenate(chunks)
    assert len(glued) == len(whole) == 1000
    for field in ("name", "sequence"):
        assert getattr(glued, field).tolist() == getattr(whole, field).tolist()
    assert np.array_equal(np.asarray(glued.quality.ravel()), np.asarray(whole.quality.ravel()))
    counts = bnp.count_encoded(bnp.sequence.get_kmers(bnp.change_encoding(glued.sequence, bnp.DNAEncoding), 3), axis=None)
    assert counts["AAA"] == 3920 and counts["ACT"] == 3038                 # SURVEY §8c derived goldens


def test_kmers_topic_doc(bnp):
    # docs_source/topics/kmers.rst:11-27
    sequences = bnp.as_encoded_array(["ACTG", "GGGACT", "G"], bnp.DNAEncoding)
    kmers = bnp.sequence.get_kmers(sequences, 3)
    assert [[str(k) for k in row] for row in kmers] == [["ACT", "CTG"], ["GGG", "GGA", "GAC", "ACT"], []]
    counts = bnp.count_encoded(kmers, axis=None)
    assert counts["ACT"] == 2
    minimizers = bnp.sequence.get_minimizers(sequences, k=2, window_size=4)
    assert [[str(k) for k in row] for row in minimizers] == [["AC"], ["GA", "GA", "GA"], []]


def test_count_kmers(bnp):
    # tests/test_kmer.py:97-102
    sequences = bnp.as_encoded_array(["ACTG", "AAA", "TTGGC"], bnp.DNAEncoding)
    kmers = bnp.sequence.get_kmers(sequences, 3)
    counts = bnp.count_encoded(kmers, axis=None)
    assert counts["ACT"] == 1
    assert counts["GGG"] == 0
    assert counts.alphabet[:5] == ["AAA", "CAA", "GAA", "TAA", "ACA"]        # tests/test_kmer.py:85-94
    assert bnp.sequence.count_kmers(sequences, 3) == counts
    per_row = bnp.count_encoded(kmers, axis=-1)
    assert per_row.counts.shape == (3, 64) and per_row.counts.sum(axis=-1).tolist() == [2, 1, 3]


def test_get_kmers_one_and_lower_case(bnp):
    # tests/test_kmer.py:60-63, :27-30 (lower case)
    kmers = bnp.sequence.get_kmers(bnp.as_encoded_array(["ACTG"], bnp.DNAEncoding), 1)
    assert len(kmers[0]) == 4 and kmers.raw().ravel().tolist() == [0, 1, 3, 2]
    lower = bnp.sequence.get_kmers(bnp.as_encoded_array("cgtt", bnp.DNAEncoding), 3)
    upper = bnp.sequence.get_kmers(bnp.as_encoded_array("CGTT", bnp.DNAEncoding), 3)
    assert np.array_equal(lower.raw(), upper.raw())


def test_rolling_hash_shape(bnp):
    # tests/test_kmer.py:33-40
    lengths = np.arange(3, 10)
    codes = bnp.EncodedArray((np.arange(lengths.sum()) % 4).astype(np.uint8), bnp.DNAEncoding)
    ragged = bnp.EncodedRaggedArray(codes, lengths)
    encoded = bnp.get_kmers(ragged, 3)
    assert np.array_equal(encoded.lengths, lengths - 3 + 1)
    expect, _ = oracle.get_kmers(codes.raw(), lengths, 3)
    assert np.array_equal(encoded.raw().ravel(), expect)


def test_bad_k_asserts(bnp):
    seqs = bnp.as_encoded_array(["ACTG"], bnp.DNAEncoding)
    for k in (0, 32):
        with pytest.raises(AssertionError):
            bnp.get_kmers(seqs, k)


@pytest.mark.parametrize("seed", [0, 1])
def test_kmers_random_ragged_vs_oracle(bnp, seed):
    rng = np.random.default_rng(seed)
    lengths = rng.integers(0, 200, size=300)
    lengths[rng.integers(0, 300, size=30)] = 0                     # empty rows
    lengths[5] = 5000                                              # one long row
    codes = rng.integers(0, 4, size=int(lengths.sum())).astype(np.uint8)
    ragged = bnp.EncodedRaggedArray(bnp.EncodedArray(codes, bnp.DNAEncoding), lengths)
    for k in (1, 2, 5, 16, 31):
        got = bnp.get_kmers(ragged, k)
        expect, el = oracle.get_kmers(codes, lengths, k)
        assert np.array_equal(got.lengths, el)
        assert np.array_equal(got.raw().ravel(), expect)
    for k, w in ((2, 4), (5, 5), (31, 40), (7, 100)):
        got = bnp.get_minimizers(ragged, k, w)
        expect, el = oracle.get_minimizers(codes, lengths, k, w)
        assert np.array_equal(got.lengths, el)
        assert np.array_equal(got.raw().ravel(), expect)


# ------------------------------------------------------------------------------------ minimizers
def test_minimizers_numeric(bnp):
    # tests/test_minimizers.py:44-62
    sequence = bnp.EncodedArray(np.array([0, 3, 1, 2, 2, 1, 0], dtype=np.uint8), bnp.DNAEncoding)
    minimizers = bnp.get_minimizers(sequence, 2, 4)
    assert minimizers.raw().tolist() == [7, 7, 6, 1]
    rows = [[0, 3, 1, 2, 2, 1, 0], [0, 3, 1, 2, 2, 1], [0, 3, 1, 2, 2], [0, 3, 1, 2]]
    flat = np.concatenate(rows).astype(np.uint8)
    ragged = bnp.EncodedRaggedArray(bnp.EncodedArray(flat, bnp.DNAEncoding), [len(r) for r in rows])
    minimizers = bnp.get_minimizers(ragged, 2, 4)
    assert minimizers.raw().tolist() == [[7, 7, 6, 1], [7, 7, 6], [7, 7], [7]]


def test_streamable_numpy_reductions(bnp, big_fq_gz):
    # bionumpy/streams/reductions.py:1-66: bincount / histogram / mean / quantile of an array or of a stream of chunks
    from bionumpy_amd.streams import BnpStream
    chunks = [np.array([1, 2, 2, 5]), np.array([0, 2, 7, 7, 7]), np.array([3])]
    whole = np.concatenate(chunks)
    assert np.array_equal(bnp.bincount(BnpStream(iter(chunks))), np.bincount(whole))
    assert np.array_equal(bnp.bincount(whole), np.bincount(whole))
    hist, edges = bnp.histogram(BnpStream(iter(chunks)), bins=4, range=(0, 8))
    assert np.array_equal(hist, np.histogram(whole, bins=4, range=(0, 8))[0]) and edges[-1] == 8
    assert bnp.mean(BnpStream(iter(chunks))) == whole.mean() and bnp.mean(whole) == whole.mean()
    blocks = [np.arange(6, dtype=float).reshape(2, 3), np.ones((4, 3))]
    assert np.allclose(bnp.mean(BnpStream(iter(blocks)), axis=0), np.concatenate(blocks).mean(axis=0))
    assert np.array_equal(bnp.quantile(BnpStream(iter(chunks)), np.array([0.5, 0.9])), [2, 7])
    # over the chunks of a file: read lengths, and the mean base quality of every read (one array per chunk)
    lengths = bnp.open(big_fq_gz).read().sequence.lengths
    stream = bnp.open(big_fq_gz).read_chunks(100000)
    assert np.array_equal(bnp.bincount(BnpStream(chunk.sequence.lengths for chunk in stream)), np.bincount(lengths))
    per_chunk = list(bnp.mean(bnp.open(big_fq_gz).read_chunks(100000).quality, axis=1))
    assert len(per_chunk) > 1 and np.allclose(np.concatenate([np.asarray(m) for m in per_chunk]),
                                               np.asarray(np.mean(bnp.open(big_fq_gz).read().quality, axis=1)))


def test_rollable_forms_minimizers_and_position_weight_matrix(bnp):
    # tests/test_minimizers.py:36-46 (Minimizers(3, KmerEncoder(2, DNAEncoding)) on the window [0, 3, 1, 2] -> [7]) and
    # tests/test_position_weight_matrix.py:54-63 (PositionWeightMatrix(pwm)(window), .rolling_window(sequence))
    encoding = bnp.Minimizers(3, bnp.KmerEncoder(2, bnp.DNAEncoding))
    assert encoding.window_size == 4
    window = bnp.EncodedArray(np.array([0, 3, 1, 2], dtype=np.uint8), bnp.DNAEncoding)
    minimizer = encoding(window)
    assert minimizer.encoding == bnp.KmerEncoding(bnp.DNAEncoding, 2) and np.asarray(minimizer.raw()).tolist() == [7]
    sequence = bnp.EncodedArray(np.array([0, 3, 1, 2, 2, 1, 0], dtype=np.uint8), bnp.DNAEncoding)
    assert np.asarray(encoding.rolling_window(sequence).raw()).tolist() == [7, 7, 6, 1]
    rows = bnp.as_encoded_array(["ATCGGCA", "ATCGGC", "ATCG"], bnp.DNAEncoding)
    assert [np.asarray(r).tolist() for r in encoding.rolling_window(rows).raw()] == \
        [np.asarray(r).tolist() for r in bnp.get_minimizers(rows, 2, 4).raw()]
    with np.errstate(divide="ignore"):
        matrix = np.log([[0.4, 0.25], [0.1, 0.25], [0.4, 0.25], [0.1, 0.25]])
    pwm = bnp.PWM(matrix, "ACGT")
    scorer = bnp.PositionWeightMatrix(pwm)
    assert scorer.window_size == 2
    assert np.allclose(np.exp(scorer(bnp.EncodedArray(np.array([0, 1], dtype=np.uint8), bnp.DNAEncoding))), 0.4 * 0.25)
    acgt = bnp.EncodedArray(np.array([0, 1, 2, 3], dtype=np.uint8), bnp.DNAEncoding)
    assert np.allclose(np.exp(scorer.rolling_window(acgt)), [0.4 * 0.25, 0.025, 0.4 * 0.25])


def test_minimizer_strings(bnp):
    # tests/test_minimizers.py:65-80, bionumpy/sequence/minimizers.py:39-46
    sequences = bnp.as_encoded_array(["CCCAAACCCC", "TTTTCCCTTT"], bnp.DNAEncoding)
    minimizers = bnp.get_minimizers(sequences, 3, 10)
    assert [[str(m) for m in row] for row in minimizers] == [["AAA"], ["CCC"]]
    sequences = bnp.as_encoded_array(["ACTG", "AAA", "TTGGC"], bnp.DNAEncoding)
    minimizers = bnp.get_minimizers(sequences, 2, 4)
    assert [[str(m) for m in row] for row in minimizers] == [["AC"], [], ["GG", "GC"]]


# ------------------------------------------------------------------------------------ k-mer index
def test_kmer_index(bnp):
    # tests/test_kmer_index.py:12-28
    sequences = bnp.as_encoded_array(["ACGTAA", "GCTAAA"], bnp.DNAEncoding)
    index = bnp.KmerIndex.create_index(sequences, k=3)
    assert index.get_indices("ACG") == [0]
    assert index.get_indices("AAA") == [1]
    np.testing.assert_equal(index.get_indices("TAA"), [0, 1])
    assert index.get_indices("GAA") == []
    index2 = bnp.KmerIndex.create_index(sequences, k=2)
    np.testing.assert_equal(index2.get_indices("AA"), [0, 1])
    lookup = bnp.KmerLookup.create_lookup(sequences, k=3)
    assert lookup.get_sequences(kmer="CGT").tolist() == ["ACGTAA"]


def test_row_values_stay_numpy_like(bnp):
    # scripts/small_example.py:36-46: np.mean(chunk.quality, axis=1) > 30, masks combined, chunk[mask] — the per-row values
    # live in HBM (device_vector.py) and have to behave like the numpy arrays the reference returns
    rng = np.random.default_rng(9)
    n = 3000
    lens = rng.integers(1, 60, size=n)
    seqs = ["".join(rng.choice(list("ACGT"), size=int(l))) for l in lens]
    quals = [bytes(rng.integers(33, 74, size=int(l)).astype(np.uint8)).decode() for l in lens]
    text = "".join("@r%d\n%s\n+\n%s\n" % (i, s, q) for i, (s, q) in enumerate(zip(seqs, quals)))
    chunk = _reader(bnp, text, bnp.FastQBuffer).read()
    scores = [np.frombuffer(q.encode(), dtype=np.uint8).astype(np.int64) - 33 for q in quals]
    means_ref = np.array([s.mean() for s in scores])
    mins_ref = np.array([s.min() for s in scores])
    means = np.mean(chunk.quality, axis=1)
    mins = np.min(chunk.quality, axis=1)
    assert np.array_equal(np.asarray(means), means_ref) and np.array_equal(np.asarray(mins), mins_ref)
    assert means.shape == (n,) and len(means) == n and means.dtype == np.float64
    assert float(np.median(means[:1000])) == float(np.median(means_ref[:1000]))
    assert means[7] == means_ref[7] and np.array_equal(means[10:20], means_ref[10:20])
    mask = (means > 20) & (mins >= 3)
    mask_ref = (means_ref > 20) & (mins_ref >= 3)
    assert np.array_equal(np.asarray(mask), mask_ref) and mask.sum() == mask_ref.sum() and np.sum(mask) == mask_ref.sum()
    assert np.array_equal(np.flatnonzero(mask), np.flatnonzero(mask_ref))
    assert np.array_equal(np.asarray(~mask), ~mask_ref) and np.array_equal(np.asarray(mask | (means < 5)), mask_ref | (means_ref < 5))
    assert np.array_equal(np.asarray(mask & mask_ref), mask_ref)                    # a device mask and a numpy mask
    mask[::3] = False
    mask_ref[::3] = False
    mask[5:9] = True
    mask_ref[5:9] = True
    assert np.array_equal(np.asarray(mask), mask_ref)
    kept = chunk[mask]
    assert len(kept) == int(mask_ref.sum())
    assert kept.sequence.tolist() == [s for s, m in zip(seqs, mask_ref) if m]
    assert kept.get_buffer().entry_bytes().host().tobytes().decode() == \
        "".join("@r%d\n%s\n+\n%s\n" % (i, s, q) for i, (s, q, m) in enumerate(zip(seqs, quals, mask_ref)) if m)
    assert np.allclose(means * 2 + 1, means_ref * 2 + 1) and np.allclose(np.sqrt(means), np.sqrt(means_ref))
    assert means.tolist() == means_ref.tolist() and (means == means_ref).all() and not (means != means_ref).any()
    assert np.array_equal(np.asarray(means[mask]), means_ref[mask_ref])


def test_debruijn_graphs(bnp):
    # tests/test_debruijn.py:10-35
    from bionumpy_amd.sequence.debruin import DeBruijnGraph, ColoredDeBruijnGraph
    sequences = ["acg", "cgtc"]
    graph = DeBruijnGraph.from_sequences(sequences, 2)
    assert isinstance(graph, DeBruijnGraph) and len(graph) == 4          # AC CG GT TC
    assert graph.forward("ac") == ["CG"]
    assert graph.backward("tc") == ["GT"]
    assert graph.forward("tc") == ["CG"] and graph.backward("ac") == []      # C? after TC: CG is there; ?A before AC: nothing
    colored = ColoredDeBruijnGraph.from_sequences(sequences, 2)
    assert colored["ac"] == [0]
    assert colored["tc"] == [1]
    assert colored["cg"] == [0, 1]
    assert colored["aa"] == []
    # larger, against the oracle: every neighbour list, and rows with a k-mer more than once
    rng = np.random.default_rng(5)
    reads = ["".join(rng.choice(list("ACGT"), size=int(n))) for n in rng.integers(0, 60, size=40)] + ["ACACACAC"]
    k = 4
    codes = np.concatenate([oracle.encode_dna(np.frombuffer(r.encode(), dtype=np.uint8)) for r in reads]).astype(np.uint8)
    lens = np.array([len(r) for r in reads])
    kmers, _ = oracle.get_kmers(codes, lens, k)
    kmer_set = np.unique(kmers)
    graph = DeBruijnGraph.from_sequences(reads, k)
    assert len(graph) == kmer_set.size
    for q in list(kmer_set[:25]) + [0, 4 ** k - 1]:
        s = oracle.kmer_to_string(int(q), k)
        assert graph.forward(s) == [oracle.kmer_to_string(x, k) for x in oracle.debruijn_neighbours(kmer_set, int(q), k, True)]
        assert graph.backward(s) == [oracle.kmer_to_string(x, k) for x in oracle.debruijn_neighbours(kmer_set, int(q), k, False)]
    colors = oracle.colored_debruijn(codes, lens, k)
    colored = ColoredDeBruijnGraph.from_sequences(reads, k)
    for q in list(colors)[:40] + [oracle.kmer_from_string("ACAC")]:
        assert colored[oracle.kmer_to_string(q, k)] == colors.get(q, [])
    assert colored["ACAC"].count(len(reads) - 1) == 3


# ------------------------------------------------------------------------------------ encodings
def test_encodings(bnp):
    # docs_source/source/encoding.rst:15-19,40-42; tests/test_encodings.py:30-41,118-122
    assert bnp.as_encoded_array("ACGT", bnp.DNAEncoding).raw().tolist() == [0, 1, 2, 3]
    assert bnp.as_encoded_array("acgt", bnp.DNAEncoding).raw().tolist() == [0, 1, 2, 3]
    encoded = bnp.as_encoded_array(["AacG", "", "t"], bnp.DNAEncoding)
    assert encoded.tolist() == ["AACG", "", "T"]
    assert encoded.raw().tolist() == [[0, 0, 1, 2], [], [3]]
    back = bnp.change_encoding(encoded, bnp.BaseEncoding)
    assert back.tolist() == ["AACG", "", "T"] and back.encoding == bnp.BaseEncoding
    assert back.raw().ravel().tolist() == [65, 65, 67, 71, 84]
    with pytest.raises(bnp.EncodingError) as e:
        bnp.as_encoded_array(["ACG", "TNA"], bnp.DNAEncoding)
    assert e.value.offset == 4
    q = bnp.QualityEncoding.encode(bnp.as_encoded_array("!#"))
    assert np.asarray(q).tolist() == [0, 2]
    assert str(bnp.EncodedArray(np.array([0, 1, 2, 3]), bnp.DNAEncoding)) == "ACGT"      # encoded_array.py:263-266


def test_change_encoding_roundtrip_random(bnp):
    # tests/property_tests/test_encodings.py:18-25: encode . decode == upper-cased input
    rng = np.random.default_rng(5)
    strings = ["".join(rng.choice(list("ACGTacgt"), size=n)) for n in rng.integers(0, 70, size=40)]
    base = bnp.as_encoded_array(strings)
    dna = bnp.change_encoding(base, bnp.DNAEncoding)
    assert bnp.change_encoding(dna, bnp.BaseEncoding).tolist() == [s.upper() for s in strings]


# ------------------------------------------------------------------------------------ file decode
def test_fastq_buffer(bnp):
    # tests/buffers.py:17-25,104-106; tests/test_parsers.py:27-31
    buf = bnp.FastQBuffer.from_raw_buffer(np.frombuffer(FASTQ_TEXT.encode(), dtype=np.uint8))
    data = buf.get_data()
    assert data.name.tolist() == ["headerishere", "anotherheader"]
    assert data.sequence.tolist() == ["CTTGTTGA", "CGG"]
    assert data.quality.tolist() == [[0] * 8, [93] * 3]
    assert len(data) == 2 and buf.size == len(FASTQ_TEXT) and buf.n_lines == 8


def test_two_line_fasta_buffer(bnp):
    # tests/buffers.py:26-31,107-109
    buf = bnp.TwoLineFastaBuffer.from_raw_buffer(np.frombuffer(FASTA_TEXT.encode(), dtype=np.uint8))
    data = buf.get_data()
    assert data.name.tolist() == ["header", "header2"] and data.sequence.tolist() == ["CTTGTTGA", "CGG"]


def test_multiline_fasta(bnp, tmp_path):
    # tests/buffers.py:32-40,110-112
    data = _reader(bnp, MULTILINE_TEXT, bnp.MultiLineFastaBuffer).read()
    assert data.name.tolist() == ["header", "header2"]
    assert data.sequence.tolist() == ["CTTGCCGCCTCC", "CCCCCCGGGCCCTTT"]
    p = tmp_path / "x.fa"
    p.write_text(MULTILINE_TEXT * 3)
    names, seqs = [], []
    for chunk in bnp.open(str(p)).read_chunks(min_chunk_size=30):
        names += chunk.name.tolist()
        seqs += chunk.sequence.tolist()
    assert names == ["header", "header2"] * 3 and seqs == ["CTTGCCGCCTCC", "CCCCCCGGGCCCTTT"] * 3
    kmers = bnp.get_kmers(bnp.change_encoding(data.sequence, bnp.DNAEncoding), 5)
    assert kmers.lengths.tolist() == [8, 11]


def test_multiline_fasta_writer_and_tables(bnp, tmp_path, golden_dir):
    # multiline_buffer.py:46-62 (get_data) and :68-86 (from_data: lines of 80 letters)
    rng = np.random.default_rng(3)
    names = ["chrI", "x", "a_rather_long_header with spaces", "last"]
    lens = [1, 80, 81, 333]
    seqs = ["".join(rng.choice(list("ACGT"), size=n)) for n in lens]
    entries = bnp.SequenceEntry(bnp.as_encoded_array(names), bnp.as_encoded_array(seqs, bnp.DNAEncoding))
    text = bnp.MultiLineFastaBuffer.from_data(entries).host().tobytes().decode("ascii")
    expect = "".join(">%s\n%s" % (n, "".join(s[i:i + 80] + "\n" for i in range(0, len(s), 80))) for n, s in zip(names, seqs))
    assert text == expect
    flat_names = np.frombuffer("".join(names).encode(), dtype=np.uint8)
    flat_seqs = np.frombuffer("".join(seqs).encode(), dtype=np.uint8)
    assert text.encode() == oracle.multiline_from_data(flat_names, [len(n) for n in names], flat_seqs, lens).tobytes()
    # written through bnp.open(.., "w") and read back (with CRLF line ends too)
    p = tmp_path / "out.fa"
    with bnp.open(str(p), "w") as f:
        f.write(entries)
    assert p.read_text() == expect
    back = bnp.open(str(p)).read()
    assert back.name.tolist() == names and back.sequence.tolist() == seqs
    crlf = tmp_path / "crlf.fa"
    crlf.write_bytes(expect.replace("\n", "\r\n").encode())
    back = bnp.open(str(crlf)).read()
    assert back.name.tolist() == names and back.sequence.tolist() == seqs
    # the reference's fixture file and the oracle's tables
    raw, res = oracle.open_text(os.path.join(golden_dir, "multi_line.fa")).read()
    data = bnp.open(os.path.join(golden_dir, "multi_line.fa")).read()
    assert [len(s) for s in data.sequence.tolist()] == res.seq_lens.tolist()
    assert "".join(data.sequence.tolist()).encode() == oracle.gather_rows(raw, res.line_starts, res.line_lens).tobytes()


def test_carriage_returns(bnp):
    # tests/test_io.py:233-249
    data = _reader(bnp, FASTQ_TEXT.replace("\n", "\r\n"), bnp.FastQBuffer).read()
    assert data.sequence.tolist() == ["CTTGTTGA", "CGG"] and data.name.tolist() == ["headerishere", "anotherheader"]


MALFORMED = [("@header\nactg\n-\n!!!!\n", 2), ("header\nactg\n+\n!!!!\n", 0),
             ("@header\nactg\n+\n@header\nactg\n+\n@header\nactg\n+\n", 4)]


@pytest.mark.parametrize("text,line", MALFORMED)
def test_fastq_raises_format_exception(bnp, text, line):
    # tests/test_io_exceptions.py:55-61
    with pytest.raises(bnp.FormatException) as e:
        bnp.FastQBuffer.from_raw_buffer(np.frombuffer(text.encode(), dtype=np.uint8)).get_data()
    assert e.value.line_number == line


def test_fasta_raises_format_exception(bnp):
    # tests/test_io_exceptions.py:35-41,64-73
    text = ">header\nacggtt\nacggtt\n>header\nacgtt\n"
    with pytest.raises(bnp.FormatException) as e:
        bnp.TwoLineFastaBuffer.from_raw_buffer(np.frombuffer(text.encode(), dtype=np.uint8)).get_data()
    assert e.value.line_number == 2


@pytest.mark.parametrize("text,line", MALFORMED)
def test_npdataclass_raises_format_exception(bnp, text, line):
    # tests/test_io_exceptions.py:86-100: line numbers accumulate across chunks
    valid = "@header\nacgtt\n+\n!!!!!\n"
    reader = _reader(bnp, valid * 100 + text, bnp.FastQBuffer)
    with pytest.raises(bnp.FormatException) as e:
        for _ in reader.read_chunks(200):
            pass
    assert e.value.line_number == 4 * 100 + line


def test_read_chunks_big_fq(bnp, big_fq_gz):
    # bionumpy/io/files.py:117-162 (511 + 489 entries at 300000-byte chunks); README.rst:38-42 (53686 G's)
    f = bnp.open(big_fq_gz)
    chunk = f.read_chunk(min_chunk_size=300000)
    assert len(chunk) == 511
    chunk2 = f.read_chunk(min_chunk_size=300000)
    assert len(chunk2) == 489
    n_g = sum(int(np.sum((c.sequence == "G").ravel())) for c in bnp.open(big_fq_gz).read_chunks(300000))
    assert n_g == 53686
    whole = bnp.open(big_fq_gz).read()
    assert len(whole) == 1000
    seq_whole = whole.sequence.ravel().raw()
    parts = [c.sequence.ravel().raw() for c in bnp.open(big_fq_gz).read_chunks(min_chunk_size=1000)]
    assert np.array_equal(np.concatenate(parts), seq_whole)                      # tests/test_io.py:95-118


def test_doc_31mers_of_big_fq(bnp, big_fq_gz):
    # docs_source/topics/kmers.rst:40-78
    file = bnp.open(big_fq_gz)
    for chunk in file.read_chunks():
        sequences = bnp.change_encoding(chunk.sequence, bnp.DNAEncoding)
        kmers = bnp.get_kmers(sequences, k=31)
        assert _ragged_strings(kmers[0:3, 0:2]) == [
            "[CGGTAGCCAGCTGCGTTCAGTATGGAAGATT, GGTAGCCAGCTGCGTTCAGTATGGAAGATTT]",
            "[GATGCATACTTCGTTCGATTTCGTTTCAACT, ATGCATACTTCGTTCGATTTCGTTTCAACTG]",
            "[GTTTTGTCGCTGCGTTCAGTTTATGGGTGCG, TTTTGTCGCTGCGTTCAGTTTATGGGTGCGG]"]
        numeric = kmers.raw()
        assert numeric[0:3, 0:2].tolist() == [[4360244785522956521, 4548825710201280058],
                                             [3755975642940518834, 3244836919948823660],
                                             [2804282287455632382, 3006913581077602047]]
        assert numeric.ravel()[0:4].tolist() == [4360244785522956521, 4548825710201280058,
                                                 3443049436764013966, 860762359191003491]
        assert str(bnp.get_kmers(sequences, 31)[0, 0:3]) == \
            "[CGGTAGCCAGCTGCGTTCAGTATGGAAGATT, GGTAGCCAGCTGCGTTCAGTATGGAAGATTT, GTAGCCAGCTGCGTTCAGTATGGAAGATTTG]"


def test_count_kmers_fused_equals_get_kmers_then_count(bnp):
    # count_kmers(axis=None, k > 8) generates the hashes inside the counting sort; same result as the
    # reference's two-step form (sequence/kmers.py:129-145), rows shorter than k included
    rng = np.random.default_rng(5)
    rows = ["".join(rng.choice(list("ACGT"), size=n)) for n in (40, 3, 31, 0 + 12, 150, 30, 64, 65, 11)]
    sequences = bnp.as_encoded_array(rows, bnp.DNAEncoding)
    for k in (9, 11, 14, 31):
        two_step = bnp.count_encoded(bnp.sequence.get_kmers(sequences, k), axis=None)
        fused = bnp.sequence.count_kmers(sequences, k)
        assert fused == two_step
        assert int(fused.counts.sum()) == sum(max(0, len(r) - k + 1) for r in rows)
    short = bnp.as_encoded_array(["ACGT", "AC"], bnp.DNAEncoding)
    assert len(bnp.sequence.count_kmers(short, 31)) == 0
    # 9 <= k <= 13 with enough k-mers: one dense counting pass + compaction of the non-zero bins, the same pairs
    rows = ["".join(rng.choice(list("ACGT"), size=int(n))) for n in rng.integers(0, 400, size=300)]
    sequences = bnp.as_encoded_array(rows, bnp.DNAEncoding)
    codes = np.concatenate([oracle.encode_dna(np.frombuffer(r.encode(), dtype=np.uint8)) for r in rows]).astype(np.uint8)
    lens = np.array([len(r) for r in rows])
    for k in (9, 10):
        fused = bnp.sequence.count_kmers(sequences, k)
        ek, ec = oracle.count_sparse(oracle.get_kmers(codes, lens, k)[0])
        assert np.array_equal(np.asarray(fused.keys), ek) and np.array_equal(np.asarray(fused.counts), ec)
        assert fused == bnp.count_encoded(bnp.sequence.get_kmers(sequences, k), axis=None)


def test_reverse_complement(bnp):
    # tests/test_dna.py:18-20 (ASCII); 2-bit DNA and ragged rows, empty rows included
    assert str(bnp.sequence.get_reverse_complement(bnp.as_encoded_array("ACGTG"))) == "CACGT"
    dna = bnp.as_encoded_array("ACGTG", bnp.DNAEncoding)
    rc = bnp.sequence.get_reverse_complement(dna)
    assert rc.encoding == bnp.DNAEncoding and str(rc) == "CACGT"
    rows = ["ACGT", "", "A", "GGATTTC", "ACGTACGTACGTACGTACGTACGTACGTACGTACGTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTG"]
    for enc in (bnp.DNAEncoding, None):
        seqs = bnp.as_encoded_array(rows, enc) if enc is not None else bnp.as_encoded_array(rows)
        out = bnp.sequence.get_reverse_complement(seqs)
        comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
        assert out.tolist() == ["".join(comp[c] for c in reversed(r)) for r in rows]
        assert out.lengths.tolist() == [len(r) for r in rows]
        assert bnp.sequence.get_reverse_complement(out).tolist() == rows              # an involution
    # the other alphabets over A, C, G, T, N (dna.py:13-26: the complemented alphabet encoded with the alphabet itself)
    comp_n = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    rows_n = ["ACGNT", "", "N", "GGANNTTC", "ACGTN" * 9]
    for enc in (bnp.encodings.ACGTnEncoding, bnp.encodings.ACTGEncoding, bnp.encodings.ACTGnEncoding):
        use = [r.replace("N", "A") for r in rows_n] if len(enc.get_alphabet()) == 4 else rows_n
        seqs = bnp.as_encoded_array(use, enc)
        out = bnp.sequence.get_reverse_complement(seqs)
        assert out.encoding == enc and [x.upper() for x in out.tolist()] == ["".join(comp_n[c] for c in reversed(r)) for r in use]
        assert [x.upper() for x in bnp.sequence.get_reverse_complement(out).tolist()] == use
        one = bnp.sequence.get_reverse_complement(bnp.as_encoded_array(use[3], enc))
        assert str(one).upper() == "".join(comp_n[c] for c in reversed(use[3]))
    with pytest.raises(KeyError):                                                     # an alphabet with a letter that has no complement
        bnp.sequence.get_reverse_complement(bnp.as_encoded_array("ACUG", bnp.encodings.ACUGEncoding))
    # the ASCII table of the reference: N stays N, anything else becomes NUL (dna.py:29-33)
    odd = bnp.sequence.get_reverse_complement(bnp.as_encoded_array("ANxT"))
    assert np.asarray(odd.raw()).tolist() == [ord("A"), 0, ord("N"), ord("T")]
    # k-mers of the reverse complement are the reverse complements of the k-mers, in reverse order
    k = 5
    h = np.asarray(bnp.sequence.get_kmers(bnp.as_encoded_array(rows[-1], bnp.DNAEncoding), k).raw())
    h_rc = np.asarray(bnp.sequence.get_kmers(bnp.sequence.get_reverse_complement(
        bnp.as_encoded_array(rows[-1], bnp.DNAEncoding)), k).raw())
    assert np.array_equal(h_rc, oracle.reverse_complement_hash(h, k)[::-1])


def test_canonical_kmers_are_strand_independent(bnp):
    # extension: min(h, rc(h)); both strands of the same reads give the same histogram
    rng = np.random.default_rng(11)
    rows = ["".join(rng.choice(list("ACGT"), size=n)) for n in (60, 5, 31, 150, 33, 0, 90)]
    seqs = bnp.as_encoded_array(rows, bnp.DNAEncoding)
    rc = bnp.sequence.get_reverse_complement(seqs)
    for k in (3, 11, 31):
        fwd = np.asarray(bnp.sequence.get_kmers(seqs, k).raw().ravel())
        can = np.asarray(bnp.sequence.get_kmers(seqs, k, canonical=True).raw().ravel())
        assert np.array_equal(can, oracle.canonical_kmers(fwd, k))
        a = bnp.sequence.count_kmers(seqs, k, canonical=True)
        b = bnp.sequence.count_kmers(rc, k, canonical=True)
        assert a == b
        if k > 8:
            ek, ec = oracle.count_sparse(oracle.canonical_kmers(fwd, k))
            assert np.array_equal(a.keys, ek) and np.array_equal(a.counts, ec)


def test_filter_reads_on_base_quality_and_write_back(bnp, big_fq_gz, tmp_path):
    # scripts/small_example.py:36-46: keep = np.mean(chunk.quality, axis=1) > t; out_file.write(chunk[keep])
    import gzip
    text = gzip.open(big_fq_gz, "rb").read()
    res = oracle.scan_one_line_buffer(np.frombuffer(text, dtype=np.uint8), oracle.FASTQ)
    qual = oracle.quality_scores(oracle.gather_rows(np.frombuffer(text, dtype=np.uint8), res.field_starts[:, 3],
                                                    res.field_lens[:, 3]))
    sums, mins, maxs = oracle.row_reduce(qual, res.field_lens[:, 3])
    whole = bnp.open(big_fq_gz).read()
    assert np.array_equal(np.sum(whole.quality, axis=1), sums)
    assert np.array_equal(np.min(whole.quality, axis=1), mins) and np.array_equal(whole.quality.max(axis=-1), maxs)
    means = np.mean(whole.quality, axis=1)
    assert np.allclose(means, sums / res.field_lens[:, 3], rtol=0, atol=0)
    for name, threshold in (("a.fq", 8.0), ("b.fq.gz", 6.5)):
        out_name = str(tmp_path / name)
        kept = 0
        with bnp.open(out_name, "w") as out:
            for chunk in bnp.open(big_fq_gz).read_chunks(100000):
                keep = np.mean(chunk.quality, axis=1) > threshold
                out.write(chunk[keep])
                kept += int(keep.sum())
        expect_rows = np.flatnonzero(sums / res.field_lens[:, 3] > threshold)
        assert kept == expect_rows.size and 0 < kept < len(whole)
        data = np.frombuffer(text, dtype=np.uint8)
        expect = b"".join(data[res.entry_starts[r]:res.entry_ends[r]].tobytes() for r in expect_rows)
        got = (gzip.open(out_name, "rb") if name.endswith(".gz") else open(out_name, "rb")).read()
        assert got == expect
        assert bnp.count_entries(out_name) == kept
        again = bnp.open(out_name).read()
        assert again.sequence.tolist() == [whole.sequence[int(r)].to_string() for r in expect_rows[:50]] + \
            again.sequence.tolist()[50:]
    # mean base quality / match rate per read position (scripts/small_example.py:20-22,49-52)
    col_sums, col_counts = oracle.col_sums(qual, res.field_lens[:, 3])
    assert np.array_equal(np.sum(whole.quality, axis=0), col_sums)
    assert np.array_equal(np.mean(whole.quality, axis=0), col_sums / col_counts)
    matches = bnp.match_string(bnp.change_encoding(whole.sequence, bnp.DNAEncoding), "AC")
    per_base = np.mean(matches, axis=0)
    assert per_base.size == int(whole.sequence.lengths.max()) - 1 and 0 < per_base[0] < 1
    with pytest.raises(ValueError):
        np.min(bnp.encodings.QualityEncoding.encode(["II", "", "I"]), axis=1)


def test_match_string(bnp):
    # tests/test_string_matcher.py:53-65 and the docstring example of string_matcher.py:30-36
    seqs = bnp.as_encoded_array(["ACA", "TACTAC"], bnp.encodings.AlphabetEncoding("ACGT"))
    m = bnp.sequence.string_matcher.match_string(seqs, "AC")
    assert m.tolist() == [[True, False], [False, True, False, False, True]]
    ascii_seqs = bnp.as_encoded_array(["ACGT", "TACTAC"])
    m = bnp.match_string(ascii_seqs, bnp.as_encoded_array("AC", ascii_seqs.encoding))
    assert m.tolist() == [[True, False, False], [False, True, False, False, True]]
    assert np.sum(m, axis=1).tolist() == [1, 2]                       # matches per read (scripts/small_example.py:13-18)
    # rows shorter than the pattern, case sensitivity on text, a longer random case against the oracle
    short = bnp.match_string(bnp.as_encoded_array(["A", "", "ACGTA", "acgta"]), "CGT")
    assert short.tolist() == [[], [], [False, True, False], [False, False, False]]
    rng = np.random.default_rng(3)
    rows = ["".join(rng.choice(list("ACGT"), size=n)) for n in (500, 3, 0, 64, 31, 7000)]
    for enc in (bnp.DNAEncoding, None):
        seqs = bnp.as_encoded_array(rows, enc) if enc is not None else bnp.as_encoded_array(rows)
        for pattern in ("A", "GAT", "ACGTACGTACGTACGTACGTACGTACGTACG"):
            got = bnp.match_string(seqs, pattern)
            flat = np.frombuffer("".join(rows).encode(), dtype=np.uint8)
            hit, lens = oracle.match_string(flat, [len(r) for r in rows], np.frombuffer(pattern.encode(), dtype=np.uint8))
            assert got.lengths.tolist() == lens.tolist()
            assert np.array_equal(np.asarray(got.ravel()).astype(np.uint8), hit)
    assert bnp.match_string(bnp.as_encoded_array("TACTAC", bnp.DNAEncoding), "AC").tolist() == \
        [False, True, False, False, True]
    # a pattern of more than 64 symbols (more than one launch of the byte kernel compares): matched piece by piece
    long_rows = ["".join(rng.choice(list("ACGT"), size=n)) for n in (400, 100, 150, 0)]
    pattern = long_rows[0][37:37 + 150]
    long_rows[2] = pattern                                            # a row that IS the pattern; one that is shorter
    for enc in (bnp.DNAEncoding, None):
        seqs = bnp.as_encoded_array(long_rows, enc) if enc is not None else bnp.as_encoded_array(long_rows)
        got = bnp.match_string(seqs, pattern)
        flat = np.frombuffer("".join(long_rows).encode(), dtype=np.uint8)
        hit, lens = oracle.match_string(flat, [len(r) for r in long_rows], np.frombuffer(pattern.encode(), dtype=np.uint8))
        assert got.lengths.tolist() == lens.tolist() == [251, 0, 1, 0]
        assert np.array_equal(np.asarray(got.ravel()).astype(np.uint8), hit) and int(hit.sum()) == 2


def test_motif_scores(bnp):
    # docstring of get_motif_scores (position_weight_matrix.py:186-192) and tests/test_position_weight_matrix.py
    PWM = bnp.sequence.position_weight_matrix.PWM
    pwm = PWM.from_dict({"A": [5, 1], "C": [1, 5], "G": [0, 0], "T": [0, 0]})
    scores = bnp.get_motif_scores(bnp.as_encoded_array(["ACTGAC", "CA", "GG"]), pwm)
    with np.errstate(divide="ignore"):
        e = np.log(5 * 5 * 16.0)
        assert [r.tolist() for r in scores] == [[np.log(20.0) + np.log(20.0), -np.inf, -np.inf, -np.inf,
                                                np.log(20.0) + np.log(20.0)], [np.log(4.0) + np.log(4.0)], [-np.inf]]
    assert abs(scores[0][0] - 5.99146455) < 1e-8 and abs(scores[1][0] - 2.77258872) < 1e-8
    a_only = PWM.from_dict({"A": [1.0, 1.0], "C": [0.0, 0.0], "G": [0.0, 0.0], "T": [0.0, 0.0]})
    got = bnp.get_motif_scores(bnp.as_encoded_array("AAC", bnp.DNAEncoding), a_only)      # test_a_motifs, trimmed
    assert got.tolist() == [np.log(4 ** 2), -np.inf]
    matrix = np.log([[0.4, 0.25], [0.1, 0.25], [0.4, 0.25], [0.1, 0.25]])                   # the tests' fixture
    pwm = PWM(matrix, "ACGT")
    assert np.allclose(np.exp(pwm.calculate_score(bnp.as_encoded_array("AC", bnp.DNAEncoding))), 0.4 * 0.25)
    assert np.allclose(np.exp(bnp.get_motif_scores(bnp.as_encoded_array("ACGT", bnp.DNAEncoding), pwm)),
                       [0.4 * 0.25, 0.025, 0.4 * 0.25])
    # random ragged reads, a 12-column matrix: bit-identical to the oracle's accumulation order
    rng = np.random.default_rng(8)
    rows = ["".join(rng.choice(list("ACGT"), size=n)) for n in (300, 11, 12, 0, 13, 5000)]
    m = np.log(rng.dirichlet(np.ones(4), size=12).T / 0.25)
    got = bnp.get_motif_scores(bnp.as_encoded_array(rows, bnp.DNAEncoding), PWM(m, "ACGT"))
    codes = np.frombuffer("".join(rows).encode(), dtype=np.uint8)
    codes = np.searchsorted(np.frombuffer(b"ACGT", dtype=np.uint8), codes)
    expect, lens = oracle.pwm_scores(codes, [len(r) for r in rows], m)
    assert got.lengths.tolist() == lens.tolist() and np.array_equal(np.asarray(got.ravel()), expect)


def test_memory_mapped_encoded_reads(bnp, big_fq_gz, tmp_path):
    # streams/memory_mapping.py:10-90: decode once, cache as data.dat / lengths.dat / encoding.pkl, load without text
    base = str(tmp_path / "reads")

    def loader():
        for chunk in bnp.open(big_fq_gz).read_chunks(100000):
            yield bnp.change_encoding(chunk.sequence, bnp.DNAEncoding)

    with pytest.warns(FutureWarning):
        created = bnp.MemMapEncodedRaggedArray.create(loader, base)
    whole = bnp.change_encoding(bnp.open(big_fq_gz).read().sequence, bnp.DNAEncoding)
    lengths = np.memmap(base + "_lengths.dat", dtype=np.int32, mode="r")
    data = np.memmap(base + "_data.dat", dtype=np.uint8, mode="r")
    assert np.array_equal(lengths, whole.lengths) and data.size == int(whole.lengths.sum()) and data.max() <= 3
    loaded = bnp.MemMapEncodedRaggedArray.load(base)
    assert loaded.encoding == bnp.DNAEncoding and np.array_equal(loaded.lengths, whole.lengths)
    assert loaded.tolist()[:20] == whole.tolist()[:20] and created.tolist()[-5:] == whole.tolist()[-5:]
    assert bnp.sequence.count_kmers(loaded, 31) == bnp.sequence.count_kmers(whole, 31)


def test_reverse_complement_a_file(bnp, big_fq_gz, tmp_path):
    # scripts/reverse_compliment_example.py:4-15: rc = get_reverse_complement(chunk.sequence);
    # outfile.write(bnp.replace(chunk, sequence=rc)) — the text is rebuilt from the fields (join_fields)
    import gzip
    text = np.frombuffer(gzip.open(big_fq_gz, "rb").read(), dtype=np.uint8)
    res = oracle.scan_one_line_buffer(text, oracle.FASTQ)
    for name in ("rc.fq", "rc.fq.gz"):
        out_name = str(tmp_path / name)
        with bnp.open(out_name, "w") as out:
            for chunk in bnp.open(big_fq_gz).read_chunks(100000):
                rc = bnp.sequence.get_reverse_complement(chunk.sequence)
                out.write(bnp.replace(chunk, sequence=rc))
        assert bnp.count_entries(out_name) == bnp.count_entries(big_fq_gz)
        got = np.frombuffer((gzip.open(out_name, "rb") if name.endswith(".gz") else open(out_name, "rb")).read(),
                            dtype=np.uint8)
        fields = []
        for i in range(4):
            flat = oracle.gather_rows(text, res.field_starts[:, i], res.field_lens[:, i])
            if i == 1:
                flat = oracle.reverse_complement(flat, res.field_lens[:, i], ascii_bytes=True)
            if i == 2:                                              # the reference writes a bare "+" line
                flat, lens = np.full(res.n_records, ord("+"), dtype=np.uint8), np.ones(res.n_records, dtype=np.int64)
            else:
                lens = res.field_lens[:, i]
            fields.append((flat, lens))
        expect = oracle.join_fields(fields, ord("@"), (1, 0, 0, 0))
        assert np.array_equal(got, expect)
    back = bnp.open(str(tmp_path / "rc.fq")).read()
    whole = bnp.open(big_fq_gz).read()
    assert bnp.sequence.get_reverse_complement(back.sequence).tolist()[:30] == whole.sequence.tolist()[:30]
    assert np.array_equal(np.asarray(back.quality.ravel()), np.asarray(whole.quality.ravel()))
    # a DNA-encoded sequence column is decoded on the way out; two-line FASTA
    fa = str(tmp_path / "x.fa")
    with bnp.open(fa, "w", buffer_type=bnp.TwoLineFastaBuffer) as out:
        out.write(bnp.SequenceEntry(bnp.as_encoded_array(["r1", "read2"]),
                                    bnp.as_encoded_array(["ACGT", "GGA"], bnp.DNAEncoding)))
    assert open(fa, "rb").read() == b">r1\nACGT\n>read2\nGGA\n"


def test_fused_minimizer_pipeline_equals_the_api_path(bnp, big_fq_gz):
    # BASELINE config 3: the pipeline form (fused decode) gives exactly get_minimizers of the decoded reads
    import gzip
    from bionumpy_amd.pipeline import fastq_minimizers
    from bionumpy_amd.device import HArray
    text = np.frombuffer(gzip.open(big_fq_gz, "rb").read(), dtype=np.uint8)
    whole = bnp.open(big_fq_gz).read()
    seqs = bnp.change_encoding(whole.sequence, bnp.DNAEncoding)
    for k, w in ((31, 40), (5, 5), (12, 30), (8, 60)):                # (8, 60): 53 k-mers per window — the row-lookup kernel's
        got, stats = fastq_minimizers(HArray(host=text.copy()), k, w)
        expect = np.asarray(bnp.get_minimizers(seqs, k, w).raw().ravel())
        assert stats.n_reads == len(whole) and stats.n_kmers == expect.size
        assert np.array_equal(got.host(), expect)


def test_canonical_counts_on_every_path_of_the_pipeline(bnp, big_fq_gz):
    """pipeline.fastq_kmer_histogram(canonical=True): dense (k <= 13), sparse fused and sparse through the field tables give
    the histogram of min(h, hash of the reverse complement) — extension SURVEY 8f-1"""
    import gzip
    from bionumpy_amd.pipeline import fastq_kmer_histogram
    from bionumpy_amd.device import HArray
    text = np.frombuffer(gzip.open(big_fq_gz, "rb").read(), dtype=np.uint8)
    res = oracle.scan_one_line_buffer(text, oracle.FASTQ)
    codes = oracle.encode_dna(oracle.gather_rows(text, res.field_starts[:, 1], res.field_lens[:, 1]))
    for k, fused in ((5, True), (11, True), (15, True), (15, False), (31, False)):
        h, _ = oracle.get_kmers(codes, res.field_lens[:, 1], k)
        can = oracle.canonical_kmers(h, k)
        hist, stats = fastq_kmer_histogram(HArray(host=text.copy()), k, fused=fused, canonical=True)
        assert stats.n_kmers == h.size
        if k <= 13:
            assert np.array_equal(hist.host(), oracle.count_dense(can, k))
        else:
            ek, ec = oracle.count_sparse(can)
            assert np.array_equal(hist[0].host(), ek) and np.array_equal(hist[1].host(), ec)


def test_streamed_counts_equal_whole_file(bnp, big_fq_gz):
    # scripts/kmer_counting_example.py:4-17: sum of per-chunk counts; k=31 through the sparse extension
    whole = bnp.open(big_fq_gz).read()
    for k in (3, 5):
        streamed = bnp.sequence.count_kmers(bnp.open(big_fq_gz).read_chunks(100000).sequence, k)
        assert streamed == bnp.sequence.count_kmers(whole.sequence, k)
    c3 = bnp.sequence.count_kmers(whole.sequence, 3)
    assert c3["AAA"] == 3920 and c3["ACT"] == 3038 and int(c3.counts.sum()) == 215598
    sparse = bnp.sequence.count_kmers(bnp.open(big_fq_gz).read_chunks(100000).sequence, 31)
    sparse_whole = bnp.sequence.count_kmers(whole.sequence, 31)
    assert sparse == sparse_whole
    assert len(sparse) == 168493 and int(sparse.counts.max()) == 65 and int(sparse.counts.sum()) == 187598
    assert sparse.keys[:3].tolist() == [3848617838, 15394471354, 61577885419]
    assert sparse["CGGTAGCCAGCTGCGTTCAGTATGGAAGATT"] >= 1


def test_chunk_histograms_keep_the_reads_until_somebody_looks(bnp, big_fq_gz):
    """the reference's loop — sum(count_kmers(chunk.sequence, k) for chunk in reader), scripts/kmer_counting_example.py:4-17 — adds
    up histograms that have counted nothing yet: each holds its chunk's 2-bit reads (count_encoded.PendingReads), the sum holds all
    of them, and the first look counts them in one pass (the chunks cut at whole mask words: row counts / lengths that are no
    multiple of 64 bases, an empty chunk, rows shorter than k) == the histogram of all reads at once == the oracle's"""
    from bionumpy_amd.sequence.count_encoded import PendingReads
    rng = np.random.default_rng(77)
    k = 21
    pieces = []
    for n_rows in (1, 37, 0, 300, 5):
        rows = ["".join(rng.choice(list("ACGT"), size=int(n))) for n in rng.integers(0, 180, size=n_rows)]
        pieces.append(rows)
    hists = [bnp.count_kmers(bnp.as_encoded_array(rows, bnp.DNAEncoding), k) for rows in pieces if rows]
    total = sum(hists)
    assert all(isinstance(p, PendingReads) for p in total._pending) and len(total._pending) == sum(1 for h in hists if h._pending)
    all_rows = [r for rows in pieces for r in rows]
    codes = np.searchsorted(np.frombuffer(b"ACGT", dtype=np.uint8), np.frombuffer("".join(all_rows).encode(), dtype=np.uint8))
    h, _ = oracle.get_kmers(codes.astype(np.uint8), np.array([len(r) for r in all_rows]), k)
    ek, ec = oracle.count_sparse(h)
    assert np.array_equal(total.keys, ek) and np.array_equal(total.counts, ec) and not total._pending
    assert total == bnp.count_kmers(bnp.as_encoded_array(all_rows, bnp.DNAEncoding), k)
    # mixed with counted histograms and with hashes that wait (count_encoded on k-mers): everything merges
    again = hists[0] + total + bnp.count_encoded(bnp.get_kmers(bnp.as_encoded_array(pieces[1], bnp.DNAEncoding), k), axis=None)
    h2, _ = oracle.get_kmers(codes.astype(np.uint8)[:sum(len(r) for r in pieces[0] + pieces[1])],
                             np.array([len(r) for r in pieces[0] + pieces[1]]), k)
    ek2, ec2 = oracle.count_sparse(np.concatenate([h, h2]))
    assert np.array_equal(again.keys, ek2) and np.array_equal(again.counts, ec2)
    # a file read in small chunks by hand (no coalescing): the same histogram as the whole file
    by_hand = sum(bnp.count_kmers(chunk.sequence, 31) for chunk in bnp.open(big_fq_gz).read_chunks(20000))
    assert by_hand == bnp.count_kmers(bnp.open(big_fq_gz).read().sequence, 31) and len(by_hand) == 168493


def test_quality_reductions_before_and_after_the_column_is_gathered(bnp):
    """np.sum / mean / min / max(chunk.quality, axis=1) straight from the chunk's text ([hip]: ragged.py _DeferredRows — the
    column is only gathered when somebody looks at the values) == the same on the gathered column == numpy on the bytes minus
    33, uint8 wrap-around included (a quality byte below '!' is not valid FASTQ, but it is what uint8 arithmetic gives)"""
    rng = np.random.default_rng(5)
    n = 700
    lens = rng.integers(1, 130, size=n)
    quals = [bytes(rng.integers(34, 127, size=l).astype(np.uint8)) for l in lens]
    quals[3] = b" " + quals[3][1:]                               # a byte below '!': (32 - 33) mod 256 = 255
    quals[4] = b"\x22" * len(quals[4])
    text = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, b"A" * l, q) for i, (l, q) in enumerate(zip(lens, quals)))
    expect = [(np.frombuffer(q, dtype=np.uint8) - np.uint8(33)) for q in quals]

    def check(q):
        assert np.array_equal(np.asarray(np.sum(q, axis=1)), [int(e.astype(np.int64).sum()) for e in expect])
        assert np.array_equal(np.asarray(np.min(q, axis=1)), [e.min() for e in expect])
        assert np.array_equal(np.asarray(np.max(q, axis=-1)), [e.max() for e in expect])
        assert np.allclose(np.asarray(np.mean(q, axis=1)), [e.astype(np.int64).sum() / e.size for e in expect], rtol=0, atol=0)

    chunk = _reader(bnp, text, bnp.FastQBuffer).read()
    q = chunk.quality
    check(q)                                                     # nothing has looked at the values yet
    assert q[3].tolist() == expect[3].tolist() and q.dtype == np.uint8
    assert [r.tolist() for r in q[10:13]] == [e.tolist() for e in expect[10:13]]
    check(q)                                                     # ... and now the column has been gathered
    assert q.tolist() == [e.tolist() for e in expect]
    keep = np.mean(chunk.quality, axis=1) > 40
    assert [r.tolist() for r in chunk[keep].quality] == [e.tolist() for e in expect if e.astype(np.int64).sum() / e.size > 40]


def test_filtering_rows(bnp, big_fq_gz):
    # scripts/small_example.py:26-32 style: boolean row mask on a chunk
    chunk = bnp.open(big_fq_gz).read()
    mask = chunk.sequence.lengths > 200
    sub = chunk[mask]
    assert len(sub) == int(mask.sum())
    assert np.array_equal(sub.sequence.lengths, chunk.sequence.lengths[mask])
    assert sub.sequence[0].to_string() == chunk.sequence[int(np.flatnonzero(mask)[0])].to_string()


# ------------------------------------------------------------------------------------ a read that ends exactly at the end of the file
def _records_through(bnp, monkeypatch, text, buffer_type, min_chunk_size, ahead, tmp_path):
    """entries seen by read_chunk / read_chunks (with the read-ahead of big batches forced on when `ahead`)"""
    from bionumpy_amd.io import parser
    p = tmp_path / ("f%d_%d.txt" % (min_chunk_size, int(ahead)))
    p.write_bytes(text)
    if ahead:
        monkeypatch.setattr(parser, "_BIG", 4)
        monkeypatch.setattr(parser, "_READ_AHEAD", True)
    reader = parser.NumpyFileReader(open(str(p), "rb"), buffer_type)
    names = []
    if ahead:
        for buff in reader.read_chunks(min_chunk_size):
            names += [str(n) for n in buff.get_data().name]
    else:
        while True:
            buff = reader.read_chunk(min_chunk_size)
            if buff is None:
                break
            names += [str(n) for n in buff.get_data().name]
    reader.close()
    return names


@pytest.mark.parametrize("ahead", [False, True])
def test_last_entry_when_the_file_size_is_a_multiple_of_the_chunk_size(bnp, monkeypatch, tmp_path, ahead):
    # bionumpy/io/parser.py:183-200: the reference seeks back, reads the tail again, finds it shorter than min_chunk_size,
    # terminates it ('\n', and '>' for multi-line FASTA) and parses it.  Multi-line FASTA ALWAYS carries a tail (its last
    # record ends at the next '>'), a FASTQ file without a trailing newline does when the last read fills its window.
    fasta = b"".join(b">r%d\nACGTACG\n" % i for i in range(10))                 # 120 bytes, 10 records
    assert len(fasta) == 120
    for size in (20, 24, 30, 40, 60, 120, 121, 7, 1000):
        got = _records_through(bnp, monkeypatch, fasta, bnp.io.MultiLineFastaBuffer, size, ahead, tmp_path)
        assert got == ["r%d" % i for i in range(10)], (size, got)
    fastq = b"".join(b"@q%d\nACGT\n+\nIIII\n" % i for i in range(10))[:-1]      # 159 bytes, no trailing newline
    assert len(fastq) == 159
    for size in (159, 53, 63, 160, 16, 3, 1000):
        got = _records_through(bnp, monkeypatch, fastq, bnp.io.FastQBuffer, size, ahead, tmp_path)
        assert got == ["q%d" % i for i in range(10)], (size, got)


# ------------------------------------------------------------------------------------ count_encoded(weights=...)
def test_count_encoded_with_weights(bnp):
    # bionumpy/sequence/count_encoded.py:166-187 against its statement-by-statement restatement (oracle.count_weighted).
    # Integer / bool weights are accumulated in int64: bit-exact.  Floating-point weights are added with atomics, in
    # another order than numpy's loop: equal up to the rounding of the additions — rtol 1e-12 of the bin's sum of |w|.
    rng = np.random.default_rng(11)
    seq = "".join(rng.choice(list("ACGT"), size=5000))
    values = bnp.as_encoded_array(seq, bnp.DNAEncoding)
    codes = np.array(["ACGT".index(c) for c in seq])
    for w in (rng.integers(0, 100, size=5000), rng.integers(-5, 5, size=5000).astype(np.int32), rng.random(5000) < 0.3):
        got = bnp.count_encoded(values, weights=w)
        expect = oracle.count_weighted(codes, w, 4)
        assert got.counts.dtype == expect.dtype == np.float64 and np.array_equal(got.counts, expect)
        assert got.alphabet == ["A", "C", "G", "T"] and got["G"] == expect[2]
        assert np.array_equal(bnp.count_encoded(values, weights=w, axis=None).counts, expect)
    wf = rng.normal(size=5000) * 10.0 ** rng.integers(-3, 6, size=5000)
    got, expect = bnp.count_encoded(values, weights=wf).counts, oracle.count_weighted(codes, wf, 4)
    scale = np.bincount(codes, weights=np.abs(wf), minlength=4)
    assert got.dtype == np.float64 and np.all(np.abs(got - expect) <= 1e-12 * scale)
    # 2-D weights: one histogram of the flat values per row of the weights; integer counts unless the weights are floats
    w2 = rng.integers(0, 7, size=(6, 5000))
    got, expect = bnp.count_encoded(values, weights=w2).counts, oracle.count_weighted(codes, w2, 4)
    assert got.shape == (6, 4) and np.issubdtype(got.dtype, np.integer) and np.array_equal(got, expect)
    w2f = rng.random((3, 5000))
    got, expect = bnp.count_encoded(values, weights=w2f).counts, oracle.count_weighted(codes, w2f, 4)
    assert got.dtype == np.float64 and np.allclose(got, expect, rtol=1e-12, atol=0)
    assert np.array_equal(bnp.count_encoded(values, weights=(w2 > 3)).counts, oracle.count_weighted(codes, w2 > 3, 4))
    # rows of values under the same 1-D weights ("for row in values")
    reads = ["".join(rng.choice(list("ACGT"), size=40)) for _ in range(25)]
    rows = bnp.as_encoded_array(reads, bnp.DNAEncoding)
    w1 = rng.integers(1, 9, size=40)
    mat = np.array([["ACGT".index(c) for c in r] for r in reads])
    got, expect = bnp.count_encoded(rows, weights=w1).counts, oracle.count_weighted(mat, w1, 4)
    assert got.shape == (25, 4) and got.dtype == np.float64 and np.array_equal(got, expect)
    assert np.array_equal(bnp.count_encoded(rows, weights=np.tile(w1, 25), axis=None).counts, oracle.count_weighted(mat, np.tile(w1, 25), 4, axis=None))
    # k-mers (dense alphabets of 4^k bins): every 3-mer weighted by its position
    kmers = bnp.sequence.get_kmers(values, 3)
    h = oracle.kmer_hashes_flat(codes.astype(np.uint8), 3)
    pos = np.arange(h.size)
    got = bnp.count_encoded(kmers, weights=pos)
    assert np.array_equal(got.counts, oracle.count_weighted(h, pos, 64)) and got["ACG"] == float(pos[h == 0 + 4 * 1 + 16 * 2].sum())
    # what numpy refuses is refused
    with pytest.raises(ValueError):
        bnp.count_encoded(values, weights=np.ones(4999))
    with pytest.raises(ValueError):
        bnp.count_encoded(bnp.as_encoded_array(["ACGT", "AC"], bnp.DNAEncoding), weights=np.ones(4))


def test_minimizers_of_any_alphabet(bnp):
    # get_minimizers accepts any AlphabetEncoding (sequence/minimizers.py:48-52): the rolling hash of KmerEncoder
    # (codes . alphabet_size ** arange(k), wrapping int64) and the smallest of each window as numpy compares int64
    rng = np.random.default_rng(12)
    for letters, k, window in (("ACGTN", 3, 6), ("ACGTN", 5, 5), ("ACDEFGHIKLMNPQRSTVWY", 4, 9), ("AB", 7, 20),
                               ("ACDEFGHIKLMNPQRSTVWY", 15, 17)):          # 20^15 wraps int64: negative hashes take part in the min
        enc = bnp.AlphabetEncoding(letters)
        reads = ["".join(rng.choice(list(letters), size=n)) for n in (40, 0, window - 1, window, 200, 3)]
        seqs = bnp.as_encoded_array(reads, enc)
        got = bnp.sequence.get_minimizers(seqs, k, window)
        codes = np.concatenate([np.array([letters.index(c) for c in r], dtype=np.uint8) for r in reads])
        lens = np.array([len(r) for r in reads])
        expect, new_lens = oracle.get_minimizers(codes, lens, k, window, len(letters))
        assert [len(r) for r in got] == new_lens.tolist() == [max(0, n - window + 1) for n in lens]
        assert np.array_equal(np.concatenate([np.asarray(r) for r in got.raw()]), expect), (letters, k, window)
        assert got.encoding == bnp.sequence.get_kmers(seqs, k).encoding
    # one sequence (an EncodedArray) gives an EncodedArray
    enc = bnp.AlphabetEncoding("ACGTN")
    one = bnp.sequence.get_minimizers(bnp.as_encoded_array("ACGTNNACGTNACG", enc), 2, 5)
    codes = np.array(["ACGTN".index(c) for c in "ACGTNNACGTNACG"], dtype=np.uint8)
    assert np.array_equal(one.raw(), oracle.get_minimizers(codes, np.array([14]), 2, 5, 5)[0]) and str(one[0]) == "AC"


def test_ragged_reductions_of_every_shape(bnp):
    """what npstructures' RaggedArray (the reference's base class, encoded_array.py:161) answers: row reductions of int64 and
    float64 rows (k-mer hashes: Minimizers.__call__ is kmer_hashes.raw().min(axis=-1), minimizers.py:15-17; motif scores),
    column reductions, axis=None, strided column slices — against numpy on the same rows"""
    from bionumpy_amd.ragged import RaggedArray
    rng = np.random.default_rng(13)
    lens = np.array([5, 1, 12, 3, 40, 2, 9])
    rows_i = [rng.integers(-(1 << 40), 1 << 61, size=n) for n in lens]
    rows_f = [rng.normal(size=n) * 10.0 ** rng.integers(-2, 4) for n in lens]
    rows_f[2][3] = np.nan                                                     # numpy's min / max propagate NaN
    for rows, exact in ((rows_i, True), (rows_f, False)):
        ra = RaggedArray(np.concatenate(rows), lens)
        for name, f in (("sum", np.sum), ("min", np.min), ("max", np.max), ("mean", np.mean)):
            got = np.asarray(f(ra, axis=-1))
            expect = np.array([f(r) for r in rows])
            if exact and name != "mean":
                assert got.dtype == expect.dtype and np.array_equal(got, expect), name
            else:                                                             # float sums: another order of additions
                assert np.allclose(got, expect, rtol=1e-12, atol=0, equal_nan=True), name
            assert np.asarray(getattr(ra, name)(axis=-1)).shape == (7,)
        flat = np.concatenate(rows)
        for name, f in (("sum", np.sum), ("min", np.min), ("max", np.max), ("mean", np.mean)):
            assert np.allclose(getattr(ra, name)(axis=None), f(flat), equal_nan=True), name
            assert np.allclose(f(ra), f(flat), equal_nan=True), name           # np.sum(ragged): axis=None
        # per column, over the rows that reach it
        width = lens.max()
        cols = [np.array([r[c] for r in rows if r.size > c]) for c in range(width)]
        assert np.allclose(ra.sum(axis=0), [c.sum() for c in cols], equal_nan=True)
        assert np.allclose(ra.mean(axis=0), [c.mean() for c in cols], equal_nan=True)
        if exact:
            assert np.array_equal(ra.min(axis=0), [c.min() for c in cols]) and np.array_equal(ra.max(axis=0), [c.max() for c in cols])
        # strided and reversed column slices
        for sl in (slice(None, None, 2), slice(1, None, 3), slice(None, None, -1), slice(-2, 0, -2)):
            sub = ra[:, sl]
            assert [np.asarray(r).tolist() for r in sub] == [r[sl].tolist() for r in rows] or not exact
            assert sub.lengths.tolist() == [r[sl].size for r in rows]
    with pytest.raises(ValueError):
        RaggedArray(np.arange(3), [3, 0]).min(axis=-1)
    for name in ("sum", "mean", "min", "max"):                                # a ragged array has two axes
        with pytest.raises(ValueError):
            getattr(RaggedArray(np.arange(3), [2, 1]), name)(axis=2)
    flags = RaggedArray([[True, False, True], [False, False], [True]])       # bool rows: per column, any / all of the rows that reach it
    assert np.array_equal(flags.max(axis=0), [True, False, True]) and np.array_equal(flags.min(axis=0), [False, False, True])
    # the k-mer use: the smallest hash of every read == what get_minimizers gives for one window per read
    seqs = bnp.as_encoded_array(["ACGTACGTAC", "TTTTGGGGCC", "GATTACAGAT"], bnp.DNAEncoding)
    kmers = bnp.sequence.get_kmers(seqs, 4)
    assert np.array_equal(np.asarray(kmers.raw().min(axis=-1)), bnp.sequence.get_minimizers(seqs, 4, 10).raw().ravel())


def test_reduction_methods_default_to_the_whole_array(bnp, big_fq_gz):
    """npstructures' RaggedArray methods take ``axis=None`` by default (one scalar), per row is ``axis=-1``: the reference's
    doctest docs_source/source/reading_files.rst:46-54 prints ``chunk.quality.mean()`` of six chunks of big.fq.gz; README.rst:38-42
    counts the G's; round 5 returned one value per row here"""
    from bionumpy_amd.ragged import RaggedArray
    means = [chunk.quality.mean() for chunk in bnp.open(big_fq_gz).read_chunks(min_chunk_size=100000)]
    assert [repr(float(m)) for m in means] == ["11.243155401311078", "11.799580504498538", "11.447879005326635", "11.753348856321052",
                                                "11.67464738973286", "12.154069194606311"]
    chunk = bnp.open(big_fq_gz).read_chunk(300000)
    assert (chunk.sequence == "G").sum() == 26898 and np.ndim((chunk.sequence == "G").sum()) == 0
    quality = chunk.quality                                                   # a view of the text: reduced where it lies
    flat = np.concatenate([np.asarray(r) for r in quality])
    assert quality.sum() == flat.sum() and quality.max() == flat.max() and quality.min() == flat.min()
    assert quality.mean() == flat.mean() and np.isclose(quality.std(), flat.std(), rtol=1e-12)
    assert np.asarray(quality.sum(axis=-1)).shape == (511,) and np.asarray(quality.mean(axis=-1)).shape == (511,)
    flags = chunk.sequence == "N"
    assert flags.any() is False or flags.any() == bool(np.any(chunk.sequence.ravel().raw() == ord("N")))
    assert (chunk.sequence == "G").any() and not (chunk.sequence == "G").all()
    rows = [np.arange(4), np.arange(2) + 10, np.arange(0)]
    ra = RaggedArray(np.concatenate(rows), [4, 2, 0])
    for name in ("sum", "mean", "min", "max", "std", "any", "all"):
        assert np.ndim(getattr(ra, name)()) == 0, name
        assert getattr(ra, name)() == getattr(np.concatenate(rows), name)(), name
    assert np.allclose(np.asarray(ra[:2].std(axis=-1)), [r.std() for r in rows[:2]])
    # and they print as the reference's do (sequences.rst:173-174; string_matcher.py:34-36)
    entries = bnp.open(os.path.join(os.path.dirname(big_fq_gz), "reads.fq")).read()
    assert repr((entries.sequence == "T").sum(axis=-1)) == "array([4, 0])"
    assert repr(bnp.match_string(bnp.as_encoded_array(["ACGT", "TACTAC"]), "AC")) == \
        "ragged_array([ True False False]\n[False  True False False  True])"
    assert str(entries).splitlines()[2] == "%25s%25s%25s" % ("headerishere", "CTTGTTGA", "[2 2 2 2 2 2 2 2]")


# ------------------------------------------------------------------------------------ results that stay on the device (round 5)
def _on_device_never_downloaded(h):
    """an HArray that lives in HBM and whose host copy nobody has asked for"""
    return getattr(h, "on_device", False) and getattr(h, "_np", None) is None


def test_window_flags_and_scores_are_reduced_where_they_are(bnp, request):
    """match_string / get_motif_scores return ragged arrays over DEVICE data (string_matcher.py:16-55 and
    position_weight_matrix.py:177-196 of the reference return ragged arrays the caller reduces): .any / .all / .sum / .max per
    row, np.any / np.count_nonzero, and the flat counts never download the flags"""
    rng = np.random.default_rng(12)
    rows = ["".join(rng.choice(list("ACGT"), size=int(n))) for n in rng.integers(0, 400, size=300)] + ["GATTACA", "", "GATTAC"]
    seqs = bnp.as_encoded_array(rows, bnp.DNAEncoding)
    pattern = "GAT"
    hits = bnp.match_string(seqs, pattern)
    expect = [[r[i:i + 3] == pattern for i in range(max(0, len(r) - 2))] for r in rows]
    any_rows, per_row, n_hits = hits.any(axis=-1), hits.sum(axis=-1), hits.sum(axis=None)
    assert hits._pending is not None, "per-row counts of 2-bit DNA come from the packed words: no flags are written for them"
    assert n_hits == sum(sum(e) for e in expect)
    all_rows = hits.all(axis=-1)
    if "hip" in request.node.name:
        assert _on_device_never_downloaded(hits._data) and _on_device_never_downloaded(any_rows.harray()), "the flags crossed PCIe"
    assert np.asarray(any_rows).tolist() == [any(e) for e in expect]
    assert np.asarray(all_rows).tolist() == [all(e) for e in expect]               # (an empty row: True, as np.all)
    assert np.asarray(per_row).tolist() == [sum(e) for e in expect]
    assert np.asarray(np.any(hits, axis=-1)).tolist() == [any(e) for e in expect]
    assert int(hits.sum(axis=None)) == int(np.count_nonzero(hits)) == sum(sum(e) for e in expect)
    assert bool(hits.any(axis=None)) and not bool(hits.all(axis=None))
    if "hip" in request.node.name:
        assert _on_device_never_downloaded(hits._data), "the flags crossed PCIe"
    assert hits.tolist() == expect                                                  # (looking at them downloads them: same values)
    # the reads that contain the pattern, selected by the device mask
    picked = seqs[np.asarray(any_rows)]
    assert picked.tolist() == [r for r, e in zip(rows, expect) if any(e)]
    # motif scores: per-row maximum on the device == the oracle's
    PWM = bnp.sequence.position_weight_matrix.PWM
    m = np.log(rng.dirichlet(np.ones(4), size=7).T / 0.25)
    long_rows = [r for r in rows if len(r) >= 7]
    scores = bnp.get_motif_scores(bnp.as_encoded_array(long_rows, bnp.DNAEncoding), PWM(m, "ACGT"))
    best = scores.max(axis=-1)
    codes = np.searchsorted(np.frombuffer(b"ACGT", dtype=np.uint8), np.frombuffer("".join(long_rows).encode(), dtype=np.uint8))
    flat, lens = oracle.pwm_scores(codes, [len(r) for r in long_rows], m)
    ends = np.cumsum(lens)
    assert np.array_equal(np.asarray(best), np.array([flat[e - n:e].max() for e, n in zip(ends, lens)]))
    if "hip" in request.node.name:
        assert _on_device_never_downloaded(scores._data), "the scores crossed PCIe"
    with pytest.raises(ValueError):
        bnp.get_motif_scores(bnp.as_encoded_array(rows, bnp.DNAEncoding), PWM(m, "ACGT")).max(axis=-1)    # rows without a window


def test_letter_counts_without_a_host_copy(bnp, big_fq_gz):
    """count_encoded over LETTERS (count_encoded.py:166-182) and README.rst:38-42's np.sum(chunk.sequence == "G") == 53686"""
    reads = bnp.open(big_fq_gz).read()
    assert int(np.sum(reads.sequence == "G")) == 53686
    dna = bnp.change_encoding(reads.sequence, bnp.DNAEncoding)
    text = oracle.open_text(big_fq_gz).read()
    raw, res = text
    starts, lens = res.field_starts[:, 1], res.field_lens[:, 1]
    codes = oracle.encode_dna(oracle.gather_rows(raw, starts, lens))
    flat = bnp.count_encoded(dna, axis=None)                                         # packed 2-bit words: popcounts
    assert flat.alphabet == ["A", "C", "G", "T"] and np.array_equal(flat.counts, np.bincount(codes, minlength=4))
    assert flat["G"] == 53686
    per_row = bnp.count_encoded(dna)                                                 # axis=-1: one histogram per read
    ends = np.cumsum(lens)
    expect = np.array([np.bincount(codes[e - n:e], minlength=4) for e, n in zip(ends, lens)])
    assert per_row.counts.shape == (len(lens), 4) and np.array_equal(per_row.counts, expect)
    # other alphabets: 5 letters (bytes, register counters), amino acids (21: LDS bins), np.bincount with minlength
    rng = np.random.default_rng(4)
    for enc, letters in ((bnp.ACGTnEncoding, "ACGTn"), (bnp.AminoAcidEncoding, "ACDEFGHIKLMNPQRSTVWY*")):
        rows = ["".join(rng.choice(list(letters), size=int(n))) for n in rng.integers(0, 90, size=200)]
        arr = bnp.as_encoded_array(rows, enc)
        got = bnp.count_encoded(arr, axis=None)
        assert [int(x) for x in got.counts] == ["".join(rows).count(c) for c in letters]      # (codes follow the alphabet's order)
        rows_got = bnp.count_encoded(arr)
        assert rows_got.counts.tolist() == [[r.count(c) for c in letters] for r in rows]
        flat_arr = arr.ravel()
        assert np.bincount(flat_arr, minlength=len(letters) + 3).tolist() == \
            np.bincount(np.asarray(flat_arr.raw()), minlength=len(letters) + 3).tolist()
