"""Pins the CPU oracle to the reference's own golden vectors.

Each test names the reference file:line holding the expected values.
"""
import gzip
import hashlib
import io

import numpy as np
import pytest

import oracle
from oracle.text import FASTQ, TWO_LINE_FASTA


def _bytes(text):
    return np.frombuffer(text.encode("ascii"), dtype=np.uint8)


def _encode(strings):
    flat = _bytes("".join(strings))
    lengths = np.array([len(s) for s in strings], dtype=np.int64)
    return oracle.encode_dna(flat), lengths


def _rows(flat, lengths):
    starts = oracle.row_starts(lengths)
    return [flat[s:s + l] for s, l in zip(starts, lengths)]


@pytest.fixture(scope="module")
def big_reads(big_fq_gz):
    raw = np.frombuffer(gzip.open(big_fq_gz, "rb").read(), dtype=np.uint8)
    res = oracle.scan_one_line_buffer(raw, FASTQ)
    starts, lens = res.field_starts[:, 1], res.field_lens[:, 1]
    codes = oracle.encode_dna(oracle.gather_rows(raw, starts, lens))
    return raw, res, codes, lens


# ---------------------------------------------------------------- k-mers
def test_doc_31mers_of_big_fq(big_reads):
    # docs_source/topics/kmers.rst:73-78
    _, _, codes, lens = big_reads
    h, hl = oracle.get_kmers(codes, lens, 31)
    rows = _rows(h, hl)
    assert rows[0][:2].tolist() == [4360244785522956521, 4548825710201280058]
    assert rows[1][:2].tolist() == [3755975642940518834, 3244836919948823660]
    assert rows[2][:2].tolist() == [2804282287455632382, 3006913581077602047]
    assert h[:4].tolist() == [4360244785522956521, 4548825710201280058,
                              3443049436764013966, 860762359191003491]
    # docs_source/topics/kmers.rst:69-71 / bionumpy/sequence/kmers.py:65-66
    assert oracle.kmer_to_string(rows[0][0], 31) == "CGGTAGCCAGCTGCGTTCAGTATGGAAGATT"
    assert oracle.kmer_to_string(rows[0][2], 31) == "GTAGCCAGCTGCGTTCAGTATGGAAGATTTG"
    assert oracle.kmer_to_string(rows[2][1], 31) == "TTTTGTCGCTGCGTTCAGTTTATGGGTGCGG"


def test_survey_derived_goldens(big_reads):
    # SURVEY.md §8c "Derived goldens" (regression values, provenance = survey)
    raw, res, codes, lens = big_reads
    assert res.n_records == 1000 and codes.size == 217598
    assert lens.min() == 144 and lens.max() == 1558
    h3, _ = oracle.get_kmers(codes, lens, 3)
    c3 = oracle.count_dense(h3, 3)
    labels = oracle.kmer_labels(3)
    assert h3.size == 215598 and np.count_nonzero(c3) == 64
    assert c3[labels.index("AAA")] == 3920 and c3[labels.index("ACT")] == 3038
    h31, _ = oracle.get_kmers(codes, lens, 31)
    keys, counts = oracle.count_sparse(h31)
    assert h31.size == 187598 and keys.size == 168493 and counts.max() == 65
    assert keys[:3].tolist() == [3848617838, 15394471354, 61577885419]
    assert hashlib.sha256(h31.astype("<i8").tobytes()).hexdigest()[:16] == "8357fbfed9d5864a"
    m, ml = oracle.get_minimizers(codes, lens, 31, 40)
    assert m.size == 178598 and m[:4].tolist() == [860762359191003491] * 4
    assert hashlib.sha256(m.astype("<i8").tobytes()).hexdigest()[:16] == "655a1532b0b1dd12"


def test_readme_g_count(big_reads):
    # README.rst:38-42: number of G's in the first 1000 reads is 53686
    raw, res, codes, lens = big_reads
    assert int(np.count_nonzero(codes == 2)) == 53686


def test_doc_3mers_and_counts():
    # docs_source/topics/kmers.rst:11-19
    codes, lens = _encode(["ACTG", "GGGACT", "G"])
    h, hl = oracle.get_kmers(codes, lens, 3)
    got = [[oracle.kmer_to_string(x, 3) for x in r] for r in _rows(h, hl)]
    assert got == [["ACT", "CTG"], ["GGG", "GGA", "GAC", "ACT"], []]
    counts = oracle.count_dense(h, 3)
    assert counts[oracle.kmer_labels(3).index("ACT")] == 2


def test_docstring_kmers():
    # bionumpy/sequence/kmers.py:56-61
    codes, lens = _encode(["ACTG", "AAA", "TTGGC"])
    h, hl = oracle.get_kmers(codes, lens, 3)
    got = [[oracle.kmer_to_string(x, 3) for x in r] for r in _rows(h, hl)]
    assert got == [["ACT", "CTG"], ["AAA"], ["TTG", "TGG", "GGC"]]


def test_label_order_and_counts():
    # tests/test_kmer.py:85-94 (first base is least significant) and :97-102
    assert oracle.kmer_labels(3)[:5] == ["AAA", "CAA", "GAA", "TAA", "ACA"]
    codes, lens = _encode(["ACTG", "AAA", "TTGGC"])
    h, _ = oracle.get_kmers(codes, lens, 3)
    counts = oracle.count_dense(h, 3)
    labels = oracle.kmer_labels(3)
    assert counts[labels.index("ACT")] == 1 and counts[labels.index("GGG")] == 0


def test_fast_path_equals_generic_path():
    # tests/test_kmer.py:27-30 and :33-40 (ragged shape = lengths - k + 1)
    codes, lens = _encode(["cgtt"])
    assert np.array_equal(oracle.get_kmers(codes, lens, 3)[0], oracle.get_kmers_generic(codes, lens, 3)[0])
    lengths = np.arange(3, 10)
    codes = (np.arange(lengths.sum()) % 4).astype(np.uint8)
    for k in (1, 3, 5):
        a, al = oracle.get_kmers(codes, lengths, k)
        b, bl = oracle.get_kmers_generic(codes, lengths, k)
        assert np.array_equal(a, b) and np.array_equal(al, bl)
        assert np.array_equal(al, np.maximum(lengths - k + 1, 0))
    rng = np.random.default_rng(0)
    lengths = rng.integers(0, 80, size=50)
    codes = rng.integers(0, 4, size=lengths.sum()).astype(np.uint8)
    for k in (1, 2, 17, 31):
        a, al = oracle.get_kmers(codes, lengths, k)
        b, bl = oracle.get_kmers_generic(codes, lengths, k)
        assert np.array_equal(a, b) and np.array_equal(al, bl)


def test_k_one():
    # tests/test_kmer.py:60-63
    codes, lens = _encode(["ACTG"])
    h, hl = oracle.get_kmers(codes, lens, 1)
    assert hl.tolist() == [4] and h.tolist() == [0, 1, 3, 2]


# ---------------------------------------------------------------- minimizers
def test_minimizer_numeric_goldens():
    # tests/test_minimizers.py:44-62
    seq = np.array([0, 3, 1, 2, 2, 1, 0], dtype=np.uint8)
    m, ml = oracle.get_minimizers(seq[:4], [4], 2, 4)
    assert m.tolist() == [7]
    m, ml = oracle.get_minimizers(seq, [7], 2, 4)
    assert m.tolist() == [7, 7, 6, 1]
    rows = [seq, seq[:6], seq[:5], seq[:4]]
    m, ml = oracle.get_minimizers(np.concatenate(rows), [7, 6, 5, 4], 2, 4)
    assert [r.tolist() for r in _rows(m, ml)] == [[7, 7, 6, 1], [7, 7, 6], [7, 7], [7]]


def test_minimizer_string_goldens():
    # tests/test_minimizers.py:65-80
    codes, lens = _encode(["CCCAAACCCC", "TTTTCCCTTT"])
    m, ml = oracle.get_minimizers(codes, lens, 3, 10)
    assert [[oracle.kmer_to_string(x, 3) for x in r] for r in _rows(m, ml)] == [["AAA"], ["CCC"]]
    # bionumpy/sequence/minimizers.py:39-46
    codes, lens = _encode(["ACTG", "AAA", "TTGGC"])
    m, ml = oracle.get_minimizers(codes, lens, 2, 4)
    assert [[oracle.kmer_to_string(x, 2) for x in r] for r in _rows(m, ml)] == [["AC"], [], ["GG", "GC"]]
    # docs_source/topics/kmers.rst:24-27
    codes, lens = _encode(["ACTG", "GGGACT", "G"])
    m, ml = oracle.get_minimizers(codes, lens, 2, 4)
    assert [[oracle.kmer_to_string(x, 2) for x in r] for r in _rows(m, ml)] == [["AC"], ["GA", "GA", "GA"], []]


# ---------------------------------------------------------------- k-mer index
def test_kmer_index_goldens():
    # tests/test_kmer_index.py:12-28
    codes, lens = _encode(["ACGTAA", "GCTAAA"])
    idx = oracle.build_kmer_index(codes, lens, 3)
    assert idx[oracle.kmer_from_string("ACG")].tolist() == [0]
    assert idx[oracle.kmer_from_string("AAA")].tolist() == [1]
    assert idx[oracle.kmer_from_string("TAA")].tolist() == [0, 1]
    assert oracle.kmer_from_string("GAA") not in idx
    idx2 = oracle.build_kmer_index(codes, lens, 2)
    assert idx2[oracle.kmer_from_string("AA")].tolist() == [0, 1]
    assert idx[oracle.kmer_from_string("CGT")].tolist() == [0]       # -> get_sequences == ["ACGTAA"]


def test_debruijn_goldens():
    # tests/test_debruijn.py:16-35 (sequences 'acg', 'cgtc', k = 2)
    codes, lens = _encode(["ACG", "CGTC"])
    kmers, _ = oracle.get_kmers(codes, lens, 2)
    kmer_set = np.unique(kmers)
    fw = oracle.debruijn_neighbours(kmer_set, oracle.kmer_from_string("AC"), 2, forward=True)
    assert [oracle.kmer_to_string(x, 2) for x in fw] == ["CG"]
    bw = oracle.debruijn_neighbours(kmer_set, oracle.kmer_from_string("TC"), 2, forward=False)
    assert [oracle.kmer_to_string(x, 2) for x in bw] == ["GT"]
    colors = oracle.colored_debruijn(codes, lens, 2)
    assert colors[oracle.kmer_from_string("AC")] == [0]
    assert colors[oracle.kmer_from_string("TC")] == [1]
    assert colors[oracle.kmer_from_string("CG")] == [0, 1]


# ---------------------------------------------------------------- text decode
FASTQ_TEXT = "@headerishere\nCTTGTTGA\n+\n!!!!!!!!\n@anotherheader\nCGG\n+\n~~~\n"   # tests/buffers.py:17-25
FASTA_TEXT = ">header\nCTTGTTGA\n>header2\nCGG\n"                                       # tests/buffers.py:26-31
MULTILINE_TEXT = ">header\nCTTGCC\nGCCTCC\n>header2\nCCCCCC\nGGGCCC\nTTT\n"              # tests/buffers.py:32-40


def _fields(raw, res, i):
    return [bytes(raw[s:s + l]).decode() for s, l in zip(res.field_starts[:, i], res.field_lens[:, i])]


def test_fastq_fixture():
    # tests/buffers.py:104-106
    raw = _bytes(FASTQ_TEXT)
    res = oracle.scan_one_line_buffer(raw, FASTQ)
    assert _fields(raw, res, 0) == ["headerishere", "anotherheader"]
    assert _fields(raw, res, 1) == ["CTTGTTGA", "CGG"]
    assert _fields(raw, res, 3) == ["!!!!!!!!", "~~~"]
    q = oracle.quality_scores(oracle.gather_rows(raw, res.field_starts[:, 3], res.field_lens[:, 3]))
    assert q.tolist() == [0] * 8 + [93] * 3


def test_two_line_fasta_fixture():
    # tests/buffers.py:107-109
    raw = _bytes(FASTA_TEXT)
    res = oracle.scan_one_line_buffer(raw, TWO_LINE_FASTA)
    assert _fields(raw, res, 0) == ["header", "header2"]
    assert _fields(raw, res, 1) == ["CTTGTTGA", "CGG"]


def test_multiline_fasta_fixture():
    # tests/buffers.py:110-112
    raw = _bytes(MULTILINE_TEXT + ">")
    res = oracle.scan_multiline_fasta(raw)
    heads = [bytes(raw[s:s + l]).decode() for s, l in zip(res.header_starts, res.header_lens)]
    assert heads == ["header", "header2"]
    seq = bytes(oracle.gather_rows(raw, res.line_starts, res.line_lens)).decode()
    assert res.seq_lens.tolist() == [12, 15]
    assert seq == "CTTGCCGCCTCC" + "CCCCCCGGGCCCTTT"


def test_carriage_return():
    # tests/test_io.py:233-249 (\r\n line ends are stripped from fields)
    raw = _bytes(FASTQ_TEXT.replace("\n", "\r\n"))
    res = oracle.scan_one_line_buffer(raw, FASTQ)
    assert _fields(raw, res, 0) == ["headerishere", "anotherheader"]
    assert _fields(raw, res, 1) == ["CTTGTTGA", "CGG"]


MALFORMED = [("@header\nactg\n-\n!!!!\n", 2),                                   # tests/test_io_exceptions.py:11-33
             ("header\nactg\n+\n!!!!\n", 0),
             ("@header\nactg\n+\n@header\nactg\n+\n@header\nactg\n+\n", 4)]


@pytest.mark.parametrize("text,line", MALFORMED)
def test_malformed_fastq_line_numbers(text, line):
    with pytest.raises(oracle.FormatException) as e:
        oracle.scan_one_line_buffer(_bytes(text), FASTQ)
    assert e.value.line_number == line


def test_malformed_two_line_fasta():
    # tests/test_io_exceptions.py:35-41,66-73
    with pytest.raises(oracle.FormatException) as e:
        oracle.scan_one_line_buffer(_bytes(">header\nacggtt\nacggtt\n>header\nacgtt\n"), TWO_LINE_FASTA)
    assert e.value.line_number == 2


@pytest.mark.parametrize("text,line", MALFORMED)
def test_line_numbers_accumulate_over_chunks(text, line):
    # tests/test_io_exceptions.py:86-100
    valid = "@header\nacgtt\n+\n!!!!!\n"
    reader = oracle.ChunkReader(io.BytesIO((valid * 100 + text).encode()), FASTQ)
    with pytest.raises(oracle.FormatException) as e:
        for _ in reader.read_chunks(200):
            pass
    assert e.value.line_number == 4 * 100 + line


def test_chunked_reading_equals_whole(big_fq_gz):
    # bionumpy/io/files.py:117-162: read_chunk(300000) on big.fq.gz gives 511 then 489 entries
    reader = oracle.open_text(big_fq_gz)
    sizes = [res.n_records for _, res in reader.read_chunks(300000)]
    assert sizes == [511, 489]
    # tests/test_io.py:95-118: tiny chunks give the same entries as one read
    whole_raw, whole = oracle.open_text(big_fq_gz).read()
    seqs_whole = oracle.gather_rows(whole_raw, whole.field_starts[:, 1], whole.field_lens[:, 1])
    parts = []
    for raw, res in oracle.open_text(big_fq_gz).read_chunks(1000):
        parts.append(oracle.gather_rows(raw, res.field_starts[:, 1], res.field_lens[:, 1]))
    assert np.array_equal(np.concatenate(parts), seqs_whole)


def test_missing_final_newline_and_plain_file_seek(tmp_path):
    # io/parser.py:183-190 (newline appended at EOF only), :160-165 (seek back on plain files)
    p = tmp_path / "x.fq"
    p.write_bytes((FASTQ_TEXT * 7)[:-1].encode())
    got = []
    for raw, res in oracle.open_text(str(p)).read_chunks(50):
        got += _fields(raw, res, 1)
    assert got == ["CTTGTTGA", "CGG"] * 7


# ---------------------------------------------------------------- encoding
def test_encoding_goldens():
    # docs_source/source/encoding.rst:15-19,40-42; tests/test_encodings.py:30-41
    assert oracle.encode_dna(_bytes("ACGT")).tolist() == [0, 1, 2, 3]
    assert oracle.encode_dna(_bytes("acgt")).tolist() == [0, 1, 2, 3]
    assert bytes(oracle.decode_dna(oracle.encode_dna(_bytes("AacGt")))).decode() == "AACGT"
    with pytest.raises(oracle.EncodingError) as e:
        oracle.encode_dna(_bytes("ACGNNT"))
    assert e.value.offset == 3
    # tests/test_encodings.py:118-122: quality "!#" -> [0, 2]
    assert oracle.quality_scores(_bytes("!#")).tolist() == [0, 2]


def test_sparse_merge_equals_global_unique():
    rng = np.random.default_rng(3)
    h = rng.integers(0, 500, size=5000).astype(np.int64)
    parts = [oracle.count_sparse(h[i:i + 700]) for i in range(0, 5000, 700)]
    k, c = oracle.merge_sparse(parts)
    k2, c2 = oracle.count_sparse(h)
    assert np.array_equal(k, k2) and np.array_equal(c, c2)
