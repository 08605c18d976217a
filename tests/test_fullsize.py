"""-m gpu: the hot path at sizes the oracle cannot (or can barely) reach, and BASELINE config 5 at its configuration.

* more than 2^32 base positions in one batch (the regression test for the 32-bit position overflow that
  scripts/debug_large.py chased in round 1): size-independent checks of tests/fullsize.py;
* BASELINE config 3 at its configuration: the minimizers (k = 31, window 40) of 30 M reads, every one of them;
* S-genome (reads of a fixed genome, every 31-mer ~60 times): the WHOLE histogram of 1 M reads against
  oracle.count_sparse (np.unique) — the duplicate-heavy path: general finishing kernel, pre-counted buckets;
* config 5: the 31-mer index of sacCer3 (12 Mbases, 17 records, multi-line FASTA) pair for pair against the oracle's
  restatement of KmerIndex.create_index (bionumpy/sequence/indexing/kmer_indexing.py:24-47), and the lookups of all
  31-mers of big.fq.gz against np.searchsorted / np.isin on the oracle's pairs (kmer_indexing.py:49-55).
"""
import os

import numpy as np
import pytest

import oracle
import fullsize
from bionumpy_amd.device import HArray

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ops():
    from bionumpy_amd import ops as ops_mod
    ops_mod.set_ops(None)
    return ops_mod.get_ops()


def test_histogram_of_more_than_2_pow_32_base_positions(ops):
    from bionumpy_amd.pipeline import fastq_kmer_histogram
    n_reads, read_len, k, seed = 30_000_000, 150, 31, 77
    assert n_reads * read_len > 1 << 32
    text = ops.synth_fastq(n_reads, read_len, seed, 0, 0, 0)
    (keys, counts), stats = fastq_kmer_histogram(text, k)
    assert stats.n_reads == n_reads and stats.n_bases == n_reads * read_len
    assert stats.n_kmers == n_reads * (read_len - k + 1)
    done = fullsize.check_histogram(ops, text, n_reads, read_len, k, seed, 0, 0, 0, keys, counts)
    assert done["kmers"] == stats.n_kmers and done["sampled_kmers_vs_oracle"] > 300_000


def test_config3_minimizers_at_full_size(ops):
    """BASELINE config 3 (k = 31, window_size = 40: w = 10 k-mers per window) on 30 M reads — 3.33e9 minimizers: the fused
    pipeline's output against the row-lookup kernel element for element, and against the oracle on sampled reads"""
    n_reads, read_len, k, window, seed = 30_000_000, 150, 31, 40, 20260925
    text = ops.synth_fastq(n_reads, read_len, seed, 0, 0, 0)
    done = fullsize.check_minimizers(ops, text, n_reads, read_len, k, window, seed, 0, 0)
    assert done["minimizers"] == n_reads * (read_len - window + 1) and done["sampled_minimizers_vs_oracle"] >= 3000 * 111 - 111


def test_chunk_objects_at_full_size(ops):
    """The API objects on a chunk of 10 M reads (3.2 GB of FASTQ, 1.5e9 bases: far beyond the shapes of tests/test_api.py and
    of the kernels' parity tests): size-independent properties — the reverse complement is an involution in all three forms
    (2-bit, ASCII, ASCII read where it lies in the text), the per-read mean quality straight from the text equals the one of
    the gathered column, the rewritten chunk parses back to the fields it was written from, the k-mers of the decoded reads
    are those of the fused pipeline — and sampled reads against the oracle."""
    import bionumpy_amd as bnp
    from bionumpy_amd import synth
    from bionumpy_amd.pipeline import fastq_kmer_histogram
    n_reads, read_len, seed = 10_000_000, 150, 99
    text = ops.synth_fastq(n_reads, read_len, seed, 0, 0, 0)
    chunk = bnp.SequenceEntryWithQuality._lazy(bnp.FastQBuffer.from_raw_buffer(text))
    import torch
    same = lambda a, b: bool(torch.equal(a.dev(), b.dev()))
    # reverse complement: packed, bytes, rows of the text
    dna = bnp.change_encoding(chunk.sequence, bnp.DNAEncoding)
    rc = bnp.get_reverse_complement(dna)
    from bionumpy_amd.encoded_array import packed_words
    assert same(packed_words(bnp.get_reverse_complement(rc)._data), packed_words(dna._data))
    rc_rows = bnp.get_reverse_complement(chunk.sequence)                       # (a view of the text: bnpk_reverse_complement_rows)
    seq_bytes = bnp.get_reverse_complement(rc_rows)                            # (compact now: bnpk_reverse_complement_bytes)
    assert same(ops.unpack_codes(packed_words(dna._data), dna.total(), to_ascii=True), seq_bytes._flat_data())
    assert same(ops.unpack_codes(packed_words(rc._data), rc.total(), to_ascii=True), rc_rows._flat_data())
    codes = synth.read_codes(n_reads, read_len, seed, 0, 0, 0)
    for r in (0, 1, n_reads // 2, n_reads - 1):
        assert rc[r].to_string() == "".join("TGCA"[c] for c in codes[r][::-1])
    # the quality column: reductions from the text == reductions of the gathered column
    q = chunk.quality
    means = np.mean(q, axis=1)
    mins = np.min(q, axis=1)
    assert getattr(q, "_pending", None) is not None                            # nothing has gathered it yet
    q._compact()
    assert same(np.mean(q, axis=1).harray(), means.harray()) and same(np.min(q, axis=1).harray(), mins.harray())
    # the rewritten chunk (names and qualities joined from the text, sequences replaced) parses back to the same fields
    out = bnp.FastQBuffer.from_data(bnp.replace(chunk, sequence=rc_rows))
    assert out.size == text.size
    back = bnp.SequenceEntryWithQuality._lazy(bnp.FastQBuffer.from_raw_buffer(out))
    assert same(bnp.get_reverse_complement(back.sequence)._flat_data(), seq_bytes._flat_data())
    bq = back.quality
    bq._compact()
    assert same(bq._flat_data(), q._flat_data())
    bn, cn = back.name, chunk.name
    bn._compact(); cn._compact()
    assert same(bn._flat_data(), cn._flat_data())
    # k-mers of the decoded reads == the fused pipeline's histogram
    (keys, counts), stats = fastq_kmer_histogram(text, 31)
    hist = bnp.count_kmers(dna, 31)
    assert same(HArray(dev=hist._keys.dev()), keys) and same(HArray(dev=hist._counts.dev()), counts)


def test_the_reference_loop_over_a_file_of_several_batches(ops, tmp_path):
    """scripts/kmer_counting_example.py:4-17 over a 0.5 GB FASTQ file (four device batches of 128 MB, a hundred chunks of 5 MB
    cut out of them by the windowed reader, the sequence column and the k-mers shared per batch): the sum of the per-chunk
    histograms, in the example's form, the library's form, the stream form and with 7-mers (dense counts), equals the
    histogram of the whole text; the chunks hold every entry once, in order"""
    import torch
    import bionumpy_amd as bnp
    from bionumpy_amd.pipeline import fastq_kmer_histogram
    n_reads, k = 1_600_000, 31
    text = ops.synth_fastq(n_reads, 150, 7, 1, 20_000_000, 0)
    path = str(tmp_path / "reads.fq")
    text.host().tofile(path)
    (keys, counts), _ = fastq_kmer_histogram(text, k)
    same = lambda a, b: bool(torch.equal(a.dev(), b.dev()))

    def per_chunk(chunk, kk):
        return bnp.count_encoded(bnp.get_kmers(bnp.as_encoded_array(chunk.sequence, bnp.DNAEncoding), k=kk), axis=None)
    n_chunks, n_entries, first_names = 0, 0, []
    def counted(kk):
        nonlocal n_chunks, n_entries
        for c in bnp.open(path).read_chunks():                               # (the reference's default: 5 000 000 bytes)
            n_chunks += 1
            n_entries += len(c)
            if n_chunks in (1, 40, 90):
                first_names.append(c.name[0].to_string())
            yield per_chunk(c, kk)
    total = sum(counted(k))
    assert n_entries == n_reads and 95 <= n_chunks <= 110
    assert first_names[0] == bnp.open(path).read_chunk(1000).name[0].to_string() and len(set(first_names)) == 3
    assert same(HArray(dev=total._keys.dev()), keys) and same(HArray(dev=total._counts.dev()), counts)
    library = None
    for c in bnp.open(path).read_chunks():
        h = bnp.count_kmers(c.sequence, k)
        library = h if library is None else library + h
    assert same(HArray(dev=library._keys.dev()), keys) and same(HArray(dev=library._counts.dev()), counts)
    streamed = bnp.count_kmers(bnp.open(path).read_chunks().sequence, k)
    assert same(HArray(dev=streamed._keys.dev()), keys) and same(HArray(dev=streamed._counts.dev()), counts)
    dense = sum(counted(7))
    whole7, _ = fastq_kmer_histogram(text, 7)
    assert np.array_equal(np.asarray(dense.counts).ravel(), whole7.host())


def test_filter_and_match_at_full_size(ops):
    """a read filter with write-back and match_string on 10 M reads, checked by what must hold at any size: the kept entries
    parse back to exactly the rows the mask selects (names, sequences, qualities, in order), and the windows that match a
    7-letter pattern are the 7-mers with the pattern's hash (two different kernels)"""
    import torch
    import bionumpy_amd as bnp
    n_reads, read_len, seed = 10_000_000, 150, 123
    text = ops.synth_fastq(n_reads, read_len, seed, 1, 50_000_000, 0)
    chunk = bnp.SequenceEntryWithQuality._lazy(bnp.FastQBuffer.from_raw_buffer(text))
    same = lambda a, b: bool(torch.equal(a.dev(), b.dev()))
    keep = np.mean(chunk.quality, axis=1) >= 0.0
    keep[::3] = False
    keep[5::7] = False
    n_keep = int(keep.sum())
    assert 0 < n_keep < n_reads
    kept = chunk[keep]
    back = bnp.SequenceEntryWithQuality._lazy(bnp.FastQBuffer.from_raw_buffer(kept.get_buffer().entry_bytes()))
    assert len(back) == n_keep
    for field in ("name", "sequence", "quality"):
        a, b = getattr(back, field), getattr(kept, field)
        a._compact(); b._compact()
        assert same(a._flat_data(), b._flat_data()) and same(a.offsets(), b.offsets()), field
    dna = bnp.change_encoding(chunk.sequence, bnp.DNAEncoding)
    hits = bnp.match_string(dna, "GATTACA")
    kmers = bnp.get_kmers(dna, 7)
    want = int(bnp.get_kmers(bnp.as_encoded_array("GATTACA", bnp.DNAEncoding), 7).raw()[0])
    flags = hits._flat_data().dev().to(torch.bool)
    assert flags.numel() == kmers.total() and bool(torch.equal(flags, kmers._flat_data().dev() == want))
    assert int(flags.sum().item()) > 1000


@pytest.mark.parametrize("canonical,n_reads,genome_len", [(False, 1_000_000, 2_000_000), (True, 300_000, 600_000)])
def test_genome_reads_whole_histogram_equals_the_oracle(ops, canonical, n_reads, genome_len):
    from bionumpy_amd.pipeline import fastq_kmer_histogram
    from bionumpy_amd import synth
    read_len, k, seed = 150, 31, 5
    text = ops.synth_fastq(n_reads, read_len, seed, 1, genome_len, 0)
    (keys, counts), stats = fastq_kmer_histogram(text, k, canonical=canonical)
    codes = synth.read_codes(n_reads, read_len, seed, 1, genome_len, 0)
    h, _ = oracle.get_kmers(codes.reshape(-1), np.full(n_reads, read_len, dtype=np.int64), k)
    if canonical:
        h = oracle.canonical_kmers(h, k)
    ek, ec = oracle.count_sparse(h)
    assert np.array_equal(keys.host(), ek) and np.array_equal(counts.host(), ec)
    assert ek.size < 2 * genome_len and int(ec.max()) > 20            # (duplicate-heavy indeed)


def _oracle_fasta(path):
    raw, res = oracle.open_text(path).read()
    codes = oracle.encode_dna(oracle.gather_rows(raw, res.line_starts, res.line_lens))
    return codes, res.seq_lens


@pytest.mark.parametrize("world,mode", [(2, 0), (8, 0), (3, 1), (8, 1)])
def test_virtual_ranks_equal_the_unsharded_histogram(ops, world, mode):
    """the N-GPU sparse path played by one GPU (pipeline.fastq_kmer_histogram_virtual_ranks): N shards of 5 M reads in
    total, k-mers partitioned by the send cuts on the device, the exchange replaced by its result, every rank's key
    range counted with count_sparse(key_range=...) — the concatenation over the ranks must be the unsharded histogram"""
    from bionumpy_amd.pipeline import fastq_kmer_histogram, fastq_kmer_histogram_virtual_ranks
    n_reads, read_len, k, seed, genome_len = 5_000_000, 150, 31, 21, 3_000_000
    per = -(-n_reads // world)
    texts = [ops.synth_fastq(min(per, n_reads - r * per), read_len, seed, mode, genome_len, r * per) for r in range(world)]
    hists, stats, received, plan = fastq_kmer_histogram_virtual_ranks(texts, k, with_plan=True)
    # (mode 1: a 3 Mbase genome covered 250 times — every rank holds few distinct k-mers and sends (key, count) runs)
    assert plan == ("keys" if mode == 0 else "counts")
    assert sum(s.n_reads for s in stats) == n_reads
    assert sum(received) == sum(s.n_kmers for s in stats) if plan == "keys" else sum(received) < sum(s.n_kmers for s in stats) // 4
    if mode == 0:
        assert max(received) < 1.3 * min(received)                          # uniform keys: balanced ranges
    whole = ops.synth_fastq(n_reads, read_len, seed, mode, genome_len, 0)
    (rk, rc), _ = fastq_kmer_histogram(whole, k)
    import torch
    keys = torch.cat([h[0].dev() for h in hists])
    counts = torch.cat([h[1].dev() for h in hists])
    assert keys.numel() == rk.size and bool((keys == rk.dev()).all()) and bool((counts == rc.dev()).all())
    for a, b in zip(hists[:-1], hists[1:]):                                 # the ranks' key ranges do not overlap
        assert a[0].size == 0 or b[0].size == 0 or int(a[0].dev()[-1]) < int(b[0].dev()[0])
    done = fullsize.check_histogram(ops, whole, n_reads, read_len, k, seed, mode, genome_len, 0,
                                    HArray(dev=keys), HArray(dev=counts))
    assert done["kmers"] == sum(s.n_kmers for s in stats)


def _pinned_copy(array):
    from bionumpy_amd.io.pinned import PinnedBuffer
    buf = PinnedBuffer(array.size + 64)
    buf.array[:array.size] = array
    return buf, buf.array[:array.size]


def test_host_fed_pipeline_equals_the_oracle_and_the_resident_path(ops):
    """text in page-locked host memory, streamed in chunks on the copy stream (bionumpy_amd/hostfed.py): ragged reads
    (chunk cuts at arbitrary record boundaries, padding between the chunks' 2-bit streams) against the oracle, and two
    synthetic batches back to back against the device-resident pipeline"""
    from bionumpy_amd.hostfed import HostFedCounter, cut_points
    from bionumpy_amd.pipeline import fastq_kmer_histogram
    from bionumpy_amd import synth
    k = 31
    rng = np.random.default_rng(11)
    lens = rng.integers(1, 400, size=30_000)
    recs = []
    for i, n in enumerate(lens):
        seq = "".join(rng.choice(list("ACGTacgt"), size=n))
        recs.append("@read%d with @ and + in the name\n%s\n+\n%s\n" % (i, seq, "".join(rng.choice(list("@+IJ5"), size=n))))
    ragged = np.frombuffer("".join(recs).encode(), dtype=np.uint8)
    keep1, text1 = _pinned_copy(ragged)
    bounds = set(np.cumsum([0] + [len(r) for r in recs]).tolist())
    cuts = cut_points(text1, 1 << 20)
    assert len(cuts) > 5 and all(c in bounds for c in cuts)
    counter = HostFedCounter(k, chunk_bytes=1 << 20, ring=3)
    (keys, counts), stats = list(counter.run([text1]))[0]
    res = oracle.scan_one_line_buffer(ragged, oracle.FASTQ)
    codes = oracle.encode_dna(oracle.gather_rows(ragged, res.field_starts[:, 1], res.field_lens[:, 1]))
    h, _ = oracle.get_kmers(codes, res.field_lens[:, 1], k)
    ek, ec = oracle.count_sparse(h)
    assert stats.n_reads == 30_000 and stats.n_bases == int(lens.sum()) and stats.n_kmers == h.size
    assert np.array_equal(keys.host(), ek) and np.array_equal(counts.host(), ec)
    assert counter.timing.chunks == len(cuts) and counter.timing.bytes == ragged.size
    # two batches back to back (the second one's copies overlap the first one's counting stage)
    batches = [synth.fastq_bytes(200_000, 150, 3, 0, 0, 0), synth.fastq_bytes(150_000, 150, 4, 1, 300_000, 0)]
    pinned = [_pinned_copy(b) for b in batches]
    counter = HostFedCounter(k, chunk_bytes=8 << 20, ring=4)
    got = list(counter.run([p[1] for p in pinned]))
    assert len(got) == 2
    for ((keys, counts), stats), text in zip(got, batches):
        (rk, rc), rstats = fastq_kmer_histogram(HArray(host=text), k)
        assert stats.n_reads == rstats.n_reads and stats.n_kmers == rstats.n_kmers and stats.n_bases == rstats.n_bases
        assert np.array_equal(keys.host(), rk.host()) and np.array_equal(counts.host(), rc.host())
    assert counter.timing.h2d_gb_per_s > 0 and 0.0 <= counter.timing.overlap_frac <= 1.0
    # a bad base is reported with the reference's exception and the reference's offset: the flat index of the byte among the
    # sequence bytes of the WHOLE batch (encodings/alphabet_encoding.py:37-45 on the gathered sequences), whichever chunk of
    # the batch it was uploaded in; a malformed record anywhere comes first, as in the reference, which validates the
    # buffer before it encodes
    from bionumpy_amd.exceptions import EncodingError, FormatException

    def oracle_offset(text):
        res = oracle.scan_one_line_buffer(text, oracle.FASTQ)
        with pytest.raises(oracle.EncodingError) as e:
            oracle.encode_dna(oracle.gather_rows(text, res.field_starts[:, 1], res.field_lens[:, 1]))
        return e.value.offset

    for read, pos in ((777, 20), (0, 0), (199_999, 149), (150_000, 75)):          # first chunk, very first base, last chunk, a middle one
        bad = batches[0].copy()
        bad[316 * read + 12 + pos] = ord("N")
        bad[316 * 199_999 + 12 + 149] = ord("n") if read != 199_999 else ord("N")  # a second, later one: the first is reported
        assert oracle_offset(bad) == 150 * read + pos
        kb, tb = _pinned_copy(bad)
        counter = HostFedCounter(k, chunk_bytes=8 << 20, ring=2)
        assert len(cut_points(tb, 8 << 20)) > 4
        with pytest.raises(EncodingError) as e:
            list(counter.run([tb]))
        assert e.value.offset == 150 * read + pos
        del counter
    bad = batches[0].copy()
    bad[316 * 5 + 12 + 3] = ord("N")                      # an invalid base early ...
    bad[316 * 190_000] = ord(">")                         # ... and a malformed header in the last chunk: the format error wins
    kb, tb = _pinned_copy(bad)
    with pytest.raises(FormatException) as e:
        list(HostFedCounter(k, chunk_bytes=8 << 20, ring=2).run([tb]))
    assert e.value.line_number == 4 * 190_000
    # the feeder thread ends with the run, however it ends (it holds the counter and its ring of chunk buffers)
    import threading
    assert not [th for th in threading.enumerate() if th.name.startswith("bnpk-feeder")]


def test_saccer3_decoded_on_the_device_equals_the_oracle(ops, tmp_path):
    """A13 at the size of config 5: the device line tables (bnpk_multiline_cut / _table) of the 12 Mbase genome against
    oracle.scan_multiline_fasta, in one piece and in 1 MB chunks, and the writer's round trip"""
    import bionumpy_amd as bnp
    path = os.path.join(GOLD, "sacCer3.fa.gz")
    raw, res = oracle.open_text(path).read()
    seq = oracle.gather_rows(raw, res.line_starts, res.line_lens)
    names = oracle.gather_rows(raw, res.header_starts, res.header_lens)
    whole = bnp.open(path).read()
    assert np.array_equal(whole.sequence.lengths, res.seq_lens) and np.array_equal(whole.name.lengths, res.header_lens)
    assert np.array_equal(np.asarray(whole.sequence.ravel()), seq) and np.array_equal(np.asarray(whole.name.ravel()), names)
    n, got = 0, []
    for chunk in bnp.open(path).read_chunks(min_chunk_size=1_000_000):
        n += len(chunk)
        got.append(np.asarray(chunk.sequence.ravel()))
    assert n == 17 and np.array_equal(np.concatenate(got), seq)
    out = tmp_path / "again.fa"
    with bnp.open(str(out), "w") as f:
        f.write(whole)
    assert np.array_equal(np.fromfile(str(out), dtype=np.uint8),
                          oracle.multiline_from_data(names, res.header_lens, seq, res.seq_lens))
    again = bnp.open(str(out)).read()
    assert again.name.tolist() == whole.name.tolist()
    assert np.array_equal(np.asarray(again.sequence.ravel()), seq)


def test_config5_saccer3_index_and_lookup(ops):
    import bionumpy_amd as bnp
    k = 31
    path = os.path.join(GOLD, "sacCer3.fa.gz")
    genome = bnp.open(path).read()
    seqs = bnp.change_encoding(genome.sequence, bnp.DNAEncoding)
    codes, lens = _oracle_fasta(path)
    assert len(genome) == 17 == lens.size and int(seqs.total()) == 12_157_105 == int(lens.sum())
    index = bnp.KmerIndex.create_index(seqs, k=k)
    eh, er = oracle.kmer_index_pairs(codes, lens, k)
    assert np.array_equal(index._keys.host(), eh) and np.array_equal(index._rows.host(), er)          # the whole index
    # lookups: all 31-mers of big.fq.gz (187 598 of them)
    reads = bnp.open(os.path.join(GOLD, "big.fq.gz")).read()
    q = bnp.get_kmers(bnp.change_encoding(reads.sequence, bnp.DNAEncoding), k)
    q._compact()
    qh = q._flat_data().host()
    assert qh.size == 187_598
    lo, hi = index.get_indices_batch(q._flat_data())
    assert np.array_equal(lo.host(), np.searchsorted(eh, qh, side="left"))
    assert np.array_equal(hi.host(), np.searchsorted(eh, qh, side="right"))
    assert np.array_equal(hi.host() > lo.host(), np.isin(qh, eh))
    # the scalar API on sampled present and absent k-mers (kmer_indexing.py:49-55: ndarray of rows, [] if unseen)
    rng = np.random.default_rng(5)
    ukeys, first = np.unique(eh, return_index=True)
    bounds = np.concatenate((first, [eh.size]))
    for i in rng.integers(0, ukeys.size, size=200):
        np.testing.assert_array_equal(index.get_indices(int(ukeys[i])), er[bounds[i]:bounds[i + 1]])
    multi = np.flatnonzero(np.diff(bounds) > 1)                          # k-mers shared by several chromosomes
    for i in multi[:50]:
        np.testing.assert_array_equal(index.get_indices(int(ukeys[i])), er[bounds[i]:bounds[i + 1]])
    for x in rng.integers(0, 1 << 62, size=200):
        if not np.isin(x, ukeys):
            assert index.get_indices(int(x)) == []
    text = "".join("ACGT"[c] for c in codes[1000:1000 + k])
    np.testing.assert_array_equal(index.get_indices(text), er[np.searchsorted(eh, oracle.kmer_from_string(text)):
                                                               np.searchsorted(eh, oracle.kmer_from_string(text), side="right")])
