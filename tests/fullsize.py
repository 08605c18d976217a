"""Size-independent parity checks of a FULL-SIZE sparse k-mer histogram of the synthetic FASTQ (test infrastructure:
used by bench.py after its timed region and by the -m gpu tests; the package never imports this).

The oracle cannot count 6e9 31-mers, so the (keys, counts) the device produced are checked through what can be
computed independently of the counting kernels:

* **sampled reads against the oracle**: the k-mers of reads sampled at the start, in the middle and at the end of
  the batch are computed by the numpy oracle from the generator's numpy twin (synth.read_codes — nothing the device
  produced is involved), looked up in the device's sorted keys (bnpk_search_sorted) and their counts compared:
  every sampled k-mer must be present; its count must equal its multiplicity among the sampled reads for
  S-uniform (where a k-mer repeating anywhere else has probability ~1e-9) and reach it for S-genome;
* **order-independent checksums** of the whole multiset of k-mers: sum(h), sum(h*h) and sum(mix(h)) modulo 2^64
  and the number of k-mers, taken once over (key, count) and once over the k-mers in READ order as the
  position-flat generator (bnpk_windows_flat — a different kernel from the partition path, itself pinned to the
  oracle in tests/test_gpu_parity.py) produces them slab by slab.  A lost, duplicated or corrupted key changes them;
* keys strictly increasing, counts positive.
"""
import numpy as np

import oracle
from bionumpy_amd import synth
from bionumpy_amd.device import HArray

_MIX = 0x9E3779B97F4A7C15 - (1 << 64)       # odd multiplier, as a wrapping int64


def _sums(t, h, weights=None):
    """(sum h, sum h^2, sum (h * M) ^ ((h * M) >> 29)) over int64 tensors, wrapping"""
    w = 1 if weights is None else weights
    m = h * _MIX
    m = m ^ (m >> 29)
    return (int((h * w).sum().item()), int((h * h * w).sum().item()), int((m * w).sum().item()))


def histogram_sums(keys, counts):
    """[n k-mers, sum h, sum h^2, sum mix(h)] over (key, count); also checks keys strictly increasing, counts > 0"""
    import torch
    kd, cd = keys.dev(), counts.dev()
    n_distinct = kd.numel()
    if n_distinct > 1:
        assert bool((kd[1:] > kd[:-1]).all().item()), "keys are not strictly increasing"
    assert bool((cd > 0).all().item()), "a count is not positive"
    hist = [0, 0, 0]
    step = 1 << 28
    for a in range(0, n_distinct, step):
        s = _sums(torch, kd[a:a + step], cd[a:a + step])
        hist = [(x + y) & ((1 << 64) - 1) for x, y in zip(hist, s)]
    return [int(cd.sum().item())] + hist


def reads_sums(ops, text, n_reads, read_len, k, canonical=False, slab_reads=2_000_000):
    """the same four numbers over the k-mers in READ order (bnpk_windows_flat, slab by slab)"""
    import torch
    rec = synth.record_bytes(read_len)
    assert text.size == n_reads * rec
    flat = [0, 0, 0]
    n_kmers = 0
    t = text.dev()
    for r0 in range(0, n_reads, slab_reads):
        r1 = min(n_reads, r0 + slab_reads)
        slab = HArray(dev=t[r0 * rec:r1 * rec])
        packed, ends, n, n_bases = ops.fastq_encode(slab, slab.size, 4, 1, ord("@"), True)
        starts, m = ops.kmer_starts_from_ends(ends, n_bases, k)
        h = ops.windows_from_mask(packed, starts, n_bases, m, k, k)
        if canonical:
            h = ops.canonical_kmers(h, k)
        s = _sums(torch, h.dev())
        flat = [(x + y) & ((1 << 64) - 1) for x, y in zip(flat, s)]
        n_kmers += m
        del packed, ends, starts, h, slab
    return [n_kmers] + flat


def sampled_reads_check(ops, keys, counts, n_reads, read_len, k, seed, mode, genome_len, first_read, canonical=False,
                        sample_reads=3000):
    """k-mers of reads sampled at the start / middle / end, from the oracle, looked up in the device histogram"""
    import torch
    kd, cd = keys.dev(), counts.dev()
    n_distinct = kd.numel()
    per = max(1, min(sample_reads, n_reads) // 3)
    firsts = sorted({0, max(0, n_reads // 2 - per // 2), max(0, n_reads - per)})
    n_sampled = 0
    for f in firsts:
        m = min(per, n_reads - f)
        codes = synth.read_codes(m, read_len, seed, mode, genome_len, first_read + f)
        lens = np.full(m, read_len, dtype=np.int64)
        h, _ = oracle.get_kmers(codes.reshape(-1), lens, k)
        if canonical:
            h = oracle.canonical_kmers(h, k)
        ek, ec = oracle.count_sparse(h)
        pos = ops.search_sorted(keys, HArray(host=ek)).dev()
        assert bool((pos < n_distinct).all().item()), "a sampled k-mer lies beyond the last key"
        assert bool((kd[pos] == torch.from_numpy(ek).to(kd.device)).all().item()), "a sampled k-mer is missing"
        got = cd[pos].cpu().numpy()
        if mode == 0 and 2 * k >= 56:                    # (random reads: a k-mer of >= 28 bases occurs once in the whole batch)
            assert np.array_equal(got, ec), "count of a sampled k-mer differs (reads %d..%d)" % (f, f + m)
        else:
            assert np.all(got >= ec), "count of a sampled k-mer is too small (reads %d..%d)" % (f, f + m)
        n_sampled += ek.size
    return n_sampled


def check_histogram(ops, text, n_reads, read_len, k, seed, mode, genome_len, first_read, keys, counts,
                    canonical=False, sample_reads=3000, slab_reads=2_000_000):
    """all of the above on one GPU; raises AssertionError on the first violated check"""
    hist = histogram_sums(keys, counts)
    flat = reads_sums(ops, text, n_reads, read_len, k, canonical, slab_reads)
    assert hist[0] == flat[0], "counts sum to %d, the reads hold %d k-mers" % (hist[0], flat[0])
    assert hist == flat, "checksums differ: histogram %s vs k-mers in read order %s" % (hist, flat)
    n_sampled = sampled_reads_check(ops, keys, counts, n_reads, read_len, k, seed, mode, genome_len, first_read,
                                    canonical, sample_reads)
    return {"kmers": flat[0], "distinct": keys.size, "sampled_kmers_vs_oracle": n_sampled,
            "checksums": ["count", "sum", "sum of squares", "sum of mixed"], "slabs": -(-n_reads // slab_reads)}


def check_minimizers(ops, text, n_reads, read_len, k, window_size, seed, mode, genome_len, first_read=0, sample_reads=3000,
                     slab_reads=2_000_000):
    """BASELINE config 3 at full size: the minimizers of every window of every read as the fused pipeline produces them
    (pipeline.fastq_minimizers: fused decode + bnpk_windows_flat, position-flat, no row lookups) against

    * **another kernel, element for element**: bnpk_minimizers (one lane per output, row lookups, windows of any width —
      itself pinned to the oracle in tests/test_gpu_parity.py) over the same reads, slab by slab;
    * **the oracle on sampled reads**: the minimizers of reads at the start, in the middle and at the end of the batch,
      computed by the numpy oracle (oracle.get_minimizers, sequence/minimizers.py:8-54) from the generator's numpy twin,
      against the same rows of the device's output.
    Returns what was checked; raises AssertionError on the first difference."""
    import torch
    from bionumpy_amd.pipeline import fastq_minimizers
    rec = synth.record_bytes(read_len)
    per_read = read_len - window_size + 1
    assert text.size == n_reads * rec and per_read > 0
    out, stats = fastq_minimizers(text, k, window_size)
    assert stats.n_reads == n_reads and stats.n_kmers == n_reads * per_read == out.size
    got = out.dev()
    t = text.dev()
    compared = 0
    for r0 in range(0, n_reads, slab_reads):
        r1 = min(n_reads, r0 + slab_reads)
        slab = HArray(dev=t[r0 * rec:r1 * rec])
        packed, ends, n, n_bases = ops.fastq_encode(slab, slab.size, 4, 1, ord("@"), True)
        m = r1 - r0
        in_off = HArray(dev=torch.arange(m + 1, dtype=torch.int64, device=got.device) * read_len)
        out_off = HArray(dev=torch.arange(m + 1, dtype=torch.int64, device=got.device) * per_read)
        by_rows = ops.minimizers_by_rows(packed, in_off, out_off, m, m * per_read, k, window_size).dev()
        assert torch.equal(by_rows, got[r0 * per_read:r1 * per_read]), "minimizers of reads %d..%d differ between the kernels" % (r0, r1)
        compared += m * per_read
        del packed, ends, slab, by_rows
    per = max(1, min(sample_reads, n_reads) // 3)
    sampled = 0
    for f in sorted({0, max(0, n_reads // 2 - per // 2), max(0, n_reads - per)}):
        m = min(per, n_reads - f)
        codes = synth.read_codes(m, read_len, seed, mode, genome_len, first_read + f)
        expect, lens = oracle.get_minimizers(codes.reshape(-1), np.full(m, read_len, dtype=np.int64), k, window_size)
        assert np.all(lens == per_read)
        assert np.array_equal(got[f * per_read:(f + m) * per_read].cpu().numpy(), expect), "minimizers of reads %d..%d differ from the oracle" % (f, f + m)
        sampled += expect.size
    return {"minimizers": compared, "compared_with": "bnpk_minimizers (row-lookup kernel), element for element",
            "sampled_minimizers_vs_oracle": sampled}
