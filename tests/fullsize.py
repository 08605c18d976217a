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
        if mode == 0:
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
