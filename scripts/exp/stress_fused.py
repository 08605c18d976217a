"""Stress: the fused pipeline (decode -> k-mer generation fused with the first partition level -> levels -> finish) against
the position-flat k-mer kernel + torch.unique, over k, canonical, read counts, genome sizes (duplicates) and read lengths."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bionumpy_amd.ops import get_ops
from bionumpy_amd.pipeline import fastq_kmer_histogram
ops = get_ops()
bad = n_cases = 0
for k in (14, 16, 21, 27, 31):
    for canonical in (False, True):
        for reads, read_len, mode, glen in ((3000, 150, 0, 0), (400_000, 97, 1, 50_000), (2_000_000, 150, 1, 3_000_000), (5_000_000, 64, 0, 0)):
            text = ops.synth_fastq(reads, read_len, 11 + k, mode, glen, 0)
            (keys, counts), st = fastq_kmer_histogram(text, k, canonical=canonical)
            packed, ends, n, n_bases = ops.fastq_encode(text, text.size, 4, 1, ord("@"), True)
            starts, n_kmers = ops.kmer_starts_from_ends(ends, n_bases, k)
            flat = ops.windows_from_mask(packed, starts, n_bases, n_kmers, k, k)
            if canonical:
                flat = ops.canonical_kmers(flat, k)
            flat = flat.dev()
            ek, ec = torch.unique(flat, return_counts=True)
            ok = keys.size == ek.numel() and bool((keys.dev() == ek).all()) and bool((counts.dev() == ec).all())
            n_cases += 1
            if not ok:
                bad += 1
                print("BAD k", k, "canonical", canonical, reads, read_len, mode, glen, keys.size, ek.numel())
print("stress_fused done: %d cases, bad = %d" % (n_cases, bad))
