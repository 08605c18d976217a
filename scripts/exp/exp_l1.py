"""Experiment: the fused level-1 pass (k-mer generation + scatter) at a given digit width."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bionumpy_amd.device import Device
from bionumpy_amd.ops import get_ops
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
bits_list = [int(b) for b in sys.argv[2].split(",")] if len(sys.argv) > 2 else [9]
ops = get_ops(); dev = Device.get()
text = ops.synth_fastq(reads, 150, 20260925, 0, 0, 0)
packed, ends, n, n_bases = ops.fastq_encode(text, text.size, 4, 1, ord("@"), True)
starts, n_kmers = ops.kmer_starts_from_ends(ends, n_bases, 31)
del text, ends
for bits in bits_list:
    h, cuts = ops.kmers_partitioned(packed, starts, n_bases, n_kmers, 31, bits); del h
    dev.prof_enable(True); dev.prof_reset()
    for _ in range(2):
        h, cuts = ops.kmers_partitioned(packed, starts, n_bases, n_kmers, 31, bits); del h
    torch.cuda.synchronize()
    rep = dev.prof_report(); dev.prof_enable(False)
    c = cuts.host()
    print("level 1, %2d bits: " % bits + "  ".join("%s %.2f ms" % (k, v["total_ms"] / 2) for k, v in rep.items()) + "  buckets ok %s" % bool(c[-1] == n_kmers), flush=True)
