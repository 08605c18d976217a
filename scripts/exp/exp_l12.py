"""Experiment: the fused level 1 (k-mer generation + scatter) followed by level 2 over its buckets, 10 bits each — the two
partition passes of the 50 M-read bench, timed by the library's own event timers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bionumpy_amd.device import Device
from bionumpy_amd.ops import get_ops
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
ops = get_ops(); dev = Device.get()
text = ops.synth_fastq(reads, 150, 20260925, 0, 0, 0)
packed, ends, n, n_bases = ops.fastq_encode(text, text.size, 4, 1, ord("@"), True)
starts, n_kmers = ops.kmer_starts_from_ends(ends, n_bases, 31)
del text, ends
h, cuts = ops.kmers_partitioned(packed, starts, n_bases, n_kmers, 31, 10)
ht, ct = h.dev(), cuts.dev()
out = torch.empty_like(ht)
o, child = ops.radix_partition(ht, ct, 1 << 10, 42, 10, out)
dev.prof_enable(True); dev.prof_reset()
for _ in range(2):
    del h
    h, cuts = ops.kmers_partitioned(packed, starts, n_bases, n_kmers, 31, 10)
    o, child = ops.radix_partition(ht, ct, 1 << 10, 42, 10, out)
torch.cuda.synchronize()
rep = dev.prof_report(); dev.prof_enable(False)
print("  ".join("%s %.2f ms" % (k, v["total_ms"] / 2) for k, v in rep.items()), flush=True)
