"""Round 6: the fused first level alone (hist + scatter of bnpk_kmers_partition), 50 M reads, with either scatter kernel
(option "l1_ring").  BNPK_LIB=<variant> times an experiment build (scripts/exp/build_variant.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bionumpy_amd._native import lib
from bionumpy_amd.device import Device
from bionumpy_amd.ops import get_ops
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ops = get_ops(); dev = Device.get()
text = ops.synth_fastq(reads, 150, 20260925, int(os.environ.get("MB_MODE", "0")), 100_000_000, 0)
packed, ends, n, n_bases = ops.fastq_encode(text, text.size, 4, 1, ord("@"), True)
starts, n_kmers = ops.kmer_starts_from_ends(ends, n_bases, 31)
del text, ends
for ring in (0, 1):
    assert lib.bnpk_set_option(dev.ctx, b"l1_ring", ring) == 0
    h, cuts = ops.kmers_partitioned(packed, starts, n_bases, n_kmers, 31, bits); del h
    torch.cuda.synchronize()
    dev.prof_enable(True); dev.prof_reset()
    for _ in range(3):
        h, cuts = ops.kmers_partitioned(packed, starts, n_bases, n_kmers, 31, bits); del h
    torch.cuda.synchronize()
    rep = dev.prof_report(); dev.prof_enable(False)
    print(os.path.basename(os.environ.get("BNPK_LIB", "product")), "l1_ring=%d" % ring,
          "  ".join("%s %.2f ms" % (k, v["total_ms"] / 3) for k, v in rep.items()), flush=True)
