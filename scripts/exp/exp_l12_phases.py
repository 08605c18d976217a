"""Experiment (builds with -DRP_PHASES=<wave> only): where the rounds of the scatter kernel spend their cycles, level 1 and
level 2 of the 50 M-read bench.  Prints cycles per workgroup-slab averaged, as a share of the total."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bionumpy_amd.device import Device
from bionumpy_amd.ops import get_ops
from bionumpy_amd._native import lib
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
ops = get_ops(); dev = Device.get()
text = ops.synth_fastq(reads, 150, 20260925, 0, 0, 0)
packed, ends, n, n_bases = ops.fastq_encode(text, text.size, 4, 1, ord("@"), True)
starts, n_kmers = ops.kmer_starts_from_ends(ends, n_bases, 31)
del text, ends
names = ["layout", "meta+carry stage", "stage new", "wait+keys+rank", "issue loads", "flush", "readback+barrier", "-"]
def phases(tag):
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    lib.bnpk_debug_radix_phases.restype = ctypes.c_int
    assert lib.bnpk_debug_radix_phases(buf) == 0
    tot = float(sum(buf)) or 1.0
    print(tag, "  ".join("%s %.1f%%" % (nm, 100.0 * c / tot) for nm, c in zip(names, buf) if c), " | total Mcycles/CU %.1f" % (tot / 256 / 1e6), flush=True)
h, cuts = ops.kmers_partitioned(packed, starts, n_bases, n_kmers, 31, 10)
ht, ct = h.dev(), cuts.dev()
out = torch.empty_like(ht)
phases("(warm-up)")
del h
h, cuts = ops.kmers_partitioned(packed, starts, n_bases, n_kmers, 31, 10)
phases("level 1:")
o, child = ops.radix_partition(ht, ct, 1 << 10, 42, 10, out)
phases("level 2:")
