"""the step on ALTERNATING inputs (reads of a genome at 3x coverage, then random reads, then 12x, ...): what a stream of different
batches sees — stale per-bucket words of the previous batch are not those of an identical call"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bionumpy_amd.device import Device
from bionumpy_amd.ops import get_ops
from bionumpy_amd.pipeline import fastq_kmer_histogram
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
ops = get_ops(); dev = Device.get()
texts = [("genome 3x", ops.synth_fastq(reads, 150, 20260925, 1, 2_500_000_000, 0)),
         ("uniform", ops.synth_fastq(reads, 150, 20260926, 0, 100_000_000, 0)),
         ("genome 12x", ops.synth_fastq(reads, 150, 20260927, 1, 625_000_000, 0))]
for rnd in range(3):
    for name, text in texts:
        torch.cuda.synchronize(); t0 = time.time()
        h, st = fastq_kmer_histogram(text, 31); del h
        torch.cuda.synchronize()
        print("round %d %-10s %7.1f ms" % (rnd, name, (time.time() - t0) * 1e3), flush=True)
