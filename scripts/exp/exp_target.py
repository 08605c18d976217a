"""the partition plan's bucket size (HipOps.FINISH_TARGET) against the step: 6500 -> [10, 10] bits, 3000 -> [11, 10]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bionumpy_amd.device import Device
from bionumpy_amd import ops as ops_mod
from bionumpy_amd.pipeline import fastq_kmer_histogram
ops = ops_mod.get_ops(); dev = Device.get()
text = ops.synth_fastq(50_000_000, 150, 20260925, 0, 0, 0)
for target in (6500, 3000):
    type(ops).FINISH_TARGET = target
    h, st = fastq_kmer_histogram(text, 31); del h
    torch.cuda.synchronize(); dev.prof_enable(True); dev.prof_reset()
    for _ in range(2):
        h, st = fastq_kmer_histogram(text, 31); n = h[0].size; del h
    torch.cuda.synchronize(); rep = dev.prof_report(); dev.prof_enable(False)
    print(target, n, {k: round(v["total_ms"] / 2, 2) for k, v in rep.items()}, round(sum(v["total_ms"] for v in rep.values()) / 2, 2))
