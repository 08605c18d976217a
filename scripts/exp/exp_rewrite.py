"""Experiment: the kernels of the reverse-complement rewrite and the read filter of bench.py's extras, on 50 M reads."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bionumpy_amd as bnp
from bionumpy_amd import ops as O
from bionumpy_amd.device import Device

ops = O.get_ops(); dev = Device.get()
text = ops.synth_fastq(int(os.environ.get("READS", 50_000_000)), 150, 20260925, 0, 0, 0)


def timed(name, fn, n=3):
    r = fn(); del r; torch.cuda.synchronize()
    dev.prof_enable(True); dev.prof_reset()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn(); del r
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    prof = dev.prof_report(); dev.prof_enable(False)
    print("%-10s %6.2f ms" % (name, dt * 1e3), {a: round(b["total_ms"] / n, 2) for a, b in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])[:9]})


def rewrite():
    chunk = bnp.SequenceEntryWithQuality._lazy(bnp.FastQBuffer.from_raw_buffer(text))
    rc = bnp.get_reverse_complement(chunk.sequence)
    return bnp.FastQBuffer.from_data(bnp.replace(chunk, sequence=rc))


def packed_rc():
    return bnp.get_reverse_complement(dna)


def read_filter():
    chunk = bnp.SequenceEntryWithQuality._lazy(bnp.FastQBuffer.from_raw_buffer(text))
    keep = np.mean(chunk.quality, axis=1) >= 0.0
    keep[::3] = False
    return chunk[keep].get_buffer().entry_bytes()


def encode():
    s = bnp.change_encoding(bnp.FastQBuffer.from_raw_buffer(text).get_field_by_number(1), bnp.DNAEncoding)
    s._compact()
    return s


timed("rewrite", rewrite)
timed("filter", read_filter)
timed("encode", encode)
dna = bnp.change_encoding(bnp.FastQBuffer.from_raw_buffer(text).get_field_by_number(1), bnp.DNAEncoding)
dna._compact()
timed("rc packed", packed_rc)
