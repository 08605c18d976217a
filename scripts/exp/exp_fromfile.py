"""Experiment: where the time of count_kmers(bnp.open(path).read_chunks().sequence, 31) goes for a 2.5 GB FASTQ in the page cache."""
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
reads = int(os.environ.get("READS", 8_000_000))
path = "/dev/shm/bnpk_ff.fq"
with open(path, "wb") as f:
    for first in range(0, reads, 4_000_000):
        f.write(ops.synth_fastq(min(4_000_000, reads - first), 150, 20260925, 0, 0, first).host().tobytes())
size = os.path.getsize(path)
chunk = int(os.environ.get("CHUNK_MB", 256)) << 20


def timed(name, fn, n=3):
    fn(); torch.cuda.synchronize()
    dev.prof_enable(True); dev.prof_reset()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    prof = dev.prof_report(); dev.prof_enable(False)
    k = sum(v["total_ms"] for v in prof.values()) / n
    print("%-28s %7.1f ms  %5.1f GB/s   kernels %.1f ms" % (name, dt * 1e3, size / dt / 1e9, k))
    top = sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])[:8]
    print("      ", {a: round(b["total_ms"] / n, 2) for a, b in top})


def only_read():
    for c in bnp.open(path).read_chunks(min_chunk_size=chunk):
        pass


def decode():
    for c in bnp.open(path).read_chunks(min_chunk_size=chunk):
        s = bnp.change_encoding(c.sequence, bnp.DNAEncoding)
        s._compact()


def kmers():
    for c in bnp.open(path).read_chunks(min_chunk_size=chunk):
        km = bnp.get_kmers(bnp.change_encoding(c.sequence, bnp.DNAEncoding), 31)
        km._compact()


def count():
    h = bnp.count_kmers(bnp.open(path).read_chunks(min_chunk_size=chunk).sequence, 31)
    h._keys


def raw_read():
    r = bnp.open(path)
    f = r._reader if hasattr(r, "_reader") else None
    print("reader", type(r), type(f))


raw_read()
timed("read_chunks only", only_read)
timed("+ sequence as 2-bit DNA", decode)
timed("+ 31-mers", kmers)
timed("count_kmers (whole step)", count)
if os.environ.get("CPROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable(); count(); torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
os.unlink(path)
