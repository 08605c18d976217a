"""The reference's own loop at its own chunk size (scripts/kmer_counting_example.py:4-17 with k = 31; bionumpy/io/npdataclassreader.py:94
default min_chunk_size = 5000000): file in the page cache -> 31-mer histogram, (a) the example's form — a user function per chunk
(as_encoded_array, get_kmers, count_encoded) summed with sum() —, (b) the library's count_kmers per chunk, (c) count_kmers over the
stream (streamable(sum)); each at 5 MB chunks and, for the fixed cost per chunk, at 256 MB chunks.  Prints one JSON object.

    python scripts/exp/exp_reference_loop.py [n_reads] [k]
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bionumpy_amd as bnp
from bionumpy_amd import synth
from bionumpy_amd.ops import get_ops

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
path = "/tmp/bnpk_reference_loop.fq"
ops = get_ops()
text = ops.synth_fastq(n_reads, 150, 7, 1, 50_000_000)
text.host().tofile(path)
file_bytes = os.path.getsize(path)
del text


def user_count_kmers(sequence_entries):                     # scripts/kmer_counting_example.py:4-7 (k = 5 there)
    sequence = bnp.as_encoded_array(sequence_entries, bnp.DNAEncoding)
    kmers = bnp.get_kmers(sequence, k=k)
    return bnp.count_encoded(kmers, axis=None)


def example_form(chunk):
    stream = bnp.open(path).read_chunks(min_chunk_size=chunk)
    n = [0]
    def counted():
        for c in stream:
            n[0] += 1
            yield user_count_kmers(c.sequence)
    total = sum(counted())
    return total, n[0]


def library_form(chunk):
    total, n = None, 0
    for c in bnp.open(path).read_chunks(min_chunk_size=chunk):
        h = bnp.count_kmers(c.sequence, k)
        total = h if total is None else total + h
        n += 1
    return total, n


def stream_form(chunk):
    n = [0]
    return bnp.count_kmers(bnp.open(path).read_chunks(min_chunk_size=chunk).sequence, k), None


def timed(f, chunk):
    f(chunk)[0].keys                                          # (warm: page cache, pinned buffers, kernels)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    total, n = f(chunk)
    distinct = len(total)                                     # counting what is pending is part of the job
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dt, n, distinct, total


out = {"file_bytes": file_bytes, "reads": n_reads, "k": k}
ref = None
for name, f in (("example_form", example_form), ("library_form", library_form), ("stream_form", stream_form)):
    res = {}
    for chunk in (5_000_000, 256 << 20):
        dt, n, distinct, total = timed(f, chunk)
        if ref is None:
            ref = (total.keys.copy(), total.counts.copy())
        same = np.array_equal(total.keys, ref[0]) and np.array_equal(total.counts, ref[1])
        res["chunk_%d" % chunk] = {"ms": round(dt * 1e3, 1), "chunks": n, "gbases_per_s": round(n_reads * 150 / dt / 1e9, 2),
                                   "file_gb_per_s": round(file_bytes / dt / 1e9, 2), "distinct": distinct, "same_histogram": bool(same)}
        del total
    small, big = res["chunk_5000000"], res["chunk_%d" % (256 << 20)]
    if small["chunks"]:
        res["fixed_ms_per_5MB_chunk"] = round((small["ms"] - big["ms"]) / small["chunks"], 3)
        res["ms_per_5MB_chunk"] = round(small["ms"] / small["chunks"], 3)
    out[name] = res
os.remove(path)
print(json.dumps(out))
