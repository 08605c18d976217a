"""Experiment: where a streamed file -> histogram run spends its time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bionumpy_amd as bnp
from bionumpy_amd import synth
n_file = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 256_000_000
path = "/tmp/bnpk_stream_test.fq"
synth.fastq_bytes(n_file, 150, 7, 1, 5_000_000).tofile(path)
def sync(): torch.cuda.synchronize()
def run(verbose):
    t0 = time.perf_counter(); total = None; tr = tc = ta = 0.0
    it = iter(bnp.open(path).read_chunks(min_chunk_size=chunk))
    while True:
        a = time.perf_counter()
        try: ch = next(it)
        except StopIteration: break
        b = time.perf_counter()
        c = bnp.sequence.count_kmers(ch.sequence, 31)
        d = time.perf_counter()
        total = c if total is None else total + c
        e = time.perf_counter()
        tr += b - a; tc += d - b; ta += e - d
    sync(); dt = time.perf_counter() - t0
    if verbose: print("total %.1f ms: read_chunk %.1f, count_kmers %.1f, add %.1f  -> %.2f GB/s, %.2f Gbases/s" % (dt * 1e3, tr * 1e3, tc * 1e3, ta * 1e3, os.path.getsize(path) / dt / 1e9, n_file * 150 / dt / 1e9))
run(False); run(True); run(True)
os.remove(path)
