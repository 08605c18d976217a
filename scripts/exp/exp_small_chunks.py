"""Experiment: where a 5 MB chunk's ~1.2 ms go on the API path (cProfile by total time, and the synchronising calls)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bionumpy_amd as bnp
from bionumpy_amd import synth
n_file, chunk = 400_000, 5_000_000
path = "/tmp/bnpk_stream_small.fq"
synth.fastq_bytes(n_file, 150, 7, 1, 5_000_000).tofile(path)
def run():
    total = None
    n = 0
    for ch in bnp.open(path).read_chunks(min_chunk_size=chunk):
        c = bnp.sequence.count_kmers(ch.sequence, 31)
        total = c if total is None else total + c
        n += 1
    torch.cuda.synchronize()
    return total, n
run(); run()
t0 = time.perf_counter(); _, n = run(); dt = time.perf_counter() - t0
print("%.1f ms for %d chunks: %.2f ms per chunk, %.2f Gbases/s" % (dt * 1e3, n, dt * 1e3 / n, n_file * 150 / dt / 1e9))
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
st.sort_stats("cumulative").print_stats(30)
os.remove(path)
