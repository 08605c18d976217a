"""Experiment: file -> 31-mer histogram through the API in chunks, wall time only (no synchronisation inside the loop)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bionumpy_amd as bnp
from bionumpy_amd import synth
n_file = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 256_000_000
mode, glen = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1, 5_000_000)
path = "/tmp/bnpk_stream_test.fq"
synth.fastq_bytes(n_file, 150, 7, mode, glen).tofile(path)
def run():
    total = None
    for ch in bnp.open(path).read_chunks(min_chunk_size=chunk):
        c = bnp.sequence.count_kmers(ch.sequence, 31)
        total = c if total is None else total + c
    torch.cuda.synchronize()
    return total
run()
for rep in range(4):
    t0 = time.perf_counter(); c = run(); dt = time.perf_counter() - t0
    print("total %.1f ms -> %.2f GB/s of file, %.2f Gbases/s, %d distinct" % (dt * 1e3, os.path.getsize(path) / dt / 1e9, n_file * 150 / dt / 1e9, len(c)), flush=True)
os.remove(path)
if os.environ.get("BNPK_CPROFILE"):
    import cProfile, pstats
    synth.fastq_bytes(n_file, 150, 7, mode, glen).tofile(path)
    run()
    pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
    os.remove(path)
