"""Experiment: one 256 MB chunk through the API path (count_kmers of chunk.sequence): wall time vs the sum of kernel times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bionumpy_amd as bnp
from bionumpy_amd import synth
from bionumpy_amd.device import Device
n = int(sys.argv[1]) if len(sys.argv) > 1 else 800_000
path = "/tmp/bnpk_chunk_test.fq"
synth.fastq_bytes(n, 150, 7, 1, 5_000_000).tofile(path)
dev = Device.get()
chunk = bnp.open(path).read_chunk(min_chunk_size=1 << 30)
torch.cuda.synchronize()
for rep in range(3):
    dev.prof_enable(True); dev.prof_reset()
    t0 = time.perf_counter()
    c = bnp.sequence.count_kmers(chunk.sequence, 31)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rep_ = dev.prof_report(); dev.prof_enable(False)
    ksum = sum(v["total_ms"] for v in rep_.values())
    print("wall %.2f ms, kernels %.2f ms (%d timed calls): " % (dt * 1e3, ksum, sum(v["launches"] for v in rep_.values())) +
          "  ".join("%s %.2f" % (k, v["total_ms"]) for k, v in sorted(rep_.items(), key=lambda kv: -kv[1]["total_ms"])[:12]), flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
c = bnp.sequence.count_kmers(chunk.sequence, 31); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
os.remove(path)
