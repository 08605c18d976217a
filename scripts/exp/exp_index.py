"""config 5 (sacCer3 k = 31 KmerIndex): where do the milliseconds of create_index go?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bionumpy_amd as bnp
from bionumpy_amd.device import Device
dev = Device.get()
gold = os.path.join(ROOT, "tests", "golden")
genome = bnp.open(os.path.join(gold, "sacCer3.fa.gz")).read()
seqs = bnp.change_encoding(genome.sequence, bnp.DNAEncoding)
for rep in range(3):
    index = bnp.KmerIndex.create_index(seqs, k=31)
torch.cuda.synchronize()
dev.prof_enable(True); dev.prof_reset()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    index = bnp.KmerIndex.create_index(seqs, k=31)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n * 1e3
prof = dev.prof_report(); dev.prof_enable(False)
print("create_index %.2f ms wall; kernels (ms per build, launches per build):" % dt)
tot = 0
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"]):
    print("   %-28s %.3f  x%.1f" % (k, v["total_ms"] / n, v["launches"] / n)); tot += v["total_ms"] / n
print("   sum of kernels %.2f ms" % tot)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    index = bnp.KmerIndex.create_index(seqs, k=31)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
