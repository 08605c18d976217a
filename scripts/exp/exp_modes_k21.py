"""Experiment: the finishing stage under every finish_mode on nearly-distinct keys (random 21-mers: 0.07 % repeats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bionumpy_amd.device import Device
from bionumpy_amd.ops import get_ops
from bionumpy_amd._native import lib
from bionumpy_amd.pipeline import fastq_kmer_histogram
k = int(sys.argv[1]) if len(sys.argv) > 1 else 21
reads = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000_000
ops = get_ops(); dev = Device.get()
text = ops.synth_fastq(reads, 150, 20260925, 0, 0, 0)
for mode in [int(m) for m in (sys.argv[3].split(',') if len(sys.argv) > 3 else '0,1,2,3,4,5'.split(','))]:
    assert lib.bnpk_set_option(dev.ctx, b"finish_mode", mode) == 0
    (keys, counts), st = fastq_kmer_histogram(text, k); del keys, counts
    dev.prof_enable(True); dev.prof_reset()
    (keys, counts), st = fastq_kmer_histogram(text, k)
    torch.cuda.synchronize()
    rep = dev.prof_report(); dev.prof_enable(False)
    print("finish_mode %d: finish_sorted %.1f ms, distinct %d of %d" % (mode, rep["finish_sorted"]["total_ms"], keys.size, st.n_kmers), flush=True)
    print("   ", ops.last_sparse_info)
    del keys, counts
lib.bnpk_set_option(dev.ctx, b"finish_mode", 0)
