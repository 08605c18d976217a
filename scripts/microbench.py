"""Kernel-level timing of the pipeline stages on one batch (hipEvent timers of the C-ABI)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bionumpy_amd.device import Device
from bionumpy_amd.ops import get_ops
from bionumpy_amd.pipeline import fastq_kmer_histogram

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mode = int(os.environ.get("MB_MODE", "0"))
k = int(os.environ.get("MB_K", "31"))
ops = get_ops(); dev = Device.get()
if os.environ.get("MB_L1_RING"):                     # (round 6 experiment: the fixed-line form of the fused first level)
    from bionumpy_amd._native import lib
    assert lib.bnpk_set_option(dev.ctx, b"l1_ring", int(os.environ["MB_L1_RING"])) == 0
if os.environ.get("MB_FINISH_MODE"):                # (force a finishing path: include/bnpk.h "finish_mode")
    from bionumpy_amd._native import lib
    assert lib.bnpk_set_option(dev.ctx, b"finish_mode", int(os.environ["MB_FINISH_MODE"])) == 0
text = ops.synth_fastq(reads, 150, 20260925, mode, int(os.environ.get("MB_GENOME_LEN", "100000000")), 0)
h, st = fastq_kmer_histogram(text, k); del h
torch.cuda.synchronize()
dev.prof_enable(True); dev.prof_reset()
for _ in range(reps):
    h, st = fastq_kmer_histogram(text, k); del h
torch.cuda.synchronize()
rep = dev.prof_report()
tot = 0
for k, v in rep.items():
    print("%-22s %8.3f ms" % (k, v["total_ms"] / reps)); tot += v["total_ms"] / reps
print("%-22s %8.3f ms  (%.2f Gbases/s)" % ("SUM", tot, st.n_bases / tot / 1e6))
