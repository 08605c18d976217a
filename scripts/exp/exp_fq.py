"""Experiment: the fused FASTQ decode (census + encode) with either tile encoder."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bionumpy_amd._native import lib
from bionumpy_amd.device import Device
from bionumpy_amd.ops import get_ops
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
read_len = int(sys.argv[2]) if len(sys.argv) > 2 else 150
modes = [int(m) for m in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 0]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
ops = get_ops(); dev = Device.get()
text = ops.synth_fastq(reads, read_len, 20260925, 0, 0, 0)
for mode in modes:
    assert lib.bnpk_set_option(dev.ctx, b"fastq_encoder", mode) == 0
    r = ops.fastq_encode(text, text.size, 4, 1, ord("@"), True); del r
    dev.prof_enable(True); dev.prof_reset()
    for _ in range(reps):
        r = ops.fastq_encode(text, text.size, 4, 1, ord("@"), True); del r
    torch.cuda.synchronize()
    rep = dev.prof_report(); dev.prof_enable(False)
    print("encoder %d, %d reads x %d (%.2f GB): " % (mode, reads, read_len, text.size / 1e9) +
          "  ".join("%s %.3f ms (%.0f GB/s of text)" % (k, v["total_ms"] / reps, text.size / (v["total_ms"] / reps) / 1e6) for k, v in rep.items()), flush=True)
