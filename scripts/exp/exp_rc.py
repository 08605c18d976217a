"""timing of reverse_complement_bytes / gather_rows / join_lines on 50 M reads (the FASTQ rewrite of scripts/bench_configs.py)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bionumpy_amd as bnp
from bionumpy_amd.device import Device
from bionumpy_amd.ops import get_ops
ops = get_ops(); dev = Device.get()
text = ops.synth_fastq(50_000_000, 150, 20260925, 0, 0, 0)
def rewrite_step():
    chunk = bnp.SequenceEntryWithQuality._lazy(bnp.FastQBuffer.from_raw_buffer(text))
    rc = bnp.sequence.get_reverse_complement(chunk.sequence)
    return bnp.FastQBuffer.from_data(bnp.replace(chunk, sequence=rc)).size
rewrite_step(); torch.cuda.synchronize(); dev.prof_enable(True); dev.prof_reset()
t0 = time.perf_counter()
for _ in range(2): rewrite_step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
print(round(dt * 1e3, 2), {k: round(v["total_ms"] / 2, 2) for k, v in dev.prof_report().items()})
