import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bionumpy_amd as bnp
from bionumpy_amd.device import Device
from bionumpy_amd.ops import get_ops
ops = get_ops(); dev = Device.get()
text = ops.synth_fastq(50_000_000, 150, 20260925, 0, 0, 0)
buf = bnp.FastQBuffer.from_raw_buffer(text)
seqs = bnp.change_encoding(buf.get_field_by_number(1), bnp.DNAEncoding); seqs._compact()
r = bnp.sequence.get_reverse_complement(seqs); del r
torch.cuda.synchronize(); dev.prof_enable(True); dev.prof_reset()
for _ in range(3):
    r = bnp.sequence.get_reverse_complement(seqs); del r
torch.cuda.synchronize()
print({k: round(v["total_ms"] / 3, 3) for k, v in dev.prof_report().items()})
