"""kernel times of the API-object decode path (FastQBuffer.from_raw_buffer + change_encoding) and of gather_rows"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bionumpy_amd as bnp
from bionumpy_amd.device import Device
from bionumpy_amd.ops import get_ops

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
ops = get_ops(); dev = Device.get()
text = ops.synth_fastq(reads, 150, 20260925, 0, 0, 0)
def step():
    buf = bnp.FastQBuffer.from_raw_buffer(text)
    seqs = bnp.change_encoding(buf.get_field_by_number(1), bnp.DNAEncoding)
    q = buf.get_field_by_number(3)
    q._compact()
    return seqs
step(); torch.cuda.synchronize()
dev.prof_enable(True); dev.prof_reset()
for _ in range(3):
    step()
torch.cuda.synchronize()
rep = dev.prof_report()
print(json.dumps({"reads": reads, "file_bytes": int(text.size), "kernels_ms": {k: round(v["total_ms"] / 3, 3) for k, v in rep.items()}}))
