"""timing of match_string's kernel on 50 M reads (match_windows_packed incl. its tile count + scan)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bionumpy_amd as bnp
from bionumpy_amd.device import Device
from bionumpy_amd.ops import get_ops
ops = get_ops(); dev = Device.get()
text = ops.synth_fastq(50_000_000, 150, 20260925, 0, 0, 0)
buf = bnp.FastQBuffer.from_raw_buffer(text)
seqs = bnp.change_encoding(buf.get_field_by_number(1), bnp.DNAEncoding); seqs._compact()
def run():
    off, n_out = ops.row_offsets(seqs._lens, 7)
    return ops.match_windows(bnp.encoded_array.packed_words(seqs._data), seqs.offsets(), len(seqs), seqs.total(), n_out, [2, 0, 3, 3, 0, 1, 0], True)
h = run(); print("hits", int(h.dev().sum().item())); del h
torch.cuda.synchronize(); dev.prof_enable(True); dev.prof_reset()
for _ in range(3):
    h = run(); del h
torch.cuda.synchronize()
print({k: round(v["total_ms"] / 3, 2) for k, v in dev.prof_report().items()})
