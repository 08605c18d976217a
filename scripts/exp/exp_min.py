"""config 3 (minimizers, k = 31, window 40) through the API objects and through the fused pipeline: step and kernel times"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bionumpy_amd as bnp
from bionumpy_amd.device import Device
from bionumpy_amd.ops import get_ops
from bionumpy_amd.pipeline import fastq_minimizers

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
ops = get_ops(); dev = Device.get()
text = ops.synth_fastq(reads, 150, 20260925, 0, 0, 0)
def api():
    buf = bnp.FastQBuffer.from_raw_buffer(text)
    seqs = bnp.change_encoding(buf.get_field_by_number(1), bnp.DNAEncoding)
    return bnp.get_minimizers(seqs, 31, 40)
def fused():
    return fastq_minimizers(text, 31, 40)[0]
out = {"reads": reads}
for name, f in (("api", api), ("fused", fused)):
    m = f(); del m
    torch.cuda.synchronize(); dev.prof_enable(True); dev.prof_reset()
    t0 = time.perf_counter()
    for _ in range(3):
        m = f(); del m
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    rep = dev.prof_report(); dev.prof_enable(False)
    out[name] = {"ms_per_step": round(dt * 1e3, 2), "kernels_ms": {k: round(v["total_ms"] / 3, 3) for k, v in rep.items()}}
print(json.dumps(out))
