"""the byte movers of the filter / rewrite configs: kernel times (scripts/bench_configs.py has the full versions)"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bionumpy_amd as bnp
from bionumpy_amd.device import Device
from bionumpy_amd.ops import get_ops
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
ops = get_ops(); dev = Device.get()
text = ops.synth_fastq(reads, 150, 20260925, 0, 0, 0)
def filter_step():
    chunk = bnp.SequenceEntryWithQuality._lazy(bnp.FastQBuffer.from_raw_buffer(text))
    means = np.mean(chunk.quality, axis=1)
    keep = means >= float(np.median(means[:100000]))
    keep[::3] = False
    return chunk[keep].get_buffer().entry_bytes().size
def rewrite_step():
    chunk = bnp.SequenceEntryWithQuality._lazy(bnp.FastQBuffer.from_raw_buffer(text))
    rc = bnp.sequence.get_reverse_complement(chunk.sequence)
    return bnp.FastQBuffer.from_data(bnp.replace(chunk, sequence=rc)).size
out = {}
for name, f in (("filter", filter_step), ("rewrite", rewrite_step)):
    f(); torch.cuda.synchronize(); dev.prof_enable(True); dev.prof_reset()
    t0 = time.perf_counter()
    for _ in range(2): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
    rep = dev.prof_report(); dev.prof_enable(False)
    out[name] = {"ms": round(dt * 1e3, 2), "kernels_ms": {k: round(v["total_ms"] / 2, 2) for k, v in rep.items()},
                 "launches": {k: v["launches"] / 2 for k, v in rep.items()}}
print(json.dumps(out))
