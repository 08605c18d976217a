"""Experiment: where the quality-filter API path spends its time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bionumpy_amd as bnp
from bionumpy_amd.ops import get_ops
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
ops = get_ops()
text = ops.synth_fastq(reads, 150, 20260925, 0, 0, 0)
def sync(): torch.cuda.synchronize()
def step(verbose):
    t = [time.perf_counter()]
    def lap(name):
        sync(); t.append(time.perf_counter())
        if verbose: print("  %-28s %7.1f ms" % (name, (t[-1] - t[-2]) * 1e3), flush=True)
    chunk = bnp.SequenceEntryWithQuality._lazy(bnp.FastQBuffer.from_raw_buffer(text)); lap("from_raw_buffer")
    q = chunk.quality; lap("chunk.quality")
    means = np.mean(q, axis=1); lap("np.mean(q, axis=1) -> %s" % type(means).__name__)
    med = float(np.median(means[:100000])); lap("median of 100k")
    keep = means >= med; lap("means >= med -> %s" % type(keep).__name__)
    keep[::3] = False; lap("keep[::3] = False")
    sub = chunk[keep]; lap("chunk[keep]")
    buf = sub.get_buffer(); lap("get_buffer")
    kept = buf.entry_bytes(); lap("entry_bytes")
    n = int(keep.sum()); lap("keep.sum()")
    return n, kept.size
step(False)
t0 = time.perf_counter(); print(step(True)); sync(); print("total %.1f ms" % ((time.perf_counter() - t0) * 1e3))
