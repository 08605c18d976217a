"""Localise the >2^32-element failure: run the pipeline stage by stage at a given size with invariants."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bionumpy_amd.ops import get_ops
from bionumpy_amd.device import HArray
import ctypes as C
from bionumpy_amd._native import lib
from bionumpy_amd.device import ptr

reads = int(sys.argv[1])
k = 31
ops = get_ops()
text = ops.synth_fastq(reads, 150, 20260925, 0, 0, 0)
scan = ops.scan_lines(text, text.size, 4, ord('@'), True)
print("records", scan.n_records, "size", scan.size, flush=True)
starts, lens = ops.field_table(text, scan.newlines, scan.n_records, 4, 1, 0, False)
print("lens ok", bool((lens.dev() == 150).all().item()), flush=True)
offsets, total = ops.row_offsets(lens, 1)
print("total bases", total, flush=True)
_, packed = ops.gather_encode_dna(text, starts, offsets, scan.n_records, total, False, True)
out_off, n_out = ops.row_offsets(lens, k)
print("n_out", n_out, flush=True)
h = ops.kmers(packed, offsets, out_off, scan.n_records, n_out, k)
t = h.dev()
print("kmers min/max", int(t.min()), int(t.max()), flush=True)
s0 = int(t.sum().item()); x0 = 0
# spot check a few k-mers at the far end against the numpy twin
from bionumpy_amd import synth
import oracle
last = synth.read_codes(1, 150, 20260925, 0, 0, reads - 1)[0]
hh, _ = oracle.get_kmers(last, np.array([150]), k)
print("last read kmers equal", bool(np.array_equal(t[-120:].cpu().numpy(), hh)), flush=True)
del packed, starts, offsets, out_off, scan
sorted_t, free_t = ops.sort_keys(t, 62)
torch.cuda.synchronize()
print("sum preserved", int(sorted_t.sum().item()) == s0, flush=True)
CH = 1 << 28
ok = True
for a in range(0, sorted_t.numel() - 1, CH):
    b = min(a + CH + 1, sorted_t.numel())
    ok &= bool((sorted_t[a + 1:b] >= sorted_t[a:b - 1]).all().item())
print("sorted", ok, flush=True)
n_runs, tile_off = ops._runs(sorted_t)
print("n_runs", n_runs, "tile_off last", int(tile_off[-1].item()), flush=True)
