"""Round 6: bnpk_index_build on the sacCer3 31-mers (12.2 M (k-mer, row) pairs, skewed top bits) under every finish_mode —
which finishing kernel a small, skewed, nearly duplicate-free input should take."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bionumpy_amd as bnp
from bionumpy_amd._native import lib
from bionumpy_amd.device import Device
from bionumpy_amd.ops import get_ops
ops = get_ops(); dev = Device.get()
genome = bnp.open(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "sacCer3.fa.gz")).read()
seqs = bnp.change_encoding(genome.sequence, bnp.DNAEncoding)
kmers = bnp.get_kmers(seqs, 31); kmers._compact()
rows = ops.row_ids(kmers.offsets(), len(kmers), kmers.total())
flat = kmers._flat_data()
ref = None
for mode in (0, 1, 2, 3, 4, 5):
    assert lib.bnpk_set_option(dev.ctx, b"finish_mode", mode) == 0
    r = ops.unique_pairs(flat, rows, key_bits=62, n_values=len(kmers)); torch.cuda.synchronize()
    got = (r[0].dev().sum().item(), r[1].dev().sum().item(), r[0].size)
    ref = ref or got
    dev.prof_enable(True); dev.prof_reset()
    t0 = time.perf_counter()
    for _ in range(5):
        r = ops.unique_pairs(flat, rows, key_bits=62, n_values=len(kmers))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5 * 1e3
    rep = dev.prof_report(); dev.prof_enable(False)
    big = {k: round(v["total_ms"] / 5, 3) for k, v in rep.items() if v["total_ms"] / 5 > 0.05}
    print("finish_mode %d: %.2f ms wall, same %s, %s" % (mode, dt, got == ref, big), flush=True)
lib.bnpk_set_option(dev.ctx, b"finish_mode", 0)
