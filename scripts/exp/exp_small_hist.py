"""Round 6: bnpk_count_sparse on ~12 M keys — uniform 62-bit and 28-bit keys against the sacCer3 31-mers (skewed top digits AND
long shared prefixes inside a bucket) — per-kernel ms and what the planner reports.  (The run recorded in NOTES.md had a
key-space equalisation in front of the planner, option "sparse_equalise": 6.4-6.7 ms against 3.5 without — the finishing kernels'
time is in the long bins INSIDE a bucket, which no stretching of the top 16 bits touches; removed.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bionumpy_amd as bnp
from bionumpy_amd._native import lib
from bionumpy_amd.device import Device, HArray
from bionumpy_amd.ops import get_ops
ops = get_ops(); dev = Device.get()
genome = bnp.open(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "sacCer3.fa.gz")).read()
kmers = bnp.get_kmers(bnp.change_encoding(genome.sequence, bnp.DNAEncoding), 31); kmers._compact()
flat = kmers._flat_data()
n = flat.size
rng = np.random.default_rng(1)
cases = [("uniform 62-bit", HArray(host=rng.integers(0, 1 << 62, size=n, dtype=np.int64)), 62, 1),
         ("uniform 28-bit", HArray(host=rng.integers(0, 1 << 28, size=n, dtype=np.int64)), 28, 1),
         ("sacCer3", flat, 62, 0)]
for name, h, bits, eq in cases:
    h.dev()
    for claim in (True, False):
        ops.claim_last_level = claim
        k, c = ops.count_sparse(h, key_bits=bits); torch.cuda.synchronize()
        dev.prof_enable(True); dev.prof_reset()
        t0 = time.perf_counter()
        for _ in range(5):
            k, c = ops.count_sparse(h, key_bits=bits)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5 * 1e3
        rep = dev.prof_report(); dev.prof_enable(False)
        print("%-20s claim=%d: %.2f ms, distinct %d, %s, %s" % (name, claim, dt, k.size, {a: b for a, b in ops.last_sparse_info.items() if a != "workspace"},
              {a: round(v["total_ms"] / 5, 3) for a, v in rep.items() if v["total_ms"] / 5 > 0.04}), flush=True)
ops.claim_last_level = True
