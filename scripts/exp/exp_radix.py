"""Experiment: timing of the radix partition levels + finishing kernel on random 62-bit keys (no FASTQ stages).
usage: exp_radix.py [n_keys] [bits1] [bits2] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from bionumpy_amd.device import Device, HArray
from bionumpy_amd.ops import get_ops
import ctypes as C
from bionumpy_amd._native import lib
from bionumpy_amd.device import ptr

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_200_000_000
b1 = int(sys.argv[2]) if len(sys.argv) > 2 else 10
b2 = int(sys.argv[3]) if len(sys.argv) > 3 else 8
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
ops = get_ops(); dev = Device.get()
g = torch.Generator(device="cuda"); g.manual_seed(1)
keys = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device="cuda", generator=g)
torch.cuda.synchronize()
dev.prof_enable(True); dev.prof_reset()
for rep in range(reps + 1):
    if rep == 1:
        dev.prof_reset()
    a, off1 = ops.radix_partition(keys, None, 1, 62 - b1, b1)
    b, off2 = ops.radix_partition(a, off1, 1 << b1, 62 - b1 - b2, b2)
    nseg = 1 << (b1 + b2)
    counts = torch.empty(n, dtype=torch.int64, device="cuda")
    state = torch.empty(lib.bnpk_finish_state_words(nseg), dtype=torch.int64, device="cuda")
    nu, ov = C.c_int64(0), C.c_int(0)
    lib.bnpk_finish_sorted(dev.ctx, ptr(b), n, ptr(off2), nseg, 62 - b1 - b2, ptr(a), ptr(counts), ptr(state), None, 0, None, None, C.byref(nu), C.byref(ov), dev.stream())
    torch.cuda.synchronize()
    if rep == 0 and not os.environ.get("BNPK_ABLATE"):
        print("n_unique", nu.value, "overflow", ov.value, "sorted", bool((a[1:nu.value] > a[:nu.value - 1]).all().item()))
    del a, b, counts, state, off1, off2
rep_ = dev.prof_report()
for k, v in rep_.items():
    ms = v["total_ms"] / reps
    print("%-24s %8.3f ms   %.2f ns/key-GB: %.0f GB/s (8B/key)" % (k, ms, 0, n * 8 / ms / 1e6))
