"""Experiment: run-to-run variation of the finishing kernel on random 62-bit keys."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ctypes as C
from bionumpy_amd.device import Device, ptr
from bionumpy_amd.ops import get_ops
from bionumpy_amd._native import lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ops = get_ops(); dev = Device.get()
g = torch.Generator(device="cuda"); g.manual_seed(1)
keys = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device="cuda", generator=g)
a, off1 = ops.radix_partition(keys, None, 1, 52, 10)
b, off2 = ops.radix_partition(a, off1, 1 << 10, 43, 9)      # 2^19 buckets of ~5.7 K keys for n = 3e9
del keys
nseg = 1 << 19
counts = torch.empty(n, dtype=torch.int64, device="cuda")
state = torch.empty(lib.bnpk_finish_state_words(nseg), dtype=torch.int64, device="cuda")
times = []
for rep in range(reps):
    nu, ov = C.c_int64(0), C.c_int(0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.bnpk_finish_sorted(dev.ctx, ptr(b), n, ptr(off2), nseg, 43, ptr(a), ptr(counts), ptr(state), None, 0, None, None, C.byref(nu), C.byref(ov), dev.stream())
    e1.record(); torch.cuda.synchronize()
    times.append(round(e0.elapsed_time(e1), 2))
print("finish ms:", times, "n_unique", nu.value, "state[0:8]", state[:8].tolist())
