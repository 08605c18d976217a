"""Experiment: does a working set that fits the 256 MiB Infinity Cache stay there between kernels?
bnpk_copy_rates (dst[i] = src[i], four forms) over buffers of 16 MB .. 2 GB: bytes read + written per second.  If a copy of
64 MB (working set 128 MB) runs no faster than one of 2 GB, re-reads do not come out of the cache at a higher rate than HBM
delivers them, and ordering the level-2 histogram / scatter / finishing per group of level-1 buckets buys nothing."""
import os, sys, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bionumpy_amd.device import Device
from bionumpy_amd._native import lib
dev = Device.get()
out = {}
for mb in (16, 32, 64, 96, 128, 256, 512, 2048):
    n = mb << 20
    src = torch.empty(n, dtype=torch.uint8, device="cuda").random_(0, 255)
    dst = torch.empty_like(src)
    rates = (C.c_double * 4)()
    reps = max(5, (4 << 30) // n)
    assert lib.bnpk_copy_rates(dev.ctx, C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), n, reps, rates, None) == 0
    out["%d MB" % mb] = {"working_set_MB": 2 * mb, "reps": reps,
                         "GB_per_s": dict(zip(("nontemporal_loop", "plain_loop", "one_float4_per_thread", "four_per_iteration"), (round(r) for r in rates)))}
    del src, dst
print(json.dumps(out, indent=1))
