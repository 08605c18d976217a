"""Experiment: the finishing kernels (general / fast + redo / auto) on random 62-bit keys: parity with torch.unique at
20 M keys, timing at 3e9 keys (2^19 buckets of ~5.7 K keys)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ctypes as C
from bionumpy_amd.device import Device, ptr, HArray
from bionumpy_amd.ops import get_ops
from bionumpy_amd._native import lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ops = get_ops(); dev = Device.get()
g = torch.Generator(device="cuda"); g.manual_seed(1)

def set_mode(m):
    assert lib.bnpk_set_option(dev.ctx, b"finish_mode", m) == 0

skip_parity = len(sys.argv) > 3 and sys.argv[3] == "noparity"
modes = [int(m) for m in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1, 2, 0]
# ---- parity ------------------------------------------------------------------------------------------------
for name, make in (("distinct", lambda: torch.randint(0, 1 << 62, (20_000_000,), dtype=torch.int64, device="cuda", generator=g)),
                   ("few dups", lambda: torch.cat([torch.randint(0, 1 << 62, (20_000_000,), dtype=torch.int64, device="cuda", generator=g)] * 1
                                                  + [torch.randint(0, 1 << 62, (1000,), dtype=torch.int64, device="cuda", generator=g)] * 3)),
                   ("dup x20", lambda: torch.randint(0, 1 << 62, (1_000_000,), dtype=torch.int64, device="cuda", generator=g).repeat(20))):
    if skip_parity:
        break
    v = make()
    ek, ec = torch.unique(v, sorted=True, return_counts=True)
    for mode in (1, 2, 0):
        set_mode(mode)
        k, c = ops.count_sparse(HArray(dev=v.clone()), key_bits=62, consume=True)
        ok = k.dev().numel() == ek.numel() and bool((k.dev() == ek).all()) and bool((c.dev() == ec).all())
        print("parity %-9s mode %d: %s (distinct %d)" % (name, mode, "OK" if ok else "MISMATCH", ek.numel()), flush=True)
    del v, ek, ec

# ---- timing ---------------------------------------------------------------------------------------------------
keys = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device="cuda", generator=g)
bits2 = 9 if n <= 3_200_000_000 else 10
a, off1 = ops.radix_partition(keys, None, 1, 52, 10)
b, off2 = ops.radix_partition(a, off1, 1 << 10, 52 - bits2, bits2)
del keys
nseg = 1 << (10 + bits2)
counts = torch.empty(n, dtype=torch.int64, device="cuda")
state = torch.empty(lib.bnpk_finish_state_words(nseg), dtype=torch.int64, device="cuda")
for mode in modes:
    set_mode(mode)
    times = []
    for rep in range(reps):
        nu, ov = C.c_int64(0), C.c_int(0)
        state[72:80] = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        st = lib.bnpk_finish_sorted(dev.ctx, ptr(b), n, ptr(off2), nseg, 52 - bits2, ptr(a), ptr(counts), ptr(state), None, 0, None, None, C.byref(nu), C.byref(ov), dev.stream())
        e1.record(); torch.cuda.synchronize()
        times.append(round(e0.elapsed_time(e1), 2))
    srt = bool((a[1:nu.value] > a[:nu.value - 1]).all()) if nu.value > 1 else True
    ph = state[72:80].tolist()
    tot = max(sum(ph), 1)
    print("   phases (share of cycles between barriers): " + " ".join("%.3f" % (x / tot) for x in ph) + "  ticks/bucket %.0f" % (tot / nseg))
    print("mode %d: finish ms %s status %d n_unique %d overflow %d sorted %s sum_counts %d header %s"
          % (mode, times, st, nu.value, ov.value, srt, int(counts[:nu.value].sum()), state[:4].tolist()), flush=True)
set_mode(0)
