"""Level 2 without its histogram pass (bnpk_radix_partition_claimed) against hist + scatter (bnpk_radix_partition): the same keys
in every bucket?  how long?     python scripts/exp/exp_claim.py [n_keys] [mode]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bionumpy_amd._native import lib, check
from bionumpy_amd.device import Device, HArray, ptr
from bionumpy_amd.ops import get_ops
ops = get_ops(); dev = Device.get()
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 300_000_000
mode = sys.argv[2] if len(sys.argv) > 2 else "uniform"
key_bits, b1, b2 = 62, 10, 10
STRIDE, CAP_LO = int(lib.bnpk_claimed_stride()), int(lib.bnpk_claimed_cap_lo())
g = torch.Generator(device="cuda"); g.manual_seed(5)
if mode == "uniform":
    keys = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device="cuda", generator=g)
else:                                                     # few distinct keys, many copies: some buckets over the capacity
    distinct = torch.randint(0, 1 << 62, (max(n // 60, 1),), dtype=torch.int64, device="cuda", generator=g)
    keys = distinct[torch.randint(0, distinct.numel(), (n,), device="cuda", generator=g)]
# level 1 by the existing kernels
l1, off1 = ops.radix_partition(keys, None, 1, key_bits - b1, b1)
del keys
n_seg = 1 << b1
n_b = n_seg << b2


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r


old_out = torch.empty(n, dtype=torch.int64, device="cuda")
ms_old, (l2, off2) = timed(lambda: ops.radix_partition(l1, off1, n_seg, key_bits - b1 - b2, b2, old_out))
buckets = torch.empty(n_b * STRIDE, dtype=torch.int64, device="cuda")
fill = torch.empty(2 * n_b, dtype=torch.int32, device="cuda")
bag_cap = max(n // 8, 1 << 20)
bag = torch.empty(bag_cap, dtype=torch.int64, device="cuda")
bag_fill = torch.zeros(1, dtype=torch.int64, device="cuda")


def claimed():
    check(lib.bnpk_radix_partition_claimed(dev.ctx, ptr(l1), n, ptr(off1), n_seg, key_bits - b1 - b2, b2, ptr(buckets), ptr(fill), ptr(bag), bag_cap,
                                           ptr(bag_fill), dev.stream()), dev.ctx)
    offs = torch.empty(n_b + 1, dtype=torch.int64, device="cuda")
    check(lib.bnpk_claimed_offsets(dev.ctx, ptr(fill), n_b, ptr(offs), dev.stream()), dev.ctx)
    return offs


ms_new, offs = timed(claimed)
print("n", n, mode, "| hist + scatter %.2f ms | claimed scatter (+ offsets) %.2f ms" % (ms_old, ms_new))
if os.environ.get("TIMING_ONLY"):
    dev.prof_enable(True); dev.prof_reset(); claimed(); ops.radix_partition(l1, off1, n_seg, key_bits - b1 - b2, b2, old_out); torch.cuda.synchronize()
    print("  ", {k: round(v["total_ms"], 2) for k, v in dev.prof_report().items()})
    sys.exit(0)
# ---- the same multiset in every bucket ----------------------------------------------------------------------------------
f = fill.view(n_b, 2).to(torch.int64)
lo, hi = f[:, 0].clamp(max=CAP_LO), f[:, 1].clamp(max=STRIDE - CAP_LO)
nb = int(bag_fill.item())
sizes_new = lo + hi
assert torch.equal(offs[1:] - offs[:-1], sizes_new) and int(offs[-1]) + nb == n, (int(offs[-1]), nb, n)
print("  bag", nb, "keys; fronts over the capacity:", int((f[:, 0] > CAP_LO).sum()), "tails over:", int((f[:, 1] > STRIDE - CAP_LO).sum()),
      "largest bucket", int(sizes_new.max()), "old largest", int((off2[1:] - off2[:-1]).max()))
shift = key_bits - b1 - b2
bview = buckets.view(n_b, STRIDE)
# every key lies in the bucket its top bits name, in both runs
step = 1 << 12
for a in range(0, n_b, step):
    part = bview[a:a + step]
    idx = torch.arange(STRIDE, device="cuda")[None, :]
    valid = (idx < lo[a:a + step, None]) | ((idx >= CAP_LO) & (idx < CAP_LO + hi[a:a + step, None]))
    want = torch.arange(a, min(a + step, n_b), device="cuda")[:, None]
    assert bool((((part >> shift) == want) | ~valid).all()), "a key in the wrong bucket (buckets %d..)" % a
# the multiset: sorted (buckets' keys + bag) == sorted input (checksums beyond 1e9 keys)
def sums(t):
    m = t * -7046029254386353131
    m = m ^ (m >> 29)
    return [int(t.numel()), int(t.sum().item()), int((t * t).sum().item()), int(m.sum().item())]
if n > 1_000_000_000:
    tot = [0, 0, 0, 0]
    for a in range(0, n_b, step):
        part = bview[a:a + step]
        idx = torch.arange(STRIDE, device="cuda")[None, :]
        valid = (idx < lo[a:a + step, None]) | ((idx >= CAP_LO) & (idx < CAP_LO + hi[a:a + step, None]))
        sel = part[valid]
        tot = [x + y for x, y in zip(tot, sums(sel))]
        del sel, valid
    tot = [x + y for x, y in zip(tot, sums(bag[:nb]))]
    ref = [0, 0, 0, 0]
    for a in range(0, n, 1 << 28):
        ref = [x + y for x, y in zip(ref, sums(l1[a:a + (1 << 28)]))]
    wrap = lambda v: [v[0]] + [x & ((1 << 64) - 1) for x in v[1:]]
    assert wrap(tot) == wrap(ref), (wrap(tot), wrap(ref))
    print("  parity: every key in its bucket, checksums of (buckets + bag) == input")
else:
    got = []
    for a in range(0, n_b, step):
        part = bview[a:a + step]
        idx = torch.arange(STRIDE, device="cuda")[None, :]
        valid = (idx < lo[a:a + step, None]) | ((idx >= CAP_LO) & (idx < CAP_LO + hi[a:a + step, None]))
        got.append(part[valid])
    got.append(bag[:nb])
    got = torch.cat(got)
    assert got.numel() == n
    a_sorted, b_sorted = torch.sort(got).values, torch.sort(l1).values
    assert torch.equal(a_sorted, b_sorted), "the claimed buckets + bag do not hold the input's keys"
    print("  parity: every key in its bucket, multiset of (buckets + bag) == input")
