"""Stress: count_sparse vs torch.unique on device-generated keys, many shapes (catches intermittent races)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bionumpy_amd.device import HArray
from bionumpy_amd.ops import get_ops
ops = get_ops()
g = torch.Generator(device="cuda")
bad = 0
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for it in range(rounds):
    g.manual_seed(it)
    n = [1000, 50_000, 2_000_003, 9_000_000, 40_000_000][it % 5] + it
    d = [300, 1 << 18, 1 << 21, 1 << 40, 1 << 61][(it // 5) % 5]
    v = torch.randint(0, d, (n,), dtype=torch.int64, device="cuda", generator=g) * 7919 % (1 << 62)
    keys, counts = ops.count_sparse(HArray(dev=v), key_bits=62)
    ek, ec = torch.unique(v, return_counts=True)
    k, c = keys.dev(), counts.dev()
    ok = k.numel() == ek.numel() and bool((k == ek).all()) and bool((c == ec).all())
    if not ok:
        bad += 1
        print("BAD it", it, "n", n, "d", d, k.numel(), ek.numel())
        if k.numel() == ek.numel():
            idx = ((k != ek) | (c != ec)).nonzero().flatten()
            print("  mismatches", idx.numel(), idx[:8].tolist(), "sum", int(c.sum()), n)
print("stress done, bad =", bad)
