import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bionumpy_amd.device import HArray
from bionumpy_amd.ops import get_ops
ops = get_ops()
g = torch.Generator(device="cuda")
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
    g.manual_seed(1000 + it)
    n = 2_000_003 + 1000 * (it % 7)
    v = torch.randint(0, 300_000, (n,), dtype=torch.int64, device="cuda", generator=g) * 7919
    keys, counts = ops.count_sparse(HArray(dev=v), key_bits=62)
    ek, ec = torch.unique(v, return_counts=True)
    k, c = keys.dev(), counts.dev()
    if not (k.numel() == ek.numel() and bool((k == ek).all()) and bool((c == ec).all())):
        bad += 1
        print("BAD it", it, k.numel(), ek.numel(), int(c.sum()), n)
        if k.numel() == ek.numel():
            idx = ((k != ek) | (c != ec)).nonzero().flatten()
            print("  mismatches", idx.numel(), idx[:8].tolist(), idx[-3:].tolist())
print("stress_small done, bad =", bad)
