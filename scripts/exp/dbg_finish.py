import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bionumpy_amd.device import HArray
from bionumpy_amd.ops import get_ops
ops = get_ops()
for seed in range(12):
  rng = np.random.default_rng(seed)
  for n, d in ((2_000_003, 300_000), (900_000, 300_000), (5000, 100), (200_000, 1 << 40), (3_000_000, 1 << 40)):
      v = (rng.integers(0, d, size=n).astype(np.int64) * 7919) & ((1 << 62) - 1)
      keys, counts = ops.count_sparse(HArray(host=v), key_bits=62)
      ek, ec = np.unique(v, return_counts=True)
      k, c = keys.host(), counts.host()
      ok = k.size == ek.size and np.array_equal(k, ek) and np.array_equal(c, ec)
      print(n, d, "ok" if ok else "BAD", k.size, ek.size)
      if not ok and k.size == ek.size:
            bad = np.flatnonzero((k != ek) | (c != ec))
            print(" mismatches", bad.size, "first", bad[:10], "sorted", bool(np.all(np.diff(k) > 0)), "sum", c.sum(), n)
            i = bad[0]; print(k[i-2:i+3], ek[i-2:i+3], c[i-2:i+3], ec[i-2:i+3])
