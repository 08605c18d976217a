import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bionumpy_amd._native import lib
from bionumpy_amd.device import Device, HArray
from bionumpy_amd.ops import get_ops
ops = get_ops(); dev = Device.get()
rng = np.random.default_rng(20260929)
def keys_of(n, bits, shape):
    top = 1 << bits
    if shape == 0:
        k = rng.integers(0, top, size=n, dtype=np.int64)
    elif shape == 1:
        k = np.minimum((rng.random(n) ** 4 * float(top)).astype(np.int64), top - 1)
    elif shape == 2:
        k = rng.integers(0, min(top, 40), size=n, dtype=np.int64)
    elif shape == 3:
        k = rng.integers(0, top, size=n, dtype=np.int64)
        hot = rng.integers(0, top, size=3, dtype=np.int64)
        k[rng.random(n) < 0.3] = hot[0]
        k[rng.random(n) < 0.05] = hot[1]
    elif shape == 4:
        k = (rng.integers(0, top, dtype=np.int64) & ~((top >> 12) - 1 if top >> 12 else 0)) | rng.integers(0, max(top >> 12, 1), size=n, dtype=np.int64)
    else:
        k = rng.integers(0, top, size=max(n // 20, 1), dtype=np.int64)[rng.integers(0, max(n // 20, 1), size=n)]
    return k.astype(np.int64)
for rounds in range(34):
    n = int(rng.choice([1, 7, 3000, 70_000, 1_300_000, 4_000_000]))
    bits = int(rng.choice([20, 33, 50, 62]))
    shape = int(rng.integers(0, 6))
    claim, direct = int(rng.integers(0, 2)), int(rng.integers(0, 2))
    keys = keys_of(n, bits, shape)
    n_rows = int(rng.choice([1, 3, 17, 1000, 1024, 1025, 40_000]))
    rows = np.sort(rng.integers(0, n_rows, size=n)).astype(np.int64)
print(rounds, n, bits, shape, claim, direct, n_rows)
want, mult = np.unique(np.stack([keys, rows]), axis=1, return_counts=True)
top10 = (keys >> 52)
print("largest L1 buckets", np.sort(np.bincount(top10, minlength=1024))[-3:], "empty L1 buckets", int((np.bincount(top10, minlength=1024) == 0).sum()))
for d in (0, 1):
    lib.bnpk_set_option(dev.ctx, b"index_pairs", d)
    got = ops.unique_pairs(HArray(host=keys.copy()), HArray(host=rows), key_bits=bits, n_values=n_rows, with_counts=True)
    gk, gr, gm = got[0].host(), got[1].host(), got[2].host()
    ok = np.array_equal(gk, want[0]) and np.array_equal(gr, want[1]) and np.array_equal(gm, mult)
    print("direct", d, "ok", ok, gk.size, want.shape[1])
    if not ok:
        bad = np.flatnonzero((gk != want[0]) | (gr != want[1]) | (gm != mult))
        print(" first bad", bad[0], "last bad", bad[-1], "n bad", bad.size)
        o = np.lexsort((gr, gk))
        print(" same pairs as a multiset:", np.array_equal(gk[o], want[0]) and np.array_equal(gr[o], want[1]) and np.array_equal(gm[o], mult))
        lowmask = (1 << 52) - 1
        print(" low parts as a multiset equal:", np.array_equal(np.sort(gk & lowmask), np.sort(want[0] & lowmask)))
        gt, wt = gk >> 52, want[0] >> 52
        print(" top-bit histogram diff (bucket: got - want):", [(int(b), int(x)) for b, x in enumerate(np.bincount(gt, minlength=1024) - np.bincount(wt, minlength=1024)) if x][:10])
        j = bad[0]
        print(" around first bad: got tops", gt[j-2:j+3].tolist(), "want tops", wt[j-2:j+3].tolist())
