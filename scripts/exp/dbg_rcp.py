import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, oracle
from bionumpy_amd.device import HArray
from bionumpy_amd.ops import get_ops
ops = get_ops()
seed, n_rows, max_len = 287332572, 40000, 700
rng = np.random.default_rng(seed)
lens = rng.integers(0, max_len + 1, size=n_rows).astype(np.int64)
lens[rng.integers(0, n_rows, size=max(1, n_rows // 10))] = 0
lens[1], lens[2] = 32, 64
total = int(lens.sum())
codes = rng.integers(0, 4, size=total).astype(np.uint8)
offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
packed = ops.pack_codes(HArray(host=codes))
out = ops.reverse_complement_packed(packed, HArray(host=offsets), n_rows, total)
got = ops.unpack_codes(out, total).host()
exp = oracle.reverse_complement(codes, lens)
bad = np.flatnonzero(got != exp)
print("total", total, "bad", bad.size, "first", bad[:10], "words", np.unique(bad // 32)[:20], "tiles", np.unique(bad // 32768)[:20])
if bad.size:
    w = bad[0] // 32
    r = np.searchsorted(offsets, bad[0], side="right") - 1
    print("word", w, "it", (w % 1024) // 256, "lane", w % 256, "row", r, "row range", offsets[r], offsets[r + 1], "rows in tile", np.searchsorted(offsets, (w // 1024 + 1) * 32768, side="right") - np.searchsorted(offsets, (w // 1024) * 32768, side="right"))
    # runs of bad words
    bw = np.unique(bad // 32)
    print("bad words count", bw.size, "per tile:", np.bincount(bw // 1024)[np.unique(bw // 1024)][:20])
    print("bad words mod 1024:", sorted(set((bw % 1024).tolist()))[:40])
