import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import oracle
from bionumpy_amd.device import HArray
from bionumpy_amd.ops import get_ops
ops = get_ops()
rng = np.random.default_rng(1)
n_rows, max_len = 1000, 40
lengths = rng.integers(0, max_len, size=n_rows).astype(np.int64)
text = rng.choice(np.frombuffer(b"ACGTacgt", dtype=np.uint8), size=int(lengths.sum()) + n_rows)
starts = np.concatenate(([0], np.cumsum(lengths[:-1] + 1))).astype(np.int64)
offsets, total = ops.row_offsets(HArray(host=lengths), 1)
import sys as _s
flag = int(_s.argv[1]) if len(_s.argv) > 1 else 0
got = ops.gather_rows(HArray(host=text), HArray(host=starts), offsets, n_rows, total, flag).host()
exp = oracle.gather_rows(text, starts, lengths)
bad = np.flatnonzero(got != exp)
print("total", total, "mismatches", bad.size, "first", bad[:20], "chunks", np.unique(bad // 16)[:20])
off = offsets.host()
for b in bad[:3]:
    r = np.searchsorted(off, b, side="right") - 1
    print("pos", b, "row", r, "row range", off[r], off[r+1], "got", got[b-4:b+12], "exp", exp[b-4:b+12])
