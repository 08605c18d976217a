import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(1, "/root/repo"); sys.path.insert(2, "/root/repo/tests")
import numpy as np, torch
import oracle
from bionumpy_amd.device import HArray
from bionumpy_amd.ops import get_ops
ops = get_ops()
rng = np.random.default_rng(1)
n_rows, max_len = 1000, 40
lengths = rng.integers(0, max_len, size=n_rows).astype(np.int64)
text = rng.choice(np.frombuffer(b"ACGTacgt", dtype=np.uint8), size=int(lengths.sum()) + n_rows)
starts = np.concatenate(([0], np.cumsum(lengths[:-1] + 1))).astype(np.int64)
mode = sys.argv[1]
print("mode", mode, flush=True)
h_text, h_starts, h_lens = HArray(host=text), HArray(host=starts), HArray(host=lengths)
offsets, total = ops.row_offsets(h_lens, 1)
torch.cuda.synchronize(); print("row_offsets ok", total, int(offsets.host()[-1]), flush=True)
if mode == "encode_first":
    ops.gather_encode_dna(h_text, h_starts, offsets, n_rows, total, want_codes=True, want_packed=True)
    torch.cuda.synchronize(); print("gather_encode ok", flush=True)
got = ops.gather_rows(h_text, h_starts, offsets, n_rows, total, 0)
torch.cuda.synchronize(); print("gather_rows launched ok", flush=True)
exp = oracle.gather_rows(text, starts, lengths)
print("equal", np.array_equal(got.host(), exp))
