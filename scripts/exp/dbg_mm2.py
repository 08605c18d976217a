import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bionumpy_amd as bnp
from bionumpy_amd.device import Device, HArray
from bionumpy_amd.ops import get_ops
from bionumpy_amd._native import lib
ops = get_ops(); dev = Device.get()
p = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "big.fq.gz")
whole = bnp.change_encoding(bnp.open(p).read().sequence, bnp.DNAEncoding)
h = np.asarray(bnp.get_kmers(whole, 31).raw().ravel())
ek, ec = np.unique(h, return_counts=True)
print("expected distinct", ek.size, "n", h.size)
ops.keep_finish_state = True
for mode in (0, 1, 2, 3, 4, 5, 0, 0, 0):
    lib.bnpk_set_option(dev.ctx, b"finish_mode", mode)
    for rep in range(3):
        k, c = ops.count_sparse(HArray(host=h.copy()), key_bits=62)
        ok = np.array_equal(k.host(), ek) and np.array_equal(c.host(), ec)
        st = ops.last_finish_state
        print("mode", mode, "rep", rep, "distinct", k.size, "ok", ok, "hdr", st[:8].tolist(), "probe", st[8:12].tolist())
        if not ok:
            kk = k.host(); m = min(kk.size, ek.size); d = np.flatnonzero(kk[:m] != ek[:m])
            print("   first diff at", d[:3], kk[d[0]-1:d[0]+2] if d.size else None, ek[d[0]-1:d[0]+2] if d.size else None)
print("---- partition check")
import torch
t = torch.from_numpy(h.copy()).cuda()
levels = ops.radix_plan(h.size, 62)
print("levels", levels)
out, offsets = ops.radix_partition(t, None, 1, 62 - levels[0], levels[0])
off = offsets.cpu().numpy(); part = out.cpu().numpy()
print("bucket sizes", np.diff(off).tolist())
ids = part >> (62 - levels[0])
bad = [b for b in range(len(off) - 1) if not np.all(ids[off[b]:off[b + 1]] == b)]
print("buckets holding foreign keys:", bad, "multiset equal:", np.array_equal(np.sort(part), np.sort(h)))
census = torch.empty(2 + 3 * 256, dtype=torch.int64, device="cuda")
from bionumpy_amd.device import ptr
print("census rc", lib.bnpk_bucket_census(dev.ctx, ptr(offsets), len(off) - 1, 8192, 256, ptr(census), dev.stream()), census[:8].tolist())
