"""Debug aid: bnpk_finish_sorted in finish_mode 5 on hand-made buckets; prints where status words / output diverge."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bionumpy_amd.device import Device, ptr
from bionumpy_amd.ops import get_ops
from bionumpy_amd._native import lib
ops = get_ops(); dev = Device.get()
top_bits = int(sys.argv[1]) if len(sys.argv) > 1 else 6
per = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
nb = 1 << top_bits
rng = np.random.default_rng(1)
keys = rng.integers(0, 1 << 62, size=nb * per, dtype=np.int64)
keys = np.concatenate([keys, keys[: nb * 3]])
ids = keys >> (62 - top_bits)
part = keys[np.argsort(ids, kind="stable")]
off = np.zeros(nb + 1, dtype=np.int64); off[1:] = np.cumsum(np.bincount(ids, minlength=nb))
ek, ec = np.unique(keys, return_counts=True)
expD = np.array([np.unique(part[off[i]:off[i + 1]]).size for i in range(nb)])
lib.bnpk_set_option(dev.ctx, b"finish_mode", 5)
work, d_off = torch.from_numpy(part.copy()).cuda(), torch.from_numpy(off).cuda()
out_k, out_c = torch.zeros_like(work), torch.zeros_like(work)
words = lib.bnpk_finish_state_words(nb)
state = torch.zeros(words, dtype=torch.int64, device="cuda")
nu, ov = C.c_int64(0), C.c_int(0)
rc = lib.bnpk_finish_sorted(dev.ctx, ptr(work), keys.size, ptr(d_off), nb, 62 - top_bits, ptr(out_k), ptr(out_c), ptr(state),
                            None, 0, None, None, C.byref(nu), C.byref(ov), dev.stream())
torch.cuda.synchronize()
st = state.cpu().numpy()
print("rc", rc, "n_unique", nu.value, "expected", ek.size, "overflow", ov.value, "header", st[:8])
FS_FAST = 112
meta = st[FS_FAST + nb + 1:].view(np.uint32)[:nb]
print("status-1 first 16:", meta[:16].astype(np.int64) - 1)
print("expected D first 16:", expD[:16])
bad = np.flatnonzero(meta.astype(np.int64) - 1 != expD)
print("buckets with wrong D:", bad[:20], bad.size)
gk, gc = out_k.cpu().numpy(), out_c.cpu().numpy()
m = min(ek.size, gk.size)
d = np.flatnonzero(gk[:m] != ek[:m])
print("first key mismatch at", d[:5], "of", ek.size)
if d.size:
    i = d[0]; print("got", gk[i - 2:i + 4], "want", ek[i - 2:i + 4], "bucket", ek[i] >> (62 - top_bits), "prefix want", np.cumsum(expD)[:8])
dc = np.flatnonzero(gc[:m] != ec[:m]); print("count mismatches", dc.size, dc[:5])
