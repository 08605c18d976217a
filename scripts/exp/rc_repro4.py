"""Where do the bad tiles of the unrolled rc_packed_kernel begin when the kernel may only use some of the CUs?  (a stream with a
CU mask: hipExtStreamCreateWithCUMask) — 'the 257th workgroup' = the second workgroup on a CU, or the tile index?
    (cd scripts/bin/rc_u4 && python scripts/exp/rc_repro4.py)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle
from bionumpy_amd._native import lib, check
from bionumpy_amd.device import HArray, Device, ptr
from bionumpy_amd.ops import get_ops
ops = get_ops(); dev = Device.get()
hip = C.CDLL("libamdhip64.so")
for name in ("libamdhip64.so.7", "libamdhip64.so"):
    try:
        hip = C.CDLL(name); break
    except OSError:
        pass


def masked_stream(n_cus):
    words = (C.c_uint32 * 8)(*[0] * 8)
    for i in range(n_cus):
        words[i // 32] |= 1 << (i % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return s


seed, n_rows, max_len = 569019405, 150000, 700
rng = np.random.default_rng(seed)
lens = rng.integers(0, max_len + 1, size=n_rows).astype(np.int64)
lens[rng.integers(0, n_rows, size=max(1, n_rows // 10))] = 0
lens[1], lens[2] = 32, 64
total = int(lens.sum())
codes = rng.integers(0, 4, size=total).astype(np.uint8)
offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
packed = ops.pack_codes(HArray(host=codes)).dev()
E = ops.pack_codes(HArray(host=oracle.reverse_complement(codes, lens))).host().view(np.uint64)
d_off = HArray(host=offsets).dev()
n_words = total // 32 + 1
torch.cuda.synchronize()
for n_cus in (256, 128, 64, 32, 8, 256):
    s = masked_stream(n_cus)
    for rep in range(2):
        out = torch.zeros(total // 32 + 2, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        check(lib.bnpk_reverse_complement_packed(dev.ctx, ptr(packed), ptr(d_off), n_rows, total, ptr(out), s))
        assert hip.hipStreamSynchronize(s) == 0
        G = out.cpu().numpy().view(np.uint64)
        bw = np.flatnonzero(G[:n_words] != E[:n_words])
        tiles = np.unique(bw // 1024)
        print("CUs", n_cus, "rep", rep, "bad words", bw.size, "bad tiles", tiles.size, "of", -(-n_words // 1024), "first bad tiles", tiles[:6].tolist(),
              "per it", np.bincount((bw % 1024) // 256, minlength=4).tolist() if bw.size else "")
