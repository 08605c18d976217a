"""Is the unrolled rc_packed_kernel reading a register it never wrote?  Every register of every SIMD is filled with a value
(scripts/exp/scrub/libscrub.so) right before the call: zeros, then a pattern.
    (cd scripts/bin/rc_u4 && python scripts/exp/rc_repro3.py)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle
from bionumpy_amd.device import HArray, Device
from bionumpy_amd.ops import get_ops
ops = get_ops(); dev = Device.get()
scrub = C.CDLL(os.path.join(REPO, "scripts", "exp", "scrub", "libscrub.so"))
sink = torch.zeros(4, dtype=torch.int32, device="cuda")


def fill(val):
    rc = scrub.scrub(C.c_uint(val), C.c_void_p(sink.data_ptr()), dev.stream())
    assert rc == 0, rc


cases = [(772671723, 150000, 151), (1016854741, 40000, 700), (569019405, 150000, 700)]
for seed, n_rows, max_len in cases:
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, max_len + 1, size=n_rows).astype(np.int64)
    lens[rng.integers(0, n_rows, size=max(1, n_rows // 10))] = 0
    lens[1], lens[2] = 32, 64
    total = int(lens.sum())
    codes = rng.integers(0, 4, size=total).astype(np.uint8)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    packed = ops.pack_codes(HArray(host=codes))
    E = ops.pack_codes(HArray(host=oracle.reverse_complement(codes, lens))).host().view(np.uint64)
    d_off = HArray(host=offsets); d_off.dev()
    n_words = total // 32 + 1
    labels = (("none", None), ("zeros", 0), ("0x5a5a5a5a", 0x5a5a5a5a), ("0xffffffff", 0xffffffff), ("0x00000001", 1), ("zeros", 0), ("none", None))
    if len(sys.argv) > 1 and sys.argv[1] == "quick":
        labels = (("none", None),)
    for label, val in labels:
        for rep in range(3 if len(labels) == 1 else 2):
            if val is not None:
                fill(val)
            out = ops.reverse_complement_packed(packed, d_off, n_rows, total)
            G = out.host().view(np.uint64)
            bw = np.flatnonzero(G[:n_words] != E[:n_words])
            tiles = np.unique(bw // 1024)
            print((seed, n_rows, max_len), "registers:", label, "rep", rep, "bad words", bw.size, "bad tiles", tiles.size, "of", -(-n_words // 1024),
                  "first", tiles[:5].tolist(), "per it", np.bincount((bw % 1024) // 256, minlength=4).tolist() if bw.size else "")
