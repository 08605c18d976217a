"""rc_packed under the conditions of the fuzz that found it: the same sequence of tests (seed 1), with the reverse complement
instrumented — for every mismatching case: which tiles / iterations / lanes, what the bad words hold.
    (cd scripts/bin/<variant> && python scripts/exp/rc_repro2.py [seconds])"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle
import test_gpu_parity as T
from bionumpy_amd.device import HArray
from bionumpy_amd.ops import get_ops
ops = get_ops()
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
skip_pre = len(sys.argv) > 2 and sys.argv[2] == "alone"


def rc_case(seed, n_rows, max_len):
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, max_len + 1, size=n_rows).astype(np.int64)
    lens[rng.integers(0, n_rows, size=max(1, n_rows // 10))] = 0
    if n_rows > 3:
        lens[1], lens[2] = 32, 64
    total = int(lens.sum())
    codes = rng.integers(0, 4, size=total).astype(np.uint8)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    packed = ops.pack_codes(HArray(host=codes))
    out = ops.reverse_complement_packed(packed, HArray(host=offsets), n_rows, total)
    exp = oracle.reverse_complement(codes, lens)
    E = ops.pack_codes(HArray(host=exp)).host().view(np.uint64)
    G = out.host().view(np.uint64)
    n_words = total // 32 + 1
    bw = np.flatnonzero(G[:n_words] != E[:n_words])
    if not bw.size:
        return 0
    tiles = np.unique(bw // 1024)
    print("MISMATCH", (seed, n_rows, max_len), "total", total, "tiles", -(-n_words // 1024), "bad words", bw.size, "bad tiles", tiles.size, "first", tiles[:8].tolist(), "last", tiles[-3:].tolist())
    print("  per iteration:", np.bincount((bw % 1024) // 256, minlength=4).tolist(), " per wave:", np.bincount((bw % 256) // 64, minlength=4).tolist(),
          " distinct lanes", np.unique(bw % 256).size)
    # is a bad word the right word of ANOTHER row alignment?  compare with expected words nearby, and classify by boundary
    kinds = {"zero": 0, "other_it": 0, "row_boundary": 0, "inside_row": 0}
    for w in bw[:3000]:
        g = G[w]
        inside = np.searchsorted(offsets, w * 32, side="right") == np.searchsorted(offsets, w * 32 + 31, side="right")
        if g == 0: kinds["zero"] += 1
        elif any(0 <= w + d < n_words and g == E[w + d] for d in (-768, -512, -256, 256, 512, 768)): kinds["other_it"] += 1
        elif not inside: kinds["row_boundary"] += 1
        else: kinds["inside_row"] += 1
    print("  kinds:", kinds)
    for w in bw[:4]:
        w = int(w); r = int(np.searchsorted(offsets, w * 32, side="right") - 1)
        rows_in_tile = int(np.searchsorted(offsets, (w // 1024 + 1) * 32768, side="right") - np.searchsorted(offsets, (w // 1024) * 32768, side="right"))
        print("  word", w, "tile", w // 1024, "it", (w % 1024) // 256, "lane", w % 256, "row", r, "span", int(offsets[r]), int(offsets[r + 1]), "rows in tile", rows_in_tile,
              "got %016x exp %016x xor %016x" % (int(G[w]), int(E[w]), int(G[w] ^ E[w])))
    # again, right away: does the same call fail the same way?
    out2 = ops.reverse_complement_packed(packed, HArray(host=offsets), n_rows, total)
    G2 = out2.host().view(np.uint64)
    bw2 = np.flatnonzero(G2[:n_words] != E[:n_words])
    print("  repeated at once: bad words", bw2.size, "same set" if np.array_equal(bw, bw2) else "different set", "same values" if bw2.size == bw.size and np.array_equal(G[bw], G2[bw2]) else "")
    return 1


t0, n, bad, rng = time.time(), 0, 0, np.random.default_rng(1)
while time.time() - t0 < seconds:
    seed = int(rng.integers(10, 1 << 30))
    rows = int(rng.choice([1, 2, 17, 300, 5000, 40000, 150000]))
    a1 = (seed, max(rows, 8), int(rng.integers(1, 60)), int(rng.integers(61, 400)), bool(rng.integers(0, 2)))
    a2 = (seed, rows, int(rng.choice([3, 40, 200, 1000])))
    a3 = (seed, rows, int(rng.choice([1, 20, 151, 700])))
    rng.choice([1, 7, 160, 3000]); rng.choice([0, 5, 160, 2000]); rng.choice([1, 50, 151, 600]); rng.choice([1, 2, 3, 7, 31, 40])
    if not skip_pre:
        T.test_gather_encode_rows_between_other_text(ops, *a1)
        T.test_gather_encode_and_kmers(ops, *a2)
    bad += rc_case(*a3)
    n += 1
print("rc_repro2: %d rounds, %d mismatching cases" % (n, bad))
