"""Reproducer for the rc_packed_kernel miscompile (NOTES round 4 -> round 5): run one library variant
(scripts/exp/make_variant.sh <name> "-DBNPK_RCP_UNROLL=4 ...") on the failing shape and describe the mismatches.
    python scripts/bin/<name>/scripts/exp/rc_repro.py [pre]
pre = "poison": run a kernel that fills registers with junk first (torch randn matmul); "quiet": nothing."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, oracle
from bionumpy_amd.device import HArray
from bionumpy_amd.ops import get_ops
ops = get_ops()
pre = sys.argv[1] if len(sys.argv) > 1 else "quiet"
for seed, n_rows, max_len in ((287332572, 40000, 700), (7, 120000, 151), (9, 300, 100000)):
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, max_len + 1, size=n_rows).astype(np.int64)
    lens[rng.integers(0, n_rows, size=max(1, n_rows // 10))] = 0
    lens[1], lens[2] = 32, 64
    total = int(lens.sum())
    codes = rng.integers(0, 4, size=total).astype(np.uint8)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    packed = ops.pack_codes(HArray(host=codes))
    exp = oracle.reverse_complement(codes, lens)
    E = ops.pack_codes(HArray(host=exp)).host().view(np.uint64)
    for rep in range(3):
        if pre == "poison":
            import torch
            a = torch.randn(4096, 4096, device="cuda"); b = (a @ a).sum().item()
        out = ops.reverse_complement_packed(packed, HArray(host=offsets), n_rows, total)
        G = out.host().view(np.uint64)
        n_words = total // 32 + 1
        bw = np.flatnonzero(G[:n_words] != E[:n_words])
        print("shape", (seed, n_rows, max_len), "rep", rep, "total", total, "tiles", -(-n_words // 1024), "bad words", bw.size)
        if not bw.size:
            continue
        tiles = np.unique(bw // 1024)
        print("  first bad tile", tiles[0], "bad tiles", tiles.size, "of", -(-n_words // 1024), "first 12:", tiles[:12].tolist())
        print("  bad words per iteration (it = word%1024//256):", np.bincount((bw % 1024) // 256, minlength=4).tolist())
        lanes = bw % 256
        print("  bad lanes: distinct", np.unique(lanes).size, "whole-wave groups:", np.bincount(lanes // 64, minlength=4).tolist())
        kinds = {"zero": 0, "other_it": 0, "prev_tile": 0, "row_boundary": 0, "other": 0}
        for w in bw[:2000]:
            g = G[w]
            inside = np.searchsorted(offsets, w * 32, side="right") == np.searchsorted(offsets, w * 32 + 31, side="right")
            if g == 0: kinds["zero"] += 1
            elif any(0 <= w + d < n_words and g == E[w + d] for d in (-768, -512, -256, 256, 512, 768)): kinds["other_it"] += 1
            elif w >= 1024 and g == E[w - 1024]: kinds["prev_tile"] += 1
            elif not inside: kinds["row_boundary"] += 1
            else: kinds["other"] += 1
        print("  what the bad words hold:", kinds)
        w = int(bw[0]); r = int(np.searchsorted(offsets, w * 32, side="right") - 1)
        print("  first bad word", w, "tile", w // 1024, "it", (w % 1024) // 256, "lane", w % 256, "row", r, "row span", int(offsets[r]), int(offsets[r + 1]),
              "got %016x exp %016x xor %016x" % (int(G[w]), int(E[w]), int(G[w] ^ E[w])))
