"""the parity tests of the kernels touched late in round 4, over many more seeds and shapes than the suite's fixed cases"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_parity as T
from bionumpy_amd.ops import get_ops
ops = get_ops()
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
bad = 0
t0, n, rng = time.time(), 0, np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
while time.time() - t0 < seconds:
    seed = int(rng.integers(10, 1 << 30))
    rows = int(rng.choice([1, 2, 17, 300, 5000, 40000, 150000]))
    cases = [(T.test_gather_encode_rows_between_other_text, (seed, max(rows, 8), int(rng.integers(1, 60)), int(rng.integers(61, 400)), bool(rng.integers(0, 2)))),
             (T.test_gather_encode_and_kmers, (seed, rows, int(rng.choice([3, 40, 200, 1000])))),
             (T.test_reverse_complement_kernels, (seed, rows, int(rng.choice([1, 20, 151, 700])))),
             (T.test_row_reductions, (seed, rows, int(rng.choice([1, 7, 160, 3000])))),
             (T.test_join_lines, (seed, min(rows, 5000), int(rng.choice([0, 5, 160, 2000])))),
             (T.test_match_windows, (seed, min(rows, 60000), int(rng.choice([1, 50, 151, 600])), int(rng.choice([1, 2, 3, 7, 31, 40]))))]
    for f, a in cases:
        try:
            f(ops, *a)
        except AssertionError:
            print("MISMATCH", f.__name__, a)
            bad += 1
    n += 1
print("fuzz_parity: %d rounds of 6 tests, %d mismatches" % (n, bad))
