"""the other primitives of tests/test_gpu_parity.py at random sizes well beyond the suite's (scans, newline positions, merges,
partition levels, sparse counts through every path, motif scores)"""
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
    big = int(rng.choice([3_000_001, 9_999_999, 33_554_433, 50_000_000]))
    na, nb = int(rng.integers(1, 8_000_000)), int(rng.integers(1, 8_000_000))
    cases = [(T.test_newline_scan_matches_flatnonzero, (big + seed % 17,)),
             (T.test_exclusive_scan, (big // 2 + seed % 5,)),
             (T.test_merge_add_of_sparse_histograms, (na, nb, int(rng.integers(0, min(na, nb) + 1)))),
             (T.test_radix_partition_levels, (seed, int(rng.choice([700_000, 5_000_000, 20_000_000])), int(rng.choice([62, 42, 30])),
                                              [[10, 10], [10], [11, 9], [7, 6, 5]][int(rng.integers(0, 4))])),
             (T.test_count_sparse_radix_path, (seed, int(rng.choice([400_000, 3_000_000, 12_000_000])), int(rng.choice([62, 42, 30, 20])),
                                               int(rng.choice([1, 2, 3, 50])))),
             (T.test_pwm_scores, (seed, int(rng.choice([300, 5000, 60000])), int(rng.choice([50, 151, 600])), int(rng.choice([1, 6, 12, 31]))))]
    for f, a in cases:
        try:
            f(ops, *a)
        except AssertionError:
            print("MISMATCH", f.__name__, a)
            bad += 1
    n += 1
print("fuzz_parity2: %d rounds of 6 tests, %d mismatches" % (n, bad))
