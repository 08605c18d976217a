"""join_lines (gathered and ungathered fields) on shapes of tens of megabytes — beyond scripts/exp/fuzz_parity.py's"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_parity as T
from bionumpy_amd.ops import get_ops
ops = get_ops()
bad = 0
for seed, rows, max_len in ((101, 300000, 160), (102, 150000, 400), (103, 600000, 40), (104, 20, 3000000)):
    try:
        T.test_join_lines(ops, seed, rows, max_len)
        print("ok", seed, rows, max_len)
    except AssertionError:
        print("MISMATCH", seed, rows, max_len); bad += 1
print("fuzz_join:", bad, "mismatches")
