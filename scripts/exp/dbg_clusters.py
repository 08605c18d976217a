import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, ctypes as C
import test_finish_modes as t
from bionumpy_amd import ops as ops_mod
from bionumpy_amd._native import lib
from bionumpy_amd.device import Device, ptr
env = (ops_mod.get_ops(), lib, Device.get(), ptr, torch)
rng = np.random.default_rng(6)
for run in (2, 3, 17, 60, 63, 64, 65, 66, 100, 300, 2500):
    tops = (np.arange(run, dtype=np.int64) * 10 + 7000) << 46
    keys = rng.permutation(np.concatenate([tops | (np.arange(run, dtype=np.int64) * 2654435761 & 0xFFFFF)] * 3))
    try:
        print(run, t._direct(env, keys, 0))
    except AssertionError as e:
        print(run, "MISMATCH", e)
