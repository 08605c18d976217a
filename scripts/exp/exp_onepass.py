"""Experiment: the one-pass FASTQ decode with its look-backs switched off (BNPK_FQ1_ABL: 8 no base look-back, 16 no line look-back —
first lines from the two-pass census —, 4 static tiles) to see what the look-backs cost and what the rest costs."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bionumpy_amd import ops as O
from bionumpy_amd._native import lib
from bionumpy_amd.device import Device, ptr

ops = O.get_ops(); dev = Device.get()
text = ops.synth_fastq(int(os.environ.get("READS", 50_000_000)), 150, 20260925, 0, 0, 0)
n = text.size
d = text.dev()
table = ops._empty(lib.bnpk_fastq_table_words(n), np.int64)
totals = (C.c_int64 * 4)()
assert lib.bnpk_fastq_census(ops.ctx, ptr(d), n, 4, 1, ptr(table), totals, ops._s()) == 0
tiles = lib.bnpk_fastq_tiles(n)
if os.environ.get("TWO_PASS_TOO"):
    p2 = ops._empty(int(totals[2]) // 32 + 2, np.int64); e2 = ops._empty(int(totals[2]) // 64 + 2, np.int64); r2 = ops._empty(3, np.int64)
    assert lib.bnpk_fastq_encode(ops.ctx, ptr(d), n, 4, 1, ord("@"), 1, ptr(table), int(totals[1]), int(totals[2]), ptr(p2), ptr(e2), ptr(r2), ops._s()) == 0
    del p2, e2
work = ops._empty(lib.bnpk_fastq_onepass_words(n), np.int64)
work[16:16 + tiles * 5] = table[8:8 + tiles * 5]
packed = ops._empty(n // 32 + 2, np.int64); ends = ops._empty(n // 64 + 2, np.int64); err = ops._empty(3, np.int64)


def f():
    return lib.bnpk_fastq_decode_onepass(ops.ctx, ptr(d), n, 4, 1, ord("@"), 1, ptr(work), ptr(packed), ptr(ends), ptr(err), totals, ops._s())


f(); torch.cuda.synchronize()
dev.prof_enable(True); dev.prof_reset()
for _ in range(2):
    f()
torch.cuda.synchronize()
print("abl", os.environ.get("BNPK_FQ1_ABL"), {k: round(v["total_ms"] / 2, 2) for k, v in dev.prof_report().items()}, list(totals), err.cpu().numpy())
