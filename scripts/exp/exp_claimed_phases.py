"""Experiment (a build of radix.hip with -DRP_PHASES=<wave> only): where the rounds of the CLAIMING level spend their cycles on
well-spread keys and on deep coverage of a small genome (a sixth of the buckets spill into the bag).
    scripts/exp/build_variant.sh phases radix.hip -DRP_PHASES=3 ; BNPK_LIB=.../libbnpk_phases.so python scripts/exp/exp_claimed_phases.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bionumpy_amd.device import Device
from bionumpy_amd.ops import get_ops
from bionumpy_amd._native import lib
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
ops = get_ops(); dev = Device.get()
names = ["layout", "meta+carry stage", "stage new+claims", "wait+keys+rank", "issue loads", "flush", "readback+barrier", "-"]
lib.bnpk_debug_radix_phases.restype = C.c_int
def phases(tag, ms=None):
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 8)()
    assert lib.bnpk_debug_radix_phases(buf) == 0
    tot = float(sum(buf)) or 1.0
    print(tag, "  ".join("%s %.1f" % (nm, c / 256 / 1e6) for nm, c in zip(names, buf) if c), " | total Mcycles/CU %.1f" % (tot / 256 / 1e6), "" if ms is None else "| %.1f ms" % ms, flush=True)
ptr = lambda t: C.c_void_p(t.data_ptr())
for name, mode, glen, k in (("uniform", 0, 0, 25), ("genome 3 Mbp", 1, 3_000_000, 25), ("genome 100 Mbp", 1, 100_000_000, 25)):
    text = ops.synth_fastq(reads, 150, 20260925, mode, glen, 0)
    packed, ends, n, n_bases = ops.fastq_encode(text, text.size, 4, 1, ord("@"), True)
    starts, n_kmers = ops.kmer_starts_from_ends(ends, n_bases, k)
    del text, ends
    h, cuts = ops.kmers_partitioned(packed, starts, n_bases, n_kmers, k, 10)
    keys, seg = h.dev(), cuts.dev()
    n_b = 1 << 20
    stride = int(lib.bnpk_claimed_stride())
    buckets = torch.empty(n_b * stride, dtype=torch.int64, device=keys.device)
    fill = torch.zeros(2 * n_b, dtype=torch.int32, device=keys.device)
    bag_cap = max(n_kmers // 8, 1 << 16)
    bag = torch.empty(bag_cap, dtype=torch.int64, device=keys.device)
    bag_fill = torch.zeros(1, dtype=torch.int64, device=keys.device)
    phases("(clear)")
    for rep in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        st = lib.bnpk_radix_partition_claimed(dev.ctx, ptr(keys), n_kmers, ptr(seg), 1 << 10, 2 * k - 20, 10, ptr(buckets), ptr(fill), ptr(bag), bag_cap, ptr(bag_fill), dev.stream())
        assert st == 0, st
        e1.record(); torch.cuda.synchronize()
        phases("%s (bag %d of %d keys):" % (name, int(bag_fill.item()), n_kmers), e0.elapsed_time(e1))
    sizes = (seg[1:] - seg[:-1]).cpu().numpy()
    print("   level-1 segments: mean %.0f  max %.0f  min %.0f" % (sizes.mean(), sizes.max(), sizes.min()), flush=True)
    out = torch.empty_like(keys)
    for rep in range(2):                                  # the same keys through the PLAIN level (histogram pass + scatter at exact places)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        o, child = ops.radix_partition(keys, seg, 1 << 10, 2 * k - 20, 10, out)
        e1.record(); torch.cuda.synchronize()
        phases("%s, plain level:" % name, e0.elapsed_time(e1))
    csz = (child[1:] - child[:-1]).cpu().numpy()
    print("   children: mean %.0f  max %.0f  over 7552: %.1f %%  empty: %.1f %%" % (csz.mean(), csz.max(), 100.0 * (csz > 7552).mean(), 100.0 * (csz == 0).mean()), flush=True)
    del buckets, fill, bag, keys, h, packed, starts, out, o, child
    torch.cuda.empty_cache()
