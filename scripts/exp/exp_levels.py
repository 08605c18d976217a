"""Experiment: cost of one MSD level (hist + scatter) over 6e9 random keys as a function of the digit width."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bionumpy_amd.device import Device
from bionumpy_amd.ops import get_ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6_000_000_000
ops = get_ops(); dev = Device.get()
g = torch.Generator(device="cuda"); g.manual_seed(1)
keys = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device="cuda", generator=g)
out = torch.empty_like(keys)
for bits in (8, 9, 10, 11):
    dev.prof_enable(True); dev.prof_reset()
    for _ in range(2):
        o, child = ops.radix_partition(keys, None, 1, 62 - bits, bits, out)
    torch.cuda.synchronize()
    rep = dev.prof_report(); dev.prof_enable(False)
    print("level-1 style, %2d bits: " % bits + "  ".join("%s %.2f ms" % (k, v["total_ms"] / 2) for k, v in rep.items()), flush=True)
# second-level style: many segments
a, off1 = ops.radix_partition(keys, None, 1, 52, 10)
for bits in (9, 10, 11):
    dev.prof_enable(True); dev.prof_reset()
    for _ in range(2):
        o, child = ops.radix_partition(a, off1, 1 << 10, 52 - bits, bits, out)
    torch.cuda.synchronize()
    rep = dev.prof_report(); dev.prof_enable(False)
    print("level-2 style (1024 segments), %2d bits: " % bits + "  ".join("%s %.2f ms" % (k, v["total_ms"] / 2) for k, v in rep.items()), flush=True)
# a 9-bit first level followed by an 11-bit second level (the other way to reach 20 bits)
del a, off1
a, off1 = ops.radix_partition(keys, None, 1, 53, 9)
for bits in (10, 11):
    dev.prof_enable(True); dev.prof_reset()
    for _ in range(2):
        o, child = ops.radix_partition(a, off1, 1 << 9, 53 - bits, bits, out)
    torch.cuda.synchronize()
    rep = dev.prof_report(); dev.prof_enable(False)
    print("level-2 style (512 segments), %2d bits: " % bits + "  ".join("%s %.2f ms" % (k, v["total_ms"] / 2) for k, v in rep.items()), flush=True)
