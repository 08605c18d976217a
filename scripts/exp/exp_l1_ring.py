"""Round 6: the fused first level with one fixed line per bucket (rp_ring_kernel, option "l1_ring") against the
write-combining scatter it replaces — per-kernel ms of the whole step with either, and that both give the same histogram.
    python scripts/exp/exp_l1_ring.py [reads] [reps]        (MB_MODE=1: S-genome, MB_K=21: random 21-mers)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bionumpy_amd._native import lib
from bionumpy_amd.device import Device
from bionumpy_amd.ops import get_ops
from bionumpy_amd.pipeline import fastq_kmer_histogram

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mode = int(os.environ.get("MB_MODE", "0"))
k = int(os.environ.get("MB_K", "31"))
ops = get_ops(); dev = Device.get()
text = ops.synth_fastq(reads, 150, 20260925, mode, 100_000_000, 0)
out = {}
sums = {}
for ring in (0, 1, 0, 1):
    assert lib.bnpk_set_option(dev.ctx, b"l1_ring", ring) == 0
    h, st = fastq_kmer_histogram(text, k)
    kd, cd = h[0].dev(), h[1].dev()
    sums[ring] = (int(kd.numel()), int(kd.sum().item()), int((kd * 31 + cd).sum().item()), int(cd.sum().item()),
                  bool((kd[1:] > kd[:-1]).all().item()))
    del h, kd, cd
    torch.cuda.synchronize()
    dev.prof_enable(True); dev.prof_reset()
    for _ in range(reps):
        h, st = fastq_kmer_histogram(text, k); del h
    torch.cuda.synchronize()
    rep = dev.prof_report()
    dev.prof_enable(False)
    out.setdefault("l1_ring=%d" % ring, []).append({name: round(v["total_ms"] / reps, 3) for name, v in rep.items()})
    print("l1_ring=%d  scatter %.2f ms  sum %.2f ms" % (ring, rep["kmers_partition_scatter"]["total_ms"] / reps,
                                                       sum(v["total_ms"] for v in rep.values()) / reps), flush=True)
print("same histogram:", sums[0] == sums[1], sums[1])
print(json.dumps({"reads": reads, "mode": mode, "k": k, "same_histogram": sums[0] == sums[1], "checks": sums[1], "runs": out}))
