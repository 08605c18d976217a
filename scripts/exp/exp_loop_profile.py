"""cProfile of the reference's loop at 5 MB chunks (where the fixed cost per chunk goes)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bionumpy_amd as bnp
from bionumpy_amd.ops import get_ops
path = "/tmp/bnpk_loop_profile.fq"
ops = get_ops()
ops.synth_fastq(2_000_000, 150, 7, 1, 50_000_000).host().tofile(path)
def user(seq):
    s = bnp.as_encoded_array(seq, bnp.DNAEncoding)
    return bnp.count_encoded(bnp.get_kmers(s, k=31), axis=None)
def run():
    total = sum(user(c.sequence) for c in bnp.open(path).read_chunks())
    n = len(total); torch.cuda.synchronize(); return n
run(); run()
t0 = time.perf_counter(); run(); print("%.1f ms" % ((time.perf_counter() - t0) * 1e3))
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(45)
os.remove(path)
