"""Experiment: where the host time of the reference's loop at 5 MB chunks goes (cProfile of the example form, exp_reference_loop.py)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bionumpy_amd as bnp
from bionumpy_amd.ops import get_ops

path = "/tmp/bnpk_loop_profile.fq"
ops = get_ops()
ops.synth_fastq(8_000_000, 150, 7, 1, 50_000_000).host().tofile(path)


def user(entries):
    return bnp.count_encoded(bnp.get_kmers(bnp.as_encoded_array(entries, bnp.DNAEncoding), k=31), axis=None)


def loop():
    total = sum(user(c.sequence) for c in bnp.open(path).read_chunks(min_chunk_size=5_000_000))
    return len(total)


loop(); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); loop(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
os.remove(path)
