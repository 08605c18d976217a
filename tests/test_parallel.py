"""N > 1 path on CPU: world_size 2, gloo, 127.0.0.1 (the kernels themselves are covered by -m gpu)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


import pytest


@pytest.mark.parametrize("ranks,port", [(2, 29731), (3, 29733)])
def test_histogram_merge_over_gloo(ranks, port):
    """2 ranks, and 3 (256 fine buckets do not divide evenly; the steps of the exchange get parts of unequal size)"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % ranks,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "DIST_OK" in r.stdout


def test_bucket_ownership_is_contiguous_and_balanced():
    import numpy as np
    from bionumpy_amd import parallel
    for world in (1, 2, 3, 4, 8):
        owner = parallel.rank_of_bucket(world)
        assert owner[0] == 0 and owner[-1] == world - 1 and np.all(np.diff(owner) >= 0)
        sizes = np.bincount(owner, minlength=world)
        assert sizes.max() - sizes.min() <= 1


def test_key_groups_cut_every_range_into_consecutive_parts():
    import numpy as np
    from bionumpy_amd import parallel
    for world in (1, 2, 3, 8):
        owner = parallel.rank_of_bucket(world)
        for groups in (1, 3, 4, 7, 40):
            b = parallel.key_groups(world, groups)
            assert b.shape == (world, groups + 1) and np.all(np.diff(b, axis=1) >= 0)
            for q in range(world):
                mine = np.flatnonzero(owner == q)
                assert b[q, 0] == mine[0] and b[q, -1] == mine[-1] + 1
                sizes = np.diff(b[q])
                assert sizes.max() - sizes.min() <= 1
            assert np.all(b[1:, 0] == b[:-1, -1])                # the ranks' ranges follow each other


def test_virtual_ranks_on_the_host_logic():
    """pipeline.fastq_kmer_histogram_virtual_ranks with the oracle-backed ops: shard -> partition by send cuts -> the
    exchange's result -> per-range counts == np.unique over all reads"""
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from oracle_ops import OracleOps
    from bionumpy_amd import ops as ops_mod, synth
    from bionumpy_amd.device import HArray
    from bionumpy_amd.pipeline import fastq_kmer_histogram_virtual_ranks
    ops_mod.set_ops(OracleOps())
    try:
        world, per, read_len, k = 3, 900, 90, 31
        texts = [HArray(host=synth.fastq_bytes(per, read_len, 9, 1, 7000, r * per)) for r in range(world)]
        codes = synth.read_codes(world * per, read_len, 9, 1, 7000, 0)
        h, _ = oracle.get_kmers(codes.reshape(-1), np.full(world * per, read_len, dtype=np.int64), k)
        ek, ec = oracle.count_sparse(h)
        for plan, groups in (("keys", None), ("counts", None), ("keys", 1), ("keys", 5), ("auto", None)):
            hists, stats, received, chosen = fastq_kmer_histogram_virtual_ranks(texts, k, plan=plan, with_plan=True, groups=groups)
            assert np.array_equal(np.concatenate([a.host() for a, _ in hists]), ek), plan
            assert np.array_equal(np.concatenate([c.host() for _, c in hists]), ec), plan
            if chosen == "keys":
                assert sum(received) == h.size                 # every k-mer crosses once, 8 bytes
            else:
                assert sum(received) == 2 * sum(np.unique(h[r * per * 60:(r + 1) * per * 60]).size for r in range(world))
        # reads of a 7000-base genome 7 times over per rank: few distinct 31-mers per rank -> (key, count) runs; unique reads -> raw keys
        assert chosen == "counts"
        texts = [HArray(host=synth.fastq_bytes(per, read_len, 9, 0, 0, r * per)) for r in range(world)]
        _, _, _, chosen = fastq_kmer_histogram_virtual_ranks(texts, k, with_plan=True)
        assert chosen == "keys"
    finally:
        ops_mod.set_ops(None)
