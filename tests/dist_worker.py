"""world_size-2 gloo worker for tests/test_parallel.py: the multi-GPU merge path of bionumpy_amd.parallel
driven on CPU with the oracle-backed ops (host logic + collectives; the kernels are covered by -m gpu)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch.distributed as dist
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    import oracle
    from oracle_ops import OracleOps
    from bionumpy_amd import ops as ops_mod, parallel, synth
    from bionumpy_amd.device import HArray
    from bionumpy_amd.pipeline import fastq_kmer_histogram
    ops_mod.set_ops(OracleOps())

    n_reads, read_len, seed, genome = 300, 80, 7, 5000
    out = {}
    # what `bench.py --gpus N` runs on every rank (pipeline.fastq_kmer_histogram with the process group up: the probe, the plan
    # it prices, the exchange, the per-range counting or the tree of merges) on both synthetic inputs of SURVEY §8d:
    # S-uniform (mode 0: all k-mers distinct — plan "keys" territory) and S-genome (mode 1: every k-mer many times — "counts")
    for mode in (0, 1):
        text = synth.fastq_bytes(n_reads, read_len, seed, mode, genome, first_read=rank * n_reads)
        for k in (4, 15, 31):
            hist, stats = fastq_kmer_histogram(HArray(host=text), k)
            if isinstance(hist, tuple):
                out[(mode, k)] = (hist[0].host().copy(), hist[1].host().copy(), parallel.last["plan"])
            else:
                out[(mode, k)] = hist.host().copy()
    # both plans of the sparse merge, forced: raw hashes to their key range / local histograms cut at the range boundaries
    ops = ops_mod.get_ops()
    res = oracle.scan_one_line_buffer(text, oracle.FASTQ)
    codes = oracle.encode_dna(oracle.gather_rows(text, res.field_starts[:, 1], res.field_lens[:, 1]))
    mine, _ = oracle.get_kmers(codes, res.field_lens[:, 1], 31)
    # ("keys" with the exchange in one step, and cut into 3 and — the default — 4 steps that overlap with the counting)
    for plan, groups in (("keys", None), ("counts", None), ("keys", 1), ("keys", 3)):
        keys, counts = parallel.count_sparse_distributed(HArray(host=mine.copy()), 62, plan=plan, groups=groups)
        assert parallel.last["plan"] == plan and (plan == "counts" or parallel.last["groups"] == (groups or parallel.KEY_GROUPS))
        out[plan if groups is None else "keys/%d" % groups] = (keys.host().copy(), counts.host().copy())
    assert parallel.collectives().name == "torch.distributed"
    # the handshake of the C-ABI communicator (parallel.agree_on_communicator) with stand-ins for its calls: whatever fails
    # on whichever rank, EVERY rank raises (and falls back together) — none is left blocked in a collective
    made, undone = [], []
    parallel.agree_on_communicator(None, lambda: b"id", lambda raw: made.append(raw), lambda: undone.append(1))
    assert made == [b"id"] and not undone
    def fails_on(bad_rank):
        def make(raw):
            if rank == bad_rank:
                raise OSError("librccl.so: cannot open shared object file")
        return make
    def no_id():
        raise OSError("no RCCL here")
    for take, make, word in ((lambda: b"id", fails_on(world - 1), "failed on"), (lambda: b"id", fails_on(0), "failed on"),
                             (no_id, lambda raw: None, "rank 0 could not")):
        undone = []
        try:
            parallel.agree_on_communicator(None, take, make, lambda: undone.append(1))
            raise AssertionError("a communicator that failed somewhere was accepted on rank %d" % rank)
        except RuntimeError as e:
            assert word in str(e), e
    # a rank that cannot even enter the collective make (its library does not load): found out BEFORE anybody enters it
    entered = []
    try:
        parallel.agree_on_communicator(None, lambda: b"id", lambda raw: entered.append(1), lambda: None, can_make=lambda: rank != world - 1)
        raise AssertionError("accepted")
    except RuntimeError as e:
        assert "cannot be made" in str(e) and not entered, e
    dist.barrier()                                            # (every rank is still in step)
    gathered = [None] * world
    dist.all_gather_object(gathered, out)
    if rank == 0:
      plans_taken = set()
      for mode in (0, 1):
        all_text = synth.fastq_bytes(n_reads * world, read_len, seed, mode, genome, first_read=0)
        res = oracle.scan_one_line_buffer(all_text, oracle.FASTQ)
        codes = oracle.encode_dna(oracle.gather_rows(all_text, res.field_starts[:, 1], res.field_lens[:, 1]))
        for k in (4, 15, 31):
            h, _ = oracle.get_kmers(codes, res.field_lens[:, 1], k)
            if k <= 13:
                expect = oracle.count_dense(h, k)
                for r in range(world):
                    assert np.array_equal(gathered[r][(mode, k)], expect), "dense all-reduce mismatch"
            else:
                ek, ec = oracle.count_sparse(h)
                keys = np.concatenate([gathered[r][(mode, k)][0] for r in range(world)])
                counts = np.concatenate([gathered[r][(mode, k)][1] for r in range(world)])
                assert np.array_equal(keys, ek) and np.array_equal(counts, ec), "sparse key-range merge mismatch"
                bounds = [g[(mode, k)][0] for g in gathered]
                for a, b in zip(bounds[:-1], bounds[1:]):
                    assert a.size == 0 or b.size == 0 or a[-1] < b[0], "rank ranges overlap"
                assert len({g[(mode, k)][2] for g in gathered}) == 1, "the ranks chose different plans"
                plans_taken.add((mode, gathered[0][(mode, k)][2]))
                if k == 31 and mode == 1:
                    for plan in ("keys", "counts", "keys/1", "keys/3"):
                        assert np.array_equal(np.concatenate([gathered[r][plan][0] for r in range(world)]), ek), plan
                        assert np.array_equal(np.concatenate([gathered[r][plan][1] for r in range(world)]), ec), plan
      # the probe prices the plans from the data: all-distinct k-mers go as raw keys, a genome's repeated k-mers as counts
      assert (0, "keys") in plans_taken and (1, "counts") in plans_taken, plans_taken
      print("DIST_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
