"""-m gpu: the multi-GPU entry points of the C-ABI (include/bnpk.h: bnpk_comm_*, bnpk_allreduce_hist, bnpk_exchange_counts,
bnpk_exchange_by_key_range, bnpk_exchange_slices) on what one GPU can run — a communicator of ONE rank (RCCL is loaded, the communicator is
made, every collective goes through RCCL's grouped send/recv or all-reduce and comes back with this rank's own data) —
and the two plans of the sparse merge played by one GPU for N ranks (parallel.count_sparse_virtual) against the
unsharded histogram.  N > 1 processes: tests/test_parallel.py (gloo, CPU)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    from bionumpy_amd._native import lib
    from bionumpy_amd.device import Device, ptr
    return lib, Device.get(), ptr, torch


def test_one_rank_communicator_through_the_abi(env):
    lib, dev, ptr, torch = env
    ident = (C.c_uint8 * 128)()
    assert lib.bnpk_comm_unique_id(ident) == 0 and any(ident)
    comm = C.c_void_p()
    assert lib.bnpk_comm_init(dev.ctx, ident, 1, 0, C.byref(comm)) == 0, lib.bnpk_last_comm_error()
    try:
        world, rank = C.c_int(-1), C.c_int(-1)
        assert lib.bnpk_comm_shape(comm, C.byref(world), C.byref(rank)) == 0 and (world.value, rank.value) == (1, 0)
        hist = torch.arange(4 ** 8, dtype=torch.int64, device="cuda") * 3 - 7
        expect = hist.clone()
        assert lib.bnpk_allreduce_hist(dev.ctx, comm, ptr(hist), hist.numel(), dev.stream()) == 0
        torch.cuda.synchronize()
        assert torch.equal(hist, expect)                                  # the sum over one rank
        send_counts = np.array([123_457], dtype=np.int64)
        recv_counts = np.zeros(1, dtype=np.int64)
        assert lib.bnpk_exchange_counts(dev.ctx, comm, send_counts.ctypes.data_as(C.c_void_p), 1,
                                        recv_counts.ctypes.data_as(C.c_void_p), dev.stream()) == 0
        assert recv_counts.tolist() == [123_457]
        per_bucket = np.arange(32, dtype=np.int64) * 11                   # (per fine bucket of the peer's range)
        got = np.zeros(32, dtype=np.int64)
        assert lib.bnpk_exchange_counts(dev.ctx, comm, per_bucket.ctypes.data_as(C.c_void_p), 32, got.ctypes.data_as(C.c_void_p),
                                        dev.stream()) == 0
        assert np.array_equal(got, per_bucket)
        keys = torch.randint(0, 1 << 62, (123_457,), dtype=torch.int64, device="cuda")
        recv = torch.zeros_like(keys)
        assert lib.bnpk_exchange_by_key_range(dev.ctx, comm, ptr(keys), send_counts.ctypes.data_as(C.c_void_p), ptr(recv),
                                              recv_counts.ctypes.data_as(C.c_void_p), dev.stream()) == 0
        torch.cuda.synchronize()
        assert torch.equal(recv, keys)
        # the same step with the slice at an offset of its own (one of several steps of the grouped exchange)
        off = np.array([23_456], dtype=np.int64)
        part_counts = np.array([50_001], dtype=np.int64)
        recv.zero_()
        assert lib.bnpk_exchange_slices(dev.ctx, comm, ptr(keys), off.ctypes.data_as(C.c_void_p), part_counts.ctypes.data_as(C.c_void_p),
                                        ptr(recv), part_counts.ctypes.data_as(C.c_void_p), dev.stream()) == 0
        torch.cuda.synchronize()
        assert torch.equal(recv[:50_001], keys[23_456:23_456 + 50_001]) and not bool(recv[50_001:].any())
        # what cannot be right is refused before RCCL sees it
        bad = np.array([5], dtype=np.int64)
        assert lib.bnpk_exchange_by_key_range(dev.ctx, comm, ptr(keys), send_counts.ctypes.data_as(C.c_void_p), ptr(recv),
                                              bad.ctypes.data_as(C.c_void_p), dev.stream()) == -1
        assert lib.bnpk_allreduce_hist(dev.ctx, None, ptr(hist), 4, dev.stream()) == -1
    finally:
        assert lib.bnpk_comm_destroy(comm) == 0


@pytest.mark.parametrize("mode,genome_len,expect_plan", [(0, 0, "keys"), (1, 400_000, "counts")])
def test_both_plans_of_the_sparse_merge_on_virtual_ranks(mode, genome_len, expect_plan):
    """3 virtual ranks of 1 M reads: duplicate-free reads choose the exchange of raw hashes, reads that cover a genome 375
    times per rank the exchange of (key, count) runs; either plan, forced, gives the histogram of all reads"""
    from bionumpy_amd import ops as ops_mod
    from bionumpy_amd.pipeline import fastq_kmer_histogram, fastq_kmer_histogram_virtual_ranks
    from bionumpy_amd.device import HArray
    ops_mod.set_ops(None)
    ops = ops_mod.get_ops()
    world, per, read_len, k, seed = 3, 1_000_000, 150, 31, 31
    texts = [ops.synth_fastq(per, read_len, seed, mode, genome_len, r * per) for r in range(world)]
    whole = ops.synth_fastq(world * per, read_len, seed, mode, genome_len, 0)
    (ek, ec), st = fastq_kmer_histogram(whole, k)
    # ("keys": the exchange in one step, and cut into steps whose pieces land back to back in one pair of arrays)
    for plan, groups in (("auto", None), ("keys", None), ("counts", None), ("keys", 1), ("keys", 7)):
        hists, stats, received, chosen = fastq_kmer_histogram_virtual_ranks(texts, k, plan=plan, with_plan=True, groups=groups)
        assert chosen == (expect_plan if plan == "auto" else plan)
        keys = ops.concat([h[0] for h in hists])
        counts = ops.concat([h[1] for h in hists])
        assert keys.size == ek.size and bool((keys.dev() == ek.dev()).all()) and bool((counts.dev() == ec.dev()).all()), plan
        if chosen == "keys":
            assert sum(received) == st.n_kmers
        else:
            assert sum(received) < st.n_kmers // 4 or mode == 0          # 16 bytes per distinct key of a rank: far fewer words


def test_eight_ranks_of_low_coverage_reads_take_the_counts_plan():
    """8 virtual ranks x 2 M reads of a 100 Mbp genome: every rank sees its k-mers ~2.4 times (38 % of them distinct), the job
    sees them ~19 times.  The plans are priced (parallel.plan_costs): 16 B per locally distinct key are fewer link bytes than
    8 B per k-mer, the received runs shrink as they are merged (the probe's sketch estimates how far), and the keys plan pays
    a third partition level — the (key, count) runs win.  (Round 3 looked at the local ratio alone and sent raw hashes.)"""
    from bionumpy_amd import ops as ops_mod, parallel
    from bionumpy_amd.pipeline import fastq_kmer_histogram, fastq_kmer_histogram_virtual_ranks
    ops_mod.set_ops(None)
    ops = ops_mod.get_ops()
    world, per, read_len, k, seed, genome = 8, 2_000_000, 150, 31, 5, 100_000_000
    texts = [ops.synth_fastq(per, read_len, seed, 1, genome, r * per) for r in range(world)]
    hists, stats, received, chosen = fastq_kmer_histogram_virtual_ranks(texts, k, with_plan=True)
    probe = parallel.last["probe"]
    assert chosen == "counts", probe
    assert 0.3 < probe["distinct_local_sum"] / probe["total"] < 0.45                     # what a rank's own histogram keeps
    # the sketch's estimate of the job's distinct keys in the probed bucket: within 10 % of the truth (the genome's k-mers there)
    whole = ops.synth_fastq(world * per, read_len, seed, 1, genome, 0)
    (ek, ec), st = fastq_kmer_histogram(whole, k)
    lo, hi = 37 << (2 * k - parallel.FINE_BITS), 38 << (2 * k - parallel.FINE_BITS)
    truth = int(((ek.dev() >= lo) & (ek.dev() < hi)).sum().item())
    assert abs(probe["distinct_global_estimate"] - truth) < 0.1 * truth, (probe, truth)
    keys = ops.concat([h[0] for h in hists])
    counts = ops.concat([h[1] for h in hists])
    assert keys.size == ek.size and bool((keys.dev() == ek.dev()).all()) and bool((counts.dev() == ec.dev()).all())


def test_exchange_stream_and_counting_stream_stay_in_order(env):
    """plan "keys" in 4 steps on ONE GPU with a one-rank RCCL communicator: step j + 1 (bnpk_exchange_slices: this rank's own
    slice, a device copy on the exchange stream) runs while the keys of step j are partitioned and finished on the counting
    stream — the ordering between the two streams is what gloo and the virtual ranks cannot exercise.  30 rounds, every one
    compared with the unsharded histogram."""
    lib, dev, ptr, torch = env
    from bionumpy_amd import ops as ops_mod, parallel
    from bionumpy_amd.device import HArray
    ops_mod.set_ops(None)
    ops = ops_mod.get_ops()
    ident = (C.c_uint8 * 128)()
    assert lib.bnpk_comm_unique_id(ident) == 0
    comm = C.c_void_p()
    assert lib.bnpk_comm_init(dev.ctx, ident, 1, 0, C.byref(comm)) == 0, lib.bnpk_last_comm_error()
    coll = parallel.AbiCollectives.__new__(parallel.AbiCollectives)      # (the handshake needs torch.distributed; one rank does not)
    coll.lib, coll.dev, coll.world, coll.rank, coll.comm = lib, dev, 1, 0, comm
    token = object()
    parallel._collectives[id(token)] = coll
    try:
        k, key_bits = 31, 62
        for it in range(30):
            n_reads = 150_000 + 37_003 * (it % 5)
            text = ops.synth_fastq(n_reads, 150, 100 + it, it % 2, 2_000_000, 0)
            packed, ends, n, n_bases = ops.fastq_encode(text, text.size, 4, 1, ord("@"), True)
            starts, n_kmers = ops.kmer_starts_from_ends(ends, n_bases, k)
            part, cuts = ops.kmers_partitioned(packed, starts, n_bases, n_kmers, k, parallel.FINE_BITS)
            flat = ops.windows_from_mask(packed, starts, n_bases, n_kmers, k, k)
            ek, ec = ops.count_sparse(flat, key_bits=key_bits)
            gk, gc = parallel.count_keys_in_groups(part, cuts, key_bits, 4, token)
            torch.cuda.synchronize()
            assert gk.size == ek.size and bool((gk.dev() == ek.dev()).all()) and bool((gc.dev() == ec.dev()).all()), it
    finally:
        del parallel._collectives[id(token)]
        assert lib.bnpk_comm_destroy(comm) == 0
