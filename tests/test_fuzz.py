"""-m gpu: the parity tests of tests/test_gpu_parity.py over random shapes far beyond their fixed cases, seeded and boxed in
time (about a minute for the three together) — the fuzzers of round 4 (scripts/exp/fuzz_parity.py, fuzz_parity2.py,
fuzz_decode.py), which found the miscompiled loop of ``rc_packed_kernel`` (NOTES.md), where the driver runs them.

Every round draws its shapes from one seeded generator, so a failure names the test and its arguments and can be replayed.
``BNPK_FUZZ_SECONDS`` stretches the boxes (the default is what fits the suite)."""
import os
import sys
import time

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
BOX = float(os.environ.get("BNPK_FUZZ_SECONDS", 20.0))
SIZES = [int(x) for x in os.environ.get("BNPK_FUZZ_SIZES", "1,7,3000,70000,1300000,4000000").split(",")]   # (the planners' fuzzer)
SEED = int(os.environ.get("BNPK_FUZZ_SEED", "0"))             # 0: the suite's own sequences; anything else: other ones


@pytest.fixture(scope="module")
def ops():
    from bionumpy_amd import ops as ops_mod
    ops_mod.set_ops(None)
    return ops_mod.get_ops()


def _rounds(ops, seed, make_cases, least=2):
    import test_gpu_parity as T
    rng = np.random.default_rng(seed + SEED)
    t0, n = time.time(), 0
    while n < least or time.time() - t0 < BOX:
        for f, args in make_cases(T, rng):
            try:
                f(ops, *args)
            except AssertionError as e:
                raise AssertionError("%s%r (round %d of seed %d): %s" % (f.__name__, args, n, seed, e)) from e
        n += 1
    return n


def test_byte_movers_and_window_kernels_at_random_shapes(ops):
    """gather + encode, reverse complement (the shape that found the miscompile is one of these), row reductions, joins,
    window matches — up to 150 000 rows and tens of millions of elements"""
    def cases(T, rng):
        seed = int(rng.integers(10, 1 << 30))
        rows = int(rng.choice([1, 2, 17, 300, 5000, 40000, 150000]))
        return [(T.test_gather_encode_rows_between_other_text, (seed, max(rows, 8), int(rng.integers(1, 60)), int(rng.integers(61, 400)), bool(rng.integers(0, 2)))),
                (T.test_gather_encode_and_kmers, (seed, rows, int(rng.choice([3, 40, 200, 1000])))),
                (T.test_reverse_complement_kernels, (seed, rows, int(rng.choice([1, 20, 151, 700])))),
                (T.test_row_reductions, (seed, rows, int(rng.choice([1, 7, 160, 3000])))),
                (T.test_join_lines, (seed, min(rows, 5000), int(rng.choice([0, 5, 160, 2000])))),
                (T.test_match_windows, (seed, min(rows, 60000), int(rng.choice([1, 50, 151, 600])), int(rng.choice([1, 2, 3, 7, 31, 40])))),
                (T.test_letter_histograms_per_row, (seed, min(rows, 60000), int(rng.choice([1, 40, 151, 3000]))))]
    assert _rounds(ops, 20260927, cases) >= 2


def test_scans_merges_and_counting_paths_at_random_sizes(ops):
    """newline positions and scans up to 50 M elements, merges of 8 M-key histograms, partition levels and sparse counts of
    up to 20 M keys through every path, motif scores"""
    def cases(T, rng):
        seed = int(rng.integers(10, 1 << 30))
        big = int(rng.choice([3_000_001, 9_999_999, 33_554_433]))
        na, nb = int(rng.integers(1, 8_000_000)), int(rng.integers(1, 8_000_000))
        return [(T.test_newline_scan_matches_flatnonzero, (big + seed % 17,)),
                (T.test_exclusive_scan, (big // 2 + seed % 5,)),
                (T.test_merge_add_of_sparse_histograms, (na, nb, int(rng.integers(0, min(na, nb) + 1)))),
                (T.test_radix_partition_levels, (seed, int(rng.choice([700_000, 5_000_000, 20_000_000])), int(rng.choice([62, 42, 30])),
                                                 [[10, 10], [10], [11, 9], [7, 6, 5]][int(rng.integers(0, 4))])),
                (T.test_count_sparse_radix_path, (seed, int(rng.choice([400_000, 3_000_000, 12_000_000])), int(rng.choice([62, 42, 30, 20])),
                                                  int(rng.choice([1, 2, 3, 50])))),
                (T.test_pwm_scores, (seed, int(rng.choice([300, 5000, 60000])), int(rng.choice([50, 151, 600])), int(rng.choice([1, 6, 12, 31])))),
                (T.test_letter_histograms, (int(rng.choice([4097, 1_000_003, 7_654_321])) + seed % 13,))]
    assert _rounds(ops, 20260928, cases, least=1) >= 1


def test_fastq_decode_fast_kernels_against_the_general_ones(ops):
    """differential: the fast tile kernels (census + encode) and the general ones give the same bits or the same exception on
    random line structures (line lengths from empty to several tiles, CRLF, trailing incomplete entries, 1-4 lines per entry,
    damaged bytes)"""
    from bionumpy_amd._native import lib
    from bionumpy_amd.device import Device, HArray
    from fuzz_text import random_text
    dev = Device.get()

    def run(buf, lpe, seq_line, check_plus, encoder):
        assert lib.bnpk_set_option(dev.ctx, b"fastq_encoder", encoder) == 0
        try:
            packed, ends, n_records, n_bases = ops.fastq_encode(HArray(host=buf), buf.size, lpe, seq_line, ord("@"), check_plus)
            return ("ok", n_records, n_bases, packed.host().tobytes(), ends.host().tobytes())
        except Exception as e:                                      # noqa: BLE001
            return ("error", type(e).__name__, str(e), getattr(e, "line_number", None), getattr(e, "offset", None))

    t0, seed, n = time.time(), 5_000_000 + 1000 * SEED, 0
    try:
        while n < 50 or time.time() - t0 < BOX:
            buf, lpe, seq_line, check_plus = random_text(np.random.default_rng(seed))
            a, b = run(buf, lpe, seq_line, check_plus, 1), run(buf, lpe, seq_line, check_plus, 0)
            assert a == b, "seed %d (lines per entry %d, sequence line %d, '+' check %s, %d bytes): %s vs %s" % (
                seed, lpe, seq_line, check_plus, buf.size, a[:3], b[:3])
            seed += 1
            n += 1
    finally:
        lib.bnpk_set_option(dev.ctx, b"fastq_encoder", 1)


def test_the_planners_on_random_key_distributions(ops):
    """bnpk_count_sparse (levels, claiming level + bag, over-full buckets counted by the planner itself, bitonic / cascade / general
    finishing, the sort fall-back) and bnpk_index_build (one partition of (k-mer, row) words; by ranks) against np.unique on keys
    of random size, width and shape: uniform, piled up near zero, a few values, heavy hitters over a uniform floor, keys that
    share their top bits, a genome-like set of repeats — with the claiming level and the pair partition switched on and off"""
    import ctypes as C
    import torch
    from bionumpy_amd._native import lib
    from bionumpy_amd.device import Device, HArray
    dev = Device.get()
    rng = np.random.default_rng(20260929 + SEED)

    def keys_of(n, bits, shape):
        top = 1 << bits
        if shape == 0:
            k = rng.integers(0, top, size=n, dtype=np.int64)
        elif shape == 1:                                      # density ~ x^(-3/4)
            k = np.minimum((rng.random(n) ** 4 * float(top)).astype(np.int64), top - 1)
        elif shape == 2:
            k = rng.integers(0, min(top, 40), size=n, dtype=np.int64)
        elif shape == 3:                                      # heavy hitters over a uniform floor
            k = rng.integers(0, top, size=n, dtype=np.int64)
            hot = rng.integers(0, top, size=3, dtype=np.int64)
            k[rng.random(n) < 0.3] = hot[0]
            k[rng.random(n) < 0.05] = hot[1]
        elif shape == 4:                                      # the top bits shared: everything in a 2^-12 slice of the range
            k = (rng.integers(0, top, dtype=np.int64) & ~((top >> 12) - 1 if top >> 12 else 0)) | rng.integers(0, max(top >> 12, 1), size=n, dtype=np.int64)
        elif shape == 5:                                      # every key ~20 times
            k = rng.integers(0, top, size=max(n // 20, 1), dtype=np.int64)[rng.integers(0, max(n // 20, 1), size=n)]
        else:                                                 # many hitters that differ in their TOP bits only, each with neighbours
            k = rng.integers(0, top, size=n, dtype=np.int64)  # (the index's words drop the top bits: equal words in different buckets)
            h = int(rng.choice([2, 20, 300]))
            low_bits = max(bits - 12, 1)
            low = int(rng.integers(0, 1 << low_bits))
            tops = rng.choice(1 << (bits - low_bits), size=min(h, 1 << (bits - low_bits)), replace=False).astype(np.int64)
            hot = (tops << low_bits) | low
            share = rng.integers(0, hot.size, size=n)
            pick = rng.random(n)
            k = np.where(pick < 0.5, hot[share], k)
            near = (pick >= 0.5) & (pick < 0.51)
            k = np.where(near, hot[share] ^ rng.integers(0, 1 << min(low_bits, 12), size=n), k)
        return k.astype(np.int64)

    t0, rounds = time.time(), 0
    while rounds < 3 or time.time() - t0 < BOX:
        n = int(rng.choice(SIZES))
        bits = int(rng.choice([20, 33, 50, 62]))
        shape = int(rng.integers(0, 7))
        claim, direct = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        keys = keys_of(n, bits, shape)
        tag = (n, bits, shape, claim, direct, rounds)
        assert lib.bnpk_set_option(dev.ctx, b"sparse_claim", claim) == 0 and lib.bnpk_set_option(dev.ctx, b"index_pairs", direct) == 0
        try:
            ek, ec = np.unique(keys, return_counts=True)
            gk, gc = ops.count_sparse(HArray(host=keys.copy()), key_bits=bits)
            assert np.array_equal(gk.host(), ek) and np.array_equal(gc.host(), ec), ("count_sparse", tag, ops.last_sparse_info)
            n_rows = int(rng.choice([1, 3, 17, 1000, 1024, 1025, 40_000]))
            rows = np.sort(rng.integers(0, n_rows, size=n)).astype(np.int64)
            want, mult = np.unique(np.stack([keys, rows]), axis=1, return_counts=True)
            got = ops.unique_pairs(HArray(host=keys.copy()), HArray(host=rows), key_bits=bits, n_values=n_rows, with_counts=True)
            gk, gr, gm = got[0].host(), got[1].host(), got[2].host()
            if not (np.array_equal(gk, want[0]) and np.array_equal(gr, want[1]) and np.array_equal(gm, mult)):
                m = min(gk.size, want.shape[1])
                bad = np.flatnonzero((gk[:m] != want[0][:m]) | (gr[:m] != want[1][:m]) | (gm[:m] != mult[:m]))
                detail = [(int(j), hex(int(gk[j])), int(gr[j]), int(gm[j]), hex(int(want[0][j])), int(want[1][j]), int(mult[j])) for j in bad[:3]]
                raise AssertionError(("unique_pairs", tag, n_rows, "pairs got / want", gk.size, want.shape[1], "occurrences", int(gm.sum()), n,
                                      "mismatches", int(bad.size), detail))
        except AssertionError:
            raise
        except Exception as e:                                # noqa: BLE001  (a status from the library: say which input it was)
            raise AssertionError(("the library raised", tag, locals().get("n_rows"), repr(e))) from e
        finally:
            lib.bnpk_set_option(dev.ctx, b"sparse_claim", 1)
            lib.bnpk_set_option(dev.ctx, b"index_pairs", 1)
        rounds += 1
    assert rounds >= 3
