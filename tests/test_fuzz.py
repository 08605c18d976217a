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


@pytest.fixture(scope="module")
def ops():
    from bionumpy_amd import ops as ops_mod
    ops_mod.set_ops(None)
    return ops_mod.get_ops()


def _rounds(ops, seed, make_cases, least=2):
    import test_gpu_parity as T
    rng = np.random.default_rng(seed)
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

    t0, seed, n = time.time(), 5_000_000, 0
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
