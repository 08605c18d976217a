"""-m gpu: the four finishing kernels of the sparse histogram (include/bnpk.h "finish_mode") give np.unique's answer
(oracle.count_sparse; reference semantics: bionumpy/sequence/count_encoded.py:150-188 extended to k > 8)
on keys of every shape — duplicate-free, every key ~60 times (S-genome), read errors on top of that, heavy hitters,
keys that differ only in their low bits (one home slot of the duplicate-aware kernel's table), more distinct keys than
that table holds — and the finishing call itself (bnpk_finish_sorted) does on hand-made buckets."""
import ctypes as C

import numpy as np
import pytest

import oracle
from bionumpy_amd.device import HArray

pytestmark = pytest.mark.gpu

M62 = (1 << 62) - 1
MODES = (1, 2, 3, 4, 5, 6, 0)   # general / fast + redo / workgroup-per-bucket duplicate-aware / whole cascade / fast with multiplicities / bitonic / chosen per call


@pytest.fixture(scope="module")
def env():
    import torch
    from bionumpy_amd import ops as ops_mod
    from bionumpy_amd._native import lib
    from bionumpy_amd.device import Device, ptr
    ops_mod.set_ops(None)
    yield ops_mod.get_ops(), lib, Device.get(), ptr, torch
    lib.bnpk_set_option(Device.get().ctx, b"finish_mode", 0)


def _mix(x):
    """a bijection of uint64 (splitmix64 finaliser) cut to 62 bits: distinct ids -> (almost surely) distinct keys"""
    x = x.astype(np.uint64)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xbf58476d1ce4e5b9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94d049bb133111eb)
    x = x ^ (x >> np.uint64(31))
    return (x & np.uint64(M62)).astype(np.int64)


def _genome_like(rng, n, coverage):
    return _mix(rng.integers(0, max(n // coverage, 1), size=n))


def _cases():
    rng = np.random.default_rng(20260926)
    n = 2_000_000
    yield "every key ~60 times", _genome_like(rng, n, 60)
    yield "every key ~6 times", _genome_like(rng, n, 6)
    yield "every key ~2 times", _genome_like(rng, n, 2)
    yield "distinct", rng.integers(0, 1 << 62, size=n, dtype=np.int64)
    # nearly distinct, but every bucket holds a few repeats (random 21-mers, reads at ~1x coverage): the multiplicity-counting
    # fast kernel's regime — pairs, triples and a few longer runs on top of distinct keys
    d = rng.integers(0, 1 << 62, size=n, dtype=np.int64)
    yield "a few repeats per bucket", rng.permutation(np.concatenate([d, d[:4000], d[:1500], d[:300], d[:300], np.repeat(d[5000:5040], 20)]))
    yield "every key 1-3 times", rng.permutation(np.concatenate([d[:n // 2], d[:n // 4], d[:n // 8]]))
    yield "runs of 2..70 copies", rng.permutation(np.repeat(d[:30000], rng.integers(2, 71, size=30000)))
    yield "read errors", np.concatenate([_genome_like(rng, 3 * n // 4, 60), rng.integers(0, 1 << 62, size=n // 4, dtype=np.int64)])
    yield "low bits", (rng.integers(0, 1 << 12, size=n, dtype=np.int64) << 50) | rng.integers(0, 3000, size=n, dtype=np.int64)
    yield "clusters", (rng.integers(0, 1 << 12, size=n, dtype=np.int64) << 50) | (rng.integers(0, 8, size=n, dtype=np.int64) << 44) \
        | rng.integers(0, 200, size=n, dtype=np.int64)
    yield "heavy hitters", np.concatenate([_genome_like(rng, n // 2, 60), np.full(300_000, 12345678901234567, dtype=np.int64),
                                           np.full(20_000, M62, dtype=np.int64), np.zeros(9000, dtype=np.int64)])
    yield "few values", rng.integers(0, 50, size=5000, dtype=np.int64)
    yield "one key", np.full(1, 7, dtype=np.int64)


@pytest.mark.parametrize("name,keys", list(_cases()), ids=[c[0] for c in _cases()])
def test_every_finishing_kernel_counts_like_np_unique(env, name, keys):
    ops, lib, dev, ptr, torch = env
    ek, ec = oracle.count_sparse(keys)
    taken = set()
    try:
        # claim: the last partition level without its histogram pass, the finishing kernels reading buckets of fixed stride
        # (round 5; keys without a place in their bucket — the heavy hitters — counted apart and merged in) / the plain level
        for claim in (True, False):
            ops.claim_last_level = claim
            for mode in MODES:
                assert lib.bnpk_set_option(dev.ctx, b"finish_mode", mode) == 0
                h = HArray(host=keys.copy())
                ops.last_claimed = None
                gk, gc = ops.count_sparse(h, key_bits=62)
                assert np.array_equal(gk.host(), ek) and np.array_equal(gc.host(), ec), (name, mode, claim)
                # bnpk_finish_sorted uses the partitioned keys as workspace — never the caller's array
                assert np.array_equal(h.dev().cpu().numpy(), keys), (name, mode)
                if ops.last_claimed is not None:
                    assert claim
                    taken.add("bag" if ops.last_claimed["bag"] else "claimed")
    finally:
        ops.claim_last_level = True
    if keys.size >= (1 << 20):
        assert taken, "the claiming level never ran on %s" % name


def _direct(env, keys, top_bits):
    """bnpk_finish_sorted on hand-made buckets: the keys grouped (not sorted) by their top `top_bits` bits"""
    ops, lib, dev, ptr, torch = env
    nb = 1 << top_bits
    ids = keys >> (62 - top_bits) if top_bits else np.zeros_like(keys)
    part = keys[np.argsort(ids, kind="stable")]
    off = np.zeros(nb + 1, dtype=np.int64)
    off[1:] = np.cumsum(np.bincount(ids, minlength=nb))
    ek, ec = oracle.count_sparse(keys)
    handed_back = {}
    for mode in MODES:
        assert lib.bnpk_set_option(dev.ctx, b"finish_mode", mode) == 0
        work, d_off = torch.from_numpy(part.copy()).cuda(), torch.from_numpy(off).cuda()
        out_k, out_c = torch.empty_like(work), torch.empty_like(work)
        state = torch.empty(lib.bnpk_finish_state_words(nb), dtype=torch.int64, device="cuda")
        nu, ov = C.c_int64(0), C.c_int(0)
        assert lib.bnpk_finish_sorted(dev.ctx, ptr(work), keys.size, ptr(d_off), nb, 62 - top_bits, ptr(out_k), ptr(out_c), ptr(state),
                                      None, 0, None, None, C.byref(nu), C.byref(ov), dev.stream()) == 0
        assert ov.value == 0 and nu.value == ek.size, mode
        assert np.array_equal(out_k[:nu.value].cpu().numpy(), ek) and np.array_equal(out_c[:nu.value].cpu().numpy(), ec), mode
        handed_back[mode] = int(state[3].item())
        handed_back[(mode, "wave")] = int(state[6].item())          # buckets the wavefront kernel left to the workgroup kernel
        # the same buckets at a fixed stride (bnpk_finish_sorted_strided: what the claiming level leaves behind)
        sizes = np.diff(off)
        stride = int(lib.bnpk_claimed_stride())
        if sizes.max() <= stride:
            spread = np.full(nb * stride, -1, dtype=np.int64)
            for b in np.flatnonzero(sizes):
                spread[b * stride:b * stride + sizes[b]] = part[off[b]:off[b + 1]]
            work, out_k, out_c = torch.from_numpy(spread).cuda(), torch.empty(keys.size, dtype=torch.int64, device="cuda"), \
                torch.empty(keys.size, dtype=torch.int64, device="cuda")
            nu, ov = C.c_int64(0), C.c_int(0)
            assert lib.bnpk_finish_sorted_strided(dev.ctx, ptr(work), keys.size, stride, ptr(d_off), nb, 62 - top_bits, ptr(out_k), ptr(out_c),
                                                  ptr(state), None, 0, None, None, C.byref(nu), C.byref(ov), dev.stream()) == 0
            assert ov.value == 0 and nu.value == ek.size, ("strided", mode)
            assert np.array_equal(out_k[:nu.value].cpu().numpy(), ek) and np.array_equal(out_c[:nu.value].cpu().numpy(), ec), ("strided", mode)
            assert handed_back[mode] == int(state[3].item()) and handed_back[(mode, "wave")] == int(state[6].item())
    return handed_back


def test_buckets_the_duplicate_aware_kernel_hands_back(env):
    """its table has 6144 slots and gives up after 48 probes: such buckets reach the general kernel through the redo list
    (state word 3) and come out right all the same"""
    rng = np.random.default_rng(5)
    r = lambda hi, m: rng.integers(0, hi, size=m, dtype=np.int64)
    assert _direct(env, (5 << 40) | r(4000, 8000), 0)[3] == 1               # ~3400 distinct keys on one home slot
    assert _direct(env, r(1 << 62, 8192), 0)[3] == 1                          # more distinct keys than slots
    assert _direct(env, M62 - r(100, 8000), 0)[3] == 1                        # a long cluster at the last home slot
    assert _direct(env, M62 - r(40, 8000), 0)[3] == 0                         # ... a short one stays
    assert _direct(env, r(40, 8000), 0)[3] == 0                               # ... and one at the first
    assert _direct(env, np.concatenate([np.full(5000, 99, dtype=np.int64), np.full(3000, 98, dtype=np.int64)]), 0)[3] == 0
    # 64 buckets, one of them hopeless (4000 keys differing in their low bits), some empty
    mixed = np.concatenate([(r(60, 200_000) << 56) | (r(60, 200_000) << 49), (7 << 56) | (1 << 40) | r(5000, 4000)])
    assert _direct(env, mixed, 6)[3] == 1
    assert _direct(env, (r(3, 9000) << 60) | r(1 << 20, 9000), 4)[3] >= 0
    # the wavefront kernel of the cascade (704 slots, 32 probes, at most 448 entries) passes on what it cannot hold
    got = _direct(env, (r(60, 200_000) << 56) | (r(60, 200_000) << 49), 6)
    assert got[(4, "wave")] == 0 and got[4] == 0
    got = _direct(env, np.concatenate([(r(60, 200_000) << 56) | (r(60, 200_000) << 49), (7 << 56) | (r(1500, 4000) << 45)]), 6)
    assert got[(4, "wave")] == 1 and got[4] == 0                              # ~1400 spread keys: the workgroup kernel's
    got = _direct(env, mixed, 6)
    assert got[(4, "wave")] == 1 and got[4] == 1                              # ... on one home slot: the general kernel's


def test_clusters_of_the_duplicate_aware_table_are_ranked(env):
    """neighbouring home slots: entries of one cluster are placed by comparing them with their neighbours — runs of
    adjacent homes of every length up to the table, keys arriving in random order"""
    rng = np.random.default_rng(6)
    # the home slot of a key is its top 16 free bits scaled to the table: 65536 / 6080 ~ 10.8 values per slot, so tops 10
    # apart fall on adjacent slots: one cluster.  A key that arrives late walks to the end of it, so runs up to the probe
    # limit (48) stay with the kernel and longer ones are handed back — right either way (_direct checks every mode)
    for run, stays in ((2, True), (3, True), (17, True), (40, True), (64, False), (65, False), (300, False), (2500, False)):
        tops = (np.arange(run, dtype=np.int64) * 10 + 7000) << 46
        keys = rng.permutation(np.concatenate([tops | (np.arange(run, dtype=np.int64) * 2654435761 & 0xFFFFF)] * 3))
        got = _direct(env, keys, 0)
        assert got[3] == 0 or not stays, run


@pytest.mark.parametrize("mode", [5, 0])
def test_repeats_in_every_bucket_at_scale(env, mode):
    """50 M keys, 8192 buckets (sixteen per workgroup of the multiplicity-counting kernel): the prefix of the distinct counts
    is looked up across thousands of status words, the parking ring goes round many times, tickets run far ahead of the
    blockIdx — what the 2 M-key cases above (one bucket per workgroup) cannot show.  0.1 % repeats, a few per bucket;
    whole-histogram equality with np.unique."""
    ops, lib, dev, ptr, torch = env
    rng = np.random.default_rng(77)
    n = 50_000_000
    base = rng.integers(0, 1 << 62, size=n, dtype=np.int64)
    again = rng.integers(0, n, size=n // 1000)
    base[rng.integers(0, n, size=n // 1000)] = base[again]            # ~50 000 keys occur twice (some three times)
    ek, ec = np.unique(base, return_counts=True)
    assert 0 < n - ek.size < n // 500
    assert lib.bnpk_set_option(dev.ctx, b"finish_mode", mode) == 0
    try:
        for claim in (True, False):
            ops.claim_last_level, ops.last_claimed = claim, None
            gk, gc = ops.count_sparse(HArray(host=base), key_bits=62)
            assert gk.size == ek.size and np.array_equal(gk.host(), ek) and np.array_equal(gc.host(), ec), claim
            # the path bnpk_count_sparse reports: the claiming level (1) or plain levels (2) — never the library sort (3), which
            # is what an overflow or a wait that gave up would have ended in
            assert ops.last_sparse_info["path"] == (1 if claim else 2), ops.last_sparse_info
            assert (ops.last_claimed is not None) == claim
    finally:
        ops.claim_last_level = True


# ---- the planners as single C calls (round 6: bnpk_count_sparse, bnpk_index_build — SURVEY §8b) -----------------------------
def _count_sparse_raw(env, keys, mode, key_bits=62):
    """bnpk_count_sparse through ctypes with nothing but device pointers: the binding INTEGRATION.md shows"""
    ops, lib, dev, ptr, torch = env
    n = keys.size
    d_keys = torch.from_numpy(keys.copy()).to(dev.tdev)
    out_k = torch.empty(max(n, 1), dtype=torch.int64, device=dev.tdev)
    out_c = torch.empty(max(n, 1), dtype=torch.int64, device=dev.tdev)
    nbytes = int(lib.bnpk_count_sparse_workspace(n, key_bits, 0, 0, 0, mode))
    work = torch.empty(nbytes, dtype=torch.uint8, device=dev.tdev)
    n_unique, info = C.c_int64(-1), (C.c_int64 * 5)()
    status = lib.bnpk_count_sparse(dev.ctx, ptr(d_keys), n, key_bits, 0, 0, None, 0, ptr(work), nbytes, ptr(out_k), ptr(out_c),
                                   C.byref(n_unique), info, dev.stream())
    return status, out_k[:max(n_unique.value, 0)].cpu().numpy(), out_c[:max(n_unique.value, 0)].cpu().numpy(), list(info)


@pytest.mark.parametrize("name,keys", list(_cases()), ids=[c[0] for c in _cases()])
def test_count_sparse_is_one_call(env, name, keys):
    """np.unique(return_counts=True) from ONE entry point and raw pointers, on every shape of keys; with the workspace of mode 2
    any input is taken; the paths it reports: claiming level / plain levels / library sort; at most a handful of round trips"""
    ops, lib, dev, ptr, torch = env
    ek, ec = oracle.count_sparse(keys)
    assert lib.bnpk_set_option(dev.ctx, b"finish_mode", 0) == 0
    for claim in (1, 0):
        assert lib.bnpk_set_option(dev.ctx, b"sparse_claim", claim) == 0
        status, gk, gc, info = _count_sparse_raw(env, keys, 2)
        assert status == 0, (name, status)
        assert np.array_equal(gk, ek) and np.array_equal(gc, ec), (name, claim, info)
        assert info[0] in (1, 2, 3) and (claim or info[0] != 1), info
        if name == "distinct":                                # (2 M keys in one level: the slabs' leftovers fill a small bag)
            assert info[0] == (1 if claim else 2) and info[2] <= (6 if claim else 2), info
    assert lib.bnpk_set_option(dev.ctx, b"sparse_claim", 1) == 0


def test_count_sparse_argument_checks_and_small_workspaces(env):
    ops, lib, dev, ptr, torch = env
    rng = np.random.default_rng(5)
    keys = rng.integers(0, 1 << 62, size=3_000_000, dtype=np.int64)
    ek, ec = np.unique(keys, return_counts=True)
    for mode in (0, 1):                                       # well-spread keys need no more than the plain / claiming workspace
        status, gk, gc, info = _count_sparse_raw(env, keys, mode)
        assert status == 0 and np.array_equal(gk, ek) and np.array_equal(gc, ec), (mode, info)
        assert info[0] in (1, 2), info                        # the claiming level or plain levels: never the library sort
    # one key 24 million times cannot be split by levels: the heavy bucket is counted on its own (a batch of three arrays of its
    # size) — with a workspace that has no room for that the call says so instead of answering wrongly; a few buckets a little
    # over the capacity fit the 256 MB every mode reserves for them
    same = np.full(24_000_000, 12345, dtype=np.int64)
    status, gk, gc, info = _count_sparse_raw(env, same, 0)
    assert status == -4                                       # BNPK_ERR_NOMEM
    status, gk, gc, info = _count_sparse_raw(env, same, 2)
    assert status == 0 and gk.tolist() == [12345] and gc.tolist() == [24_000_000] and info[4] == 1, info
    fewer = np.full(3_000_000, 12345, dtype=np.int64)
    status, gk, gc, info = _count_sparse_raw(env, fewer, 0)
    assert status == 0 and gk.tolist() == [12345] and gc.tolist() == [3_000_000]
    # empty input, output aliasing the input
    n_unique = C.c_int64(-1)
    assert lib.bnpk_count_sparse(dev.ctx, None, 0, 62, 0, 0, None, 0, None, 0, None, None, C.byref(n_unique), None, dev.stream()) == 0
    assert n_unique.value == 0
    t = torch.zeros(16, dtype=torch.int64, device=dev.tdev)
    w = torch.empty(1 << 20, dtype=torch.uint8, device=dev.tdev)
    assert lib.bnpk_count_sparse(dev.ctx, ptr(t), 16, 62, 0, 0, None, 0, ptr(w), 1 << 20, ptr(t), ptr(t), C.byref(n_unique), None,
                                 dev.stream()) == -1


def _two_hitters(rng, n):
    low = int(rng.integers(0, 1 << 52))                                  # (the first level takes 6 bits here, 10 on large inputs)
    hot = [(13 << 56) | low, (46 << 56) | (low ^ 0x5000)]
    k = rng.integers(0, 1 << 62, size=n, dtype=np.int64)
    k[:40_000] = hot[0]
    k[40_000:70_000] = hot[1]
    k[70_000:70_040] = hot[0] ^ rng.integers(0, 1 << 16, size=40)       # neighbours: same buckets on every level that runs
    k[70_040:70_080] = hot[1] ^ rng.integers(0, 1 << 16, size=40)
    return rng.permutation(k)


def _unique_pairs(kmers, rows):
    order = np.lexsort((rows, kmers))
    k, r = kmers[order], rows[order]
    first = np.flatnonzero(np.concatenate([[True], (k[1:] != k[:-1]) | (r[1:] != r[:-1])]))
    return np.stack([k[first], r[first]]), np.diff(np.append(first, k.size))


def test_index_build_is_one_call(env):
    """KmerIndex.create_index (kmer_indexing.py:24-47) from raw pointers: sorted distinct (k-mer, row) pairs and how often each
    occurred, against np.unique over the pairs — random 62-bit k-mers, a repeat-rich 'genome' and a heavy hitter"""
    ops, lib, dev, ptr, torch = env
    rng = np.random.default_rng(17)
    # (up to 1024 rows: ONE partition of (k-mer, row) words — radix.hip pair_source; more: the distinct values of rank * n_rows + row;
    #  option "index_pairs" 0 forces the second construction on everything)
    for name, n, n_rows, make in (("random", 2_500_000, 17, lambda: rng.integers(0, 1 << 62, size=2_500_000, dtype=np.int64)),
                                  ("many rows", 1_500_000, 5000, lambda: rng.integers(0, 1 << 62, size=1_500_000, dtype=np.int64)),
                                  ("1024 rows", 900_000, 1024, lambda: _genome_like(rng, 900_000, 3)),
                                  ("repeats", 1_200_000, 300, lambda: _genome_like(rng, 1_200_000, 40)),
                                  ("hitter", 400_000, 5, lambda: np.where(rng.random(400_000) < 0.5, 777, rng.integers(0, 1 << 40, size=400_000))),
                                  ("tiny", 7, 3, lambda: np.array([5, 5, 9, 1, 5, 9, 1], dtype=np.int64)),
                                  # two repeated k-mers in different first-level buckets whose words' remaining bits lie close together,
                                  # each with a few neighbours in its over-full bucket: the words do not say which bucket they are from,
                                  # so the batch that counts over-full buckets must not sort them together (found by tests/test_fuzz.py)
                                  ("two hitters", 300_000, 3, lambda: _two_hitters(rng, 300_000)),
                                  # more over-full buckets of equal words than are counted in a batch: the whole-key construction takes over
                                  ("many hitters", 1100 * 8300, 1, lambda: rng.permutation(np.repeat(rng.integers(0, 1 << 62, size=1100, dtype=np.int64), 8300)))):
        kmers = make().astype(np.int64)
        rows = np.sort(rng.integers(0, n_rows, size=n)).astype(np.int64)
        pairs, mult = _unique_pairs(kmers, rows)
        d_k, d_r = torch.from_numpy(kmers).to(dev.tdev), torch.from_numpy(rows).to(dev.tdev)
        nbytes = int(lib.bnpk_index_build_workspace(n, 62, n_rows))
        work = torch.empty(nbytes, dtype=torch.uint8, device=dev.tdev)
        ok, orow, oc = (torch.empty(n, dtype=torch.int64, device=dev.tdev) for _ in range(3))
        for direct in (1, 0):
            assert lib.bnpk_set_option(dev.ctx, b"index_pairs", direct) == 0
            m = C.c_int64(-1)
            assert lib.bnpk_index_build(dev.ctx, ptr(d_k), ptr(d_r), n, 62, n_rows, ptr(work), nbytes, ptr(ok), ptr(orow), ptr(oc),
                                        C.byref(m), dev.stream()) == 0, (name, direct)
            assert m.value == pairs.shape[1], (name, direct, m.value, pairs.shape)
            assert np.array_equal(ok[:m.value].cpu().numpy(), pairs[0]) and np.array_equal(orow[:m.value].cpu().numpy(), pairs[1]), (name, direct)
            assert np.array_equal(oc[:m.value].cpu().numpy(), mult), (name, direct)
        assert lib.bnpk_set_option(dev.ctx, b"index_pairs", 1) == 0
        assert np.array_equal(d_k.cpu().numpy(), kmers) and np.array_equal(d_r.cpu().numpy(), rows)      # the inputs are left alone


@pytest.mark.parametrize("copies,path", [(2, "finish.multi"), (8, "finish.dup"), (64, "finish.cascade")])
def test_mode_0_takes_the_path_the_coverage_scan_found(env, copies, path):
    """40 M keys, every value `copies` times, in 8192 buckets of 4.9 K keys: 2440 / 610 / 76 distinct keys per bucket.  The rules of
    bnpk_finish_sorted's mode 0 (csrc/finish.hip; DESIGN 4c has the scan they come from): over 1450 distinct keys per bucket (counted on
    256 buckets sorted whole) the multiplicity kernel, under it the workgroup table, and the cascade with the wavefront table first only
    while a bucket holds at most 150 — read from the library's timers; and np.unique's answer whichever it takes"""
    ops, lib, dev, ptr, torch = env
    from bionumpy_amd.device import HArray
    rng = np.random.default_rng(copies)
    n = 40_000_000
    values = rng.integers(0, 1 << 62, size=n // copies, dtype=np.int64)
    keys = rng.permutation(np.repeat(values, copies))
    assert lib.bnpk_set_option(dev.ctx, b"finish_mode", 0) == 0
    dev.prof_enable(True)
    dev.prof_reset()
    try:
        gk, gc = ops.count_sparse(HArray(host=keys), key_bits=62)
        torch.cuda.synchronize()
        report = dev.prof_report()
    finally:
        dev.prof_enable(False)
    taken = sorted(k for k in report if k.startswith("finish.") and k not in ("finish.probe", "finish.fast"))
    assert path in taken and not (set(taken) & ({"finish.multi", "finish.dup", "finish.cascade"} - {path})), (copies, taken)
    ek, ec = np.unique(values, return_counts=True)
    assert np.array_equal(gk.host(), ek) and np.array_equal(gc.host(), ec * copies)
