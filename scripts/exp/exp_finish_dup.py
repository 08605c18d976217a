"""Experiment: the duplicate-aware finishing kernel (finish_mode 3) against the general one (1) and the automatic choice (0):
parity with torch.unique on key sets of different shapes, timing on S-genome-like keys (every distinct key ~60 times).

    python scripts/exp/exp_finish_dup.py [n_keys] [reps] [noparity] [modes]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ctypes as C
from bionumpy_amd.device import Device, ptr, HArray
from bionumpy_amd.ops import get_ops
from bionumpy_amd._native import lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
skip_parity = len(sys.argv) > 3 and sys.argv[3] == "noparity"
modes = [int(m) for m in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1, 3, 4, 0]
ops = get_ops(); dev = Device.get()
G_GRID = 512    # workgroups of the duplicate-aware kernel (2 per CU): cycles/bucket is per workgroup
g = torch.Generator(device="cuda"); g.manual_seed(1)
M62 = (1 << 62) - 1


def set_mode(m):
    assert lib.bnpk_set_option(dev.ctx, b"finish_mode", m) == 0


def mix(x):
    """a bijection of int64 (splitmix64 finaliser) cut to 62 bits: distinct ids -> (almost surely) distinct keys"""
    x.bitwise_xor_(x >> 30).mul_(-4658895280553007687)     # 0xbf58476d1ce4e5b9   (in place: one temporary at a time)
    x.bitwise_xor_(x >> 27).mul_(-7723592293110705685)     # 0x94d049bb133111eb
    x.bitwise_xor_(x >> 31)
    return x.bitwise_and_(M62)


def rnd(hi, m):
    return torch.randint(0, hi, (m,), dtype=torch.int64, device="cuda", generator=g)


def genome_like(m, cov):
    return mix(rnd(max(m // cov, 1), m))


cases = (
    ("dup x60", lambda: genome_like(20_000_000, 60)),
    ("dup x6", lambda: genome_like(20_000_000, 6)),
    ("dup x2", lambda: genome_like(20_000_000, 2)),
    ("distinct", lambda: rnd(1 << 62, 20_000_000)),
    ("errors", lambda: torch.cat([genome_like(15_000_000, 60), rnd(1 << 62, 5_000_000)])),
    ("low bits", lambda: (rnd(1 << 20, 20_000_000) << 42) | rnd(3000, 20_000_000)),       # every bucket: 3000 keys on one home slot
    ("clusters", lambda: (rnd(1 << 20, 20_000_000) << 42) | (rnd(8, 20_000_000) << 36) | rnd(200, 20_000_000)),
    ("hitters", lambda: torch.cat([genome_like(10_000_000, 60), torch.full((300_000,), 12345678901234567, dtype=torch.int64, device="cuda"),
                                   torch.full((20_000,), M62, dtype=torch.int64, device="cuda"), torch.zeros(9000, dtype=torch.int64, device="cuda")])),
    ("tiny", lambda: rnd(50, 5000)),
    ("one", lambda: torch.full((1,), 7, dtype=torch.int64, device="cuda")),
)
bad = 0
for name, make in cases:
    if skip_parity:
        break
    v = make()
    ek, ec = torch.unique(v, sorted=True, return_counts=True)
    for mode in (1, 3, 4, 0):
        set_mode(mode)
        k, c = ops.count_sparse(HArray(dev=v.clone()), key_bits=62, consume=True)
        ok = k.dev().numel() == ek.numel() and bool((k.dev() == ek).all()) and bool((c.dev() == ec).all())
        bad += 0 if ok else 1
        print("parity %-9s mode %d: %s (keys %d distinct %d)" % (name, mode, "OK" if ok else "MISMATCH", v.numel(), ek.numel()), flush=True)
    # the caller's array must survive a call that does not consume it
    set_mode(3)
    w = v.clone()
    ops.count_sparse(HArray(dev=w), key_bits=62)
    if not bool((w == v).all()):
        bad += 1
        print("parity %-9s: the caller's keys were overwritten" % name)
    del v, ek, ec, w


def direct(name, v, top_bits):
    """the finishing call itself on hand-made buckets: keys grouped (not sorted) by their top `top_bits` bits"""
    global bad
    nb = 1 << top_bits
    ids = v >> (62 - top_bits) if top_bits else torch.zeros_like(v)
    order = torch.sort(ids, stable=True)[1]
    part = v[order].contiguous()
    off = torch.zeros(nb + 1, dtype=torch.int64, device="cuda")
    off[1:] = torch.cumsum(torch.bincount(ids, minlength=nb), 0)
    ek, ec = torch.unique(v, sorted=True, return_counts=True)
    for mode in (1, 3, 4, 0):
        set_mode(mode)
        work = part.clone()
        out_k, out_c = torch.empty_like(part), torch.empty_like(part)
        state = torch.empty(lib.bnpk_finish_state_words(nb), dtype=torch.int64, device="cuda")
        nu, ov = C.c_int64(0), C.c_int(0)
        st = lib.bnpk_finish_sorted(dev.ctx, ptr(work), v.numel(), ptr(off), nb, 62 - top_bits, ptr(out_k), ptr(out_c), ptr(state), None, 0, None, None,
                                    C.byref(nu), C.byref(ov), dev.stream())
        ok = st == 0 and ov.value == 0 and nu.value == ek.numel() and bool((out_k[:nu.value] == ek).all()) and bool((out_c[:nu.value] == ec).all())
        bad += 0 if ok else 1
        print("direct %-12s mode %d: %s (keys %d distinct %d, handed back %d / %d)" % (name, mode, "OK" if ok else "MISMATCH", v.numel(), ek.numel(), int(state[6]), int(state[3])), flush=True)


if not skip_parity:
    direct("one home", (5 << 40) | rnd(4000, 8000), 0)                       # 3400 distinct keys on one home slot: handed back
    direct("full", rnd(1 << 62, 8192), 0)                                      # more distinct keys than the table has slots
    direct("full dup", rnd(1 << 62, 4000).repeat(2), 0)
    direct("end of table", M62 - rnd(40, 8000), 0)                             # a cluster at the last home slot
    direct("end, long", M62 - rnd(100, 8000), 0)
    direct("start", rnd(40, 8000), 0)
    direct("two keys", torch.cat([torch.full((5000,), 99, dtype=torch.int64, device="cuda"), torch.full((3000,), 98, dtype=torch.int64, device="cuda")]), 0)
    direct("mixed buckets", torch.cat([(rnd(64, 200_000) << 56) | rnd(60, 200_000), (7 << 56) | (1 << 40) | rnd(5000, 4000)]), 6)
    direct("empty buckets", (rnd(3, 9000) << 60) | rnd(1 << 20, 9000), 4)
print("parity failures:", bad, flush=True)
set_mode(0)
if n <= 0:
    sys.exit(1 if bad else 0)

# ---- timing: S-genome-like keys, two partition levels as the pipeline runs them -----------------------------------
for cov in (60, 6):
    keys = genome_like(n, cov)
    bits2 = 9 if n <= 3_200_000_000 else 10
    a, off1 = ops.radix_partition(keys, None, 1, 52, 10)
    b0, off2 = ops.radix_partition(a, off1, 1 << 10, 52 - bits2, bits2)
    del keys
    nseg = 1 << (10 + bits2)
    sizes = off2[1:] - off2[:-1]
    print("coverage %d: %d keys, %d buckets, largest %d, over capacity %d" % (cov, n, nseg, int(sizes.max()), int((sizes > 8192).sum())), flush=True)
    counts = torch.empty(n, dtype=torch.int64, device="cuda")
    state = torch.empty(lib.bnpk_finish_state_words(nseg), dtype=torch.int64, device="cuda")
    b = torch.empty_like(b0)
    ref = None
    for mode in modes:
        set_mode(mode)
        times = []
        for rep in range(reps):
            b.copy_(b0)
            nu, ov = C.c_int64(0), C.c_int(0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            st = lib.bnpk_finish_sorted(dev.ctx, ptr(b), n, ptr(off2), nseg, 52 - bits2, ptr(a), ptr(counts), ptr(state), None, 0, None, None, C.byref(nu), C.byref(ov), dev.stream())
            e1.record(); torch.cuda.synchronize()
            times.append(round(e0.elapsed_time(e1), 2))
        srt = bool((a[1:nu.value] > a[:nu.value - 1]).all()) if nu.value > 1 else True
        sig = (nu.value, int(counts[:nu.value].sum()), int(a[:nu.value].sum()), int((a[:nu.value] * counts[:nu.value]).sum()))
        if ref is None:
            ref = sig
        if int(state[14]) > 0:
            print("   slow-path lanes %d, wave-instructions with a slow lane %d of %d" % (int(state[12]), int(state[13]), int(state[14])))
        wph = state[16:24].tolist()
        if sum(wph) > 0 and mode == 4:
            print("   wave phases (share of cycles): " + " ".join("%.3f" % (x / sum(wph)) for x in wph) + "  cycles per bucket and wave %.0f" % (sum(wph) / nseg))
        ph = state[8:16].tolist()
        if sum(ph) > 0 and mode == 3:
            print("   phases (share of cycles): " + " ".join("%.3f" % (x / sum(ph)) for x in ph) + "  cycles/bucket %.0f" % (sum(ph) / nseg * G_GRID))
        print("mode %d: finish ms %s status %d n_unique %d overflow %d sorted %s sum_counts %d same_as_first %s handed back %d / %d"
              % (mode, times, st, nu.value, ov.value, srt, sig[1], sig == ref, int(state[6]), int(state[3])), flush=True)
    del a, b, b0, counts, state
set_mode(0)
