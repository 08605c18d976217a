"""Multi-GPU: chunks shard embarrassingly (records are independent, k-mers never span records); the
only exchange is the final merge of the per-GPU k-mer histograms.

One process per GPU.  The collectives are the C-ABI's (include/bnpk.h: bnpk_allreduce_hist, bnpk_exchange_counts,
bnpk_exchange_by_key_range — RCCL over xGMI behind ``libbnpk.so``); ``torch.distributed`` launches the ranks, hands
rank 0's communicator id to the others, and carries the collectives itself only in the CPU tests ("gloo") or when the
ABI's communicator cannot be made (reported, never silent).

* dense histograms (k <= 13): all-reduce (sum) of the 4^k int64 bins.
* sparse histograms (k up to 31): the 62-bit key space is cut into ``world`` contiguous ranges and the result stays
  distributed — rank r holds the sorted distinct keys of range r and their global counts.  Two plans, chosen from the
  data (``choose_plan``: every rank counts one of its fine buckets, (distinct, total) are summed over the ranks and both
  plans are priced with them — ``plan_costs``: link bytes at what a GPU's xGMI links move together, the counting passes at
  what they stream at, the exchange of plan "keys" hidden behind the counting as far as its steps allow):

  - ``keys``   (nearly) duplicate-free k-mers: every rank generates its hashes already grouped by their top bits
    (bnpk_kmers_partition; the bucket boundaries are the send cuts), ONE exchange moves every raw 8-byte hash to the
    rank that owns its range (each of a GPU's 7 xGMI links carries 1/8 of its keys concurrently), and every rank
    partitions + finishes only its own range.  With ``groups`` > 1 the range of every rank is cut into that many parts and
    the exchange into as many steps (bnpk_exchange_slices): step j + 1 runs on a stream of its own while the keys step j
    delivered are counted, and the sorted pieces land next to each other in one pair of arrays;
  - ``counts`` duplicate-heavy k-mers (reads that cover a genome many times: S-genome holds every 31-mer ~60 times):
    every rank counts its OWN k-mers first — the single-GPU path, unchanged — cuts its sorted (key, count) list at the
    range boundaries, and the exchange moves 16 bytes per DISTINCT key instead of 8 per k-mer (S-genome: 1.6 GB
    instead of 48 GB per GPU); the ``world`` sorted runs a rank receives are summed by a tree of merges (bnpk_merge_add).
"""
import ctypes as C
import sys

import numpy as np

from .device import HArray
from .ops import get_ops

FINE_BITS = 8          # partition granularity: 256 fine buckets, contiguous groups of them per rank
KEY_GROUPS = 4         # plan "keys": steps the exchange is cut into (the parts of a rank's key range), overlapped with the counting
PROBE_KEYS = 4 << 20              # keys of one fine bucket that a rank counts to estimate the ratio
PROBE_LOG = __import__("os").environ.get("BNPK_PLAN_LOG", "0") != "0"     # the probe and the plan it chose, on stderr


def _dist():
    import torch.distributed as dist
    return dist


def _as_tensor(h, ops):
    import torch
    if getattr(ops, "host_only", False):           # CPU tests (gloo)
        return torch.from_numpy(np.ascontiguousarray(h.host()))
    return h.dev()


def _from_tensor(t, ops):
    if getattr(ops, "host_only", False):
        return HArray(host=t.numpy())
    return HArray(dev=t)


# ---- collectives ---------------------------------------------------------------------------------------------------
class TorchCollectives:
    """the three collectives over ``torch.distributed`` (gloo in the CPU tests)"""
    name = "torch.distributed"

    def __init__(self, group=None):
        self.group = group
        dist = _dist()
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)

    def allreduce_sum(self, hist):
        ops, dist = get_ops(), _dist()
        t = _as_tensor(hist, ops)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return _from_tensor(t, ops)

    def exchange_counts(self, send_counts):
        import torch
        dist = _dist()
        send = np.ascontiguousarray(send_counts, dtype=np.int64)
        dev = "cpu" if getattr(get_ops(), "host_only", False) else "cuda"
        t_in = torch.from_numpy(send).to(dev)
        t_out = torch.empty_like(t_in)
        dist.all_to_all_single(t_out, t_in, group=self.group)
        return t_out.cpu().numpy()

    def exchange(self, send, send_counts, recv_counts):
        import torch
        ops, dist = get_ops(), _dist()
        send_t = _as_tensor(send, ops)
        recv_t = torch.empty(int(np.sum(recv_counts)), dtype=torch.int64, device=send_t.device)
        dist.all_to_all_single(recv_t, send_t, output_split_sizes=[int(c) for c in recv_counts],
                               input_split_sizes=[int(c) for c in send_counts], group=self.group)
        return _from_tensor(recv_t, ops)


    def exchange_slices(self, send, send_offsets, send_counts, recv_counts):
        """exchange() with the slice for rank p at send[send_offsets[p]:][:send_counts[p]]"""
        import torch
        ops = get_ops()
        send_t = _as_tensor(send, ops)
        packed = torch.cat([send_t[int(o):int(o) + int(c)] for o, c in zip(send_offsets, send_counts)])
        return self.exchange(_from_tensor(packed, ops), send_counts, recv_counts)


def agree_on_communicator(group, take_id, make, undo, vote_device="cpu", can_make=None):
    """the handshake of AbiCollectives, with the communicator's own calls passed in (the CPU tests pass stand-ins):
    every rank says whether it could make a communicator at all (``can_make()``: the library loads — voted on FIRST, because
    ``make`` is itself collective: a rank that cannot even enter it would leave the others waiting inside); rank 0's
    ``take_id()`` result — or None if it raised — is broadcast; every rank calls ``make(id)``; the outcomes are all-reduced
    (MIN); unless all succeeded, every rank calls ``undo()`` and raises.  All ranks return or all ranks raise."""
    import torch
    dist = _dist()
    rank = dist.get_rank(group)
    box, why = [None], ""
    if can_make is not None:
        try:
            able = 1 if can_make() else 0
        except Exception as e:                           # noqa: BLE001
            able, why = 0, "%s: %s" % (type(e).__name__, e)
        vote = torch.tensor([able], dtype=torch.int32, device=vote_device)
        dist.all_reduce(vote, op=dist.ReduceOp.MIN, group=group)
        if int(vote.item()) == 0:
            raise RuntimeError("a communicator cannot be made on %s" % (("this rank (%s)" % (why or "can_make() said no")) if not able else "another rank"))
    if rank == 0:
        try:
            box[0] = take_id()
        except Exception as e:                           # noqa: BLE001  (told to everybody below)
            why = "%s: %s" % (type(e).__name__, e)
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    if box[0] is None:
        raise RuntimeError("rank 0 could not take a communicator id%s" % ((" (%s)" % why) if why else ""))
    ok = 1
    try:
        make(box[0])
    except Exception as e:                               # noqa: BLE001  (reported below, by every rank, after the vote)
        ok, why = 0, "%s: %s" % (type(e).__name__, e)
    vote = torch.tensor([ok], dtype=torch.int32, device=vote_device)
    dist.all_reduce(vote, op=dist.ReduceOp.MIN, group=group)
    if int(vote.item()) == 0:
        undo()
        raise RuntimeError("the communicator failed on %s" % (("this rank (%s)" % why) if not ok else "another rank"))


class AbiCollectives:
    """the same over the C-ABI (RCCL inside libbnpk.so): a communicator of its own, made from an id that rank 0 takes and
    torch.distributed broadcasts"""
    name = "bnpk C-ABI (RCCL)"

    def __init__(self, group=None):
        """Collective on every path: whether the communicator can be used is decided by ALL ranks together, so that either
        every rank gets an AbiCollectives or every rank raises (and collectives() falls back everywhere).  A rank that
        failed on its own and went on with other collectives would leave its peers blocked in this handshake.
          1. rank 0 takes the id; what is broadcast is the id or None (RCCL cannot be loaded there: nobody goes on);
          2. every rank makes its communicator and runs one small all-reduce on it; the outcomes are all-reduced (MIN) over
             torch.distributed — one failure anywhere and every rank destroys what it made and raises."""
        from ._native import lib, check
        from .device import Device
        import torch
        dist = _dist()
        self.lib, self._check = lib, check
        self.dev = Device.get()
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.comm = None

        def take_id():
            ident = (C.c_uint8 * 128)()
            status = lib.bnpk_comm_unique_id(ident)
            if status != 0:
                raise RuntimeError("bnpk_comm_unique_id: status %d" % status)
            return bytes(ident)

        def make(raw_id):
            ident = (C.c_uint8 * 128).from_buffer_copy(raw_id)
            comm = C.c_void_p()
            status = lib.bnpk_comm_init(self.dev.ctx, ident, self.world, self.rank, C.byref(comm))
            if status != 0:
                raise RuntimeError("bnpk_comm_init: status %d (%s)" % (status, lib.bnpk_last_comm_error().decode()))
            self.comm = comm
            probe = torch.ones(1, dtype=torch.int64, device=self.dev.tdev)     # one small all-reduce before anything depends on it
            self._chk(lib.bnpk_allreduce_hist(self.dev.ctx, self.comm, C.c_void_p(probe.data_ptr()), 1, self.dev.stream()))
            if int(probe.item()) != self.world:
                raise RuntimeError("bnpk_allreduce_hist over %d ranks returned %d" % (self.world, int(probe.item())))

        def can_make():                                  # (RCCL loads: dlopen + dlsym — an id is only ever taken by rank 0, once)
            return lib.bnpk_comm_available() == 1

        agree_on_communicator(group, take_id, make, self.close, self.dev.tdev if dist.get_backend(group) == "nccl" else "cpu",
                              can_make=can_make)
        import atexit
        import weakref
        ref = weakref.ref(self)                          # (the hook must not keep every communicator ever made alive)
        atexit.register(lambda: ref() is not None and ref().close())

    def __del__(self):
        self.close()

    def close(self):
        """ncclCommDestroy of the communicator this object made (at exit, or when the ranks agree not to use it)"""
        comm, self.comm = self.comm, None
        if comm is not None:
            try:
                self.lib.bnpk_comm_destroy(comm)
            except Exception:                            # noqa: BLE001  (interpreter shutdown)
                pass

    def _chk(self, status):
        if status != 0:
            raise RuntimeError("bnpk collective failed (status %d): %s" % (status, self.lib.bnpk_last_comm_error().decode()))

    def allreduce_sum(self, hist):
        t = hist.dev()
        self._chk(self.lib.bnpk_allreduce_hist(self.dev.ctx, self.comm, C.c_void_p(t.data_ptr()), t.numel(), self.dev.stream()))
        return HArray(dev=t)

    def exchange_counts(self, send_counts):
        send = np.ascontiguousarray(send_counts, dtype=np.int64)
        recv = np.empty_like(send)
        self._chk(self.lib.bnpk_exchange_counts(self.dev.ctx, self.comm, send.ctypes.data_as(C.c_void_p), send.size // self.world,
                                                recv.ctypes.data_as(C.c_void_p), self.dev.stream()))
        return recv

    def exchange(self, send, send_counts, recv_counts):
        import torch
        send_t = send.dev()
        sc = np.ascontiguousarray(send_counts, dtype=np.int64)
        rc = np.ascontiguousarray(recv_counts, dtype=np.int64)
        recv_t = torch.empty(int(rc.sum()), dtype=torch.int64, device=send_t.device)
        self._chk(self.lib.bnpk_exchange_by_key_range(self.dev.ctx, self.comm, C.c_void_p(send_t.data_ptr()), sc.ctypes.data_as(C.c_void_p),
                                                      C.c_void_p(recv_t.data_ptr()), rc.ctypes.data_as(C.c_void_p), self.dev.stream()))
        return HArray(dev=recv_t)


    def exchange_slices(self, send, send_offsets, send_counts, recv_counts):
        import torch
        send_t = send.dev()
        so = np.ascontiguousarray(send_offsets, dtype=np.int64)
        sc = np.ascontiguousarray(send_counts, dtype=np.int64)
        rc = np.ascontiguousarray(recv_counts, dtype=np.int64)
        recv_t = torch.empty(int(rc.sum()), dtype=torch.int64, device=send_t.device)
        self._chk(self.lib.bnpk_exchange_slices(self.dev.ctx, self.comm, C.c_void_p(send_t.data_ptr()), so.ctypes.data_as(C.c_void_p),
                                                sc.ctypes.data_as(C.c_void_p), C.c_void_p(recv_t.data_ptr()),
                                                rc.ctypes.data_as(C.c_void_p), self.dev.stream()))
        return HArray(dev=recv_t)


_collectives = {}
last = {"plan": None, "collectives": None, "groups": None, "probe": None}     # what the last sparse merge of this process did (bench.py reports it)


def collectives(group=None):
    """the collectives of a process group (made once): the C-ABI's on the GPU, torch.distributed's in the CPU tests"""
    key = id(group)
    if key not in _collectives:
        if getattr(get_ops(), "host_only", False):
            _collectives[key] = TorchCollectives(group)
        else:
            try:
                _collectives[key] = AbiCollectives(group)
            except Exception as e:                       # said aloud; the run goes on over torch.distributed's RCCL
                sys.stderr.write("bionumpy_amd.parallel: the C-ABI communicator could not be made (%s: %s); the collectives "
                                 "run over torch.distributed\n" % (type(e).__name__, e))
                _collectives[key] = TorchCollectives(group)
    return _collectives[key]


def group_is_up():
    """a torch.distributed process group with more than one rank has been initialised in this process"""
    try:
        dist = _dist()
        return bool(dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
    except Exception:                                    # noqa: BLE001
        return False


def all_gather_objects(obj, group=None):
    """[every rank's ``obj``] in rank order (small host objects: torch.distributed.all_gather_object)"""
    dist = _dist()
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, obj, group=group)
    return out


def allgather_runs(keys, counts, group=None):
    """every rank's (keys, counts) run, one behind the other in rank order, on every rank (torch.distributed.all_gather of
    runs padded to the longest: the gather of a range-partitioned histogram — not on the counting path)"""
    import torch
    ops, dist = get_ops(), _dist()
    world = dist.get_world_size(group)
    host = getattr(ops, "host_only", False) or dist.get_backend(group) != "nccl"
    as_t = (lambda h: torch.from_numpy(np.ascontiguousarray(h.host()))) if host else (lambda h: h.dev())
    n = torch.tensor([keys.size], dtype=torch.int64, device="cpu" if host else as_t(keys).device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(x.item()) for x in sizes]
    longest = max(max(sizes), 1)
    out = []
    for h in (keys, counts):
        t = as_t(h)
        padded = torch.zeros(longest, dtype=torch.int64, device=t.device)
        padded[:t.numel()] = t
        parts = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(parts, padded, group=group)
        joined = torch.cat([p[:m] for p, m in zip(parts, sizes)])
        out.append(HArray(host=joined.numpy()) if host else HArray(dev=joined))
    return out[0], out[1]


def allreduce_dense(hist, group=None):
    """sum of dense histograms over all ranks (EncodedCounts.__add__ across GPUs)"""
    return collectives(group).allreduce_sum(hist)


# ---- key ranges ------------------------------------------------------------------------------------------------------
def rank_of_bucket(world):
    """fine bucket -> owning rank (contiguous, balanced)"""
    return (np.arange(1 << FINE_BITS, dtype=np.int64) * world) >> FINE_BITS


def key_range_of(rank, world, key_bits):
    mine = np.flatnonzero(rank_of_bucket(world) == rank)
    return int(mine[0]) << (key_bits - FINE_BITS), (int(mine[-1]) + 1) << (key_bits - FINE_BITS)


def _slice(h, a, b):
    return HArray(dev=h.dev()[a:b]) if h.on_device else HArray(host=h.host()[a:b])


SKETCH_SLOTS = 1 << 16            # linear-counting slots of the probe's sketch
SKETCH_SAMPLE = 256               # ... over the distinct keys whose mixed hash is 0 modulo this (the same keys on every rank)
_MIX = np.uint64(0x9E3779B97F4A7C15)


def probe_bucket(sizes_sum):
    """the fine bucket every rank probes: the fullest one of the job (``sizes_sum`` = the ranks' bucket sizes added up), so
    that all sketches cover the same key range and can be summed"""
    return int(np.argmax(sizes_sum))


def probe_ratio(part, cuts, key_bits, bucket):
    """(distinct, total, sketch) of fine bucket ``bucket`` of this rank's k-mers (at most PROBE_KEYS of them) — the SAME
    bucket on every rank (probe_bucket), so that what the ranks found can be put together: the sketch marks, for a
    hash-chosen 1/SKETCH_SAMPLE of the distinct keys, one of SKETCH_SLOTS slots each; summed over the ranks, the number of
    marked slots gives the number of distinct keys of the bucket in the whole job (linear counting), i.e. how far the runs
    of the "counts" plan shrink when they are merged.  A rank that holds nothing of the bucket contributes (0, 0, empty)."""
    ops = get_ops()
    sketch = np.zeros(SKETCH_SLOTS, dtype=np.int64)
    b = int(bucket)
    if cuts[b + 1] == cuts[b]:
        return 0, 0, sketch
    a, e = int(cuts[b]), int(min(cuts[b + 1], cuts[b] + PROBE_KEYS))
    sample = _slice(part, a, e)
    sample = HArray(dev=sample.dev().clone()) if sample.on_device else HArray(host=sample.host().copy())
    keys, _ = ops.count_sparse(sample, key_bits=key_bits, consume=True)
    mixed = keys.host().astype(np.uint64) * _MIX
    mixed ^= mixed >> np.uint64(29)
    chosen = mixed[(mixed & np.uint64(SKETCH_SAMPLE - 1)) == 0]
    sketch[((chosen >> np.uint64(8)) % np.uint64(SKETCH_SLOTS)).astype(np.int64)] = 1
    return keys.size, e - a, sketch


def global_distinct(sketch_sum):
    """distinct keys of the probed bucket over all ranks, from the summed sketches (linear counting: n = -m ln(empty / m))"""
    m = sketch_sum.size
    empty = int(np.count_nonzero(sketch_sum == 0))
    if empty == 0:
        return float("inf")
    return -m * np.log(empty / m) * SKETCH_SAMPLE


# what the plans are priced with (per GPU): bytes per second all xGMI links of a GPU move together in a grouped send/recv —
# ASSUMED, never measured (no multi-GPU node has run this): 7 links x 25 GB/s of payload each way, a third of the 76.8 GB/s
# per direction the wire is quoted at (MI355X_MICROARCH.md: 153.6 GB/s per link, both ways); a pessimistic link favours
# the plan that moves fewer bytes, which is the safer mistake where the link is the bound — and what the counting kernels
# stream at (the partition levels and the finishing kernels run at ~5 TB/s)
LINK_BYTES_PER_S = 175e9
HBM_BYTES_PER_S = 5.0e12


def plan_costs(distinct, total, world, distinct_global=None):
    """seconds per k-mer of one rank for either plan of the sparse merge, from the probe summed over the ranks:
    ``distinct`` of the ``total`` sampled k-mers are distinct on the rank that holds them (what the ranks' own histograms
    keep, summed), ``distinct_global`` are distinct in the whole job (default: as if the ranks shared no key).

    keys    level 1 with the send cuts (8 B written), the exchange of 8 B per k-mer to the N - 1 peers hidden behind the
            counting of what arrived (KEY_GROUPS steps: the longer of the two, plus one step that overlaps with nothing),
            and the counting itself: the levels below the send cuts (the 8-bit pre-exchange level resolves less than a full
            one, so the received keys take two levels where a GPU on its own takes one more than the fused first) and the
            finishing pass over keys that repeat as they do in the whole job;
    counts  the single-GPU count of the rank's own k-mers, 16 B per LOCALLY distinct key over the links, and the tree of
            merges over the N runs a rank receives: level j writes (and reads) 16 B per key of runs that cannot hold more than
            the job's distinct keys of their range."""
    if total <= 0:
        return {"keys": 0.0, "counts": 0.0}
    world = max(int(world), 2)
    r = distinct / total
    rg = r if distinct_global is None else min(r, distinct_global / total)     # (per k-mer of the job: total is summed over the ranks)
    f = (world - 1) / world
    level = (8 + 16) / HBM_BYTES_PER_S                     # histogram pass + scatter of one partition level, per key
    finish = lambda ratio: (8 + 16 * ratio) / HBM_BYTES_PER_S
    link = 8 * f / LINK_BYTES_PER_S
    count_received = 2 * level + finish(rg)
    keys = 8 / HBM_BYTES_PER_S + max(link, count_received) + min(link, count_received) / KEY_GROUPS
    merges, sources = 0.0, 1
    while sources < world:                                 # (per k-mer of this rank: its runs hold r keys, the job's range world * rg)
        sources *= 2
        merges += 2 * 16 * min(r, world * rg / max(world / sources, 1)) / HBM_BYTES_PER_S
    counts = 8 / HBM_BYTES_PER_S + level + finish(r) + 16 * r * f / LINK_BYTES_PER_S + merges
    return {"keys": float(keys), "counts": float(counts)}


def choose_plan(distinct, total, world=8, distinct_global=None):
    """the cheaper plan by plan_costs (world: ranks of the job)"""
    if total <= 0:
        return "keys"
    c = plan_costs(distinct, total, world, distinct_global)
    return "counts" if c["counts"] < c["keys"] else "keys"


def _merge_runs(runs):
    """the sum of sorted (keys, counts) runs: a tree of pairwise merges along the merge path (bnpk_merge_add)"""
    ops = get_ops()
    runs = [r for r in runs if r[0].size]
    if not runs:
        empty = HArray(host=np.zeros(0, dtype=np.int64))
        return empty, HArray(host=np.zeros(0, dtype=np.int64))
    while len(runs) > 1:
        merged = [ops.merge_add(a[0], a[1], b[0], b[1]) for a, b in zip(runs[0::2], runs[1::2])]
        if len(runs) % 2:
            merged.append(runs[-1])
        runs = merged
    return runs[0]


def _range_cuts(keys, world, key_bits):
    """positions in a sorted key list where the ranks' key ranges begin (world + 1 of them)"""
    ops = get_ops()
    bounds = np.array([key_range_of(r, world, key_bits)[0] for r in range(world)], dtype=np.int64)
    pos = ops.search_sorted(keys, HArray(host=bounds)).host()
    return np.concatenate([pos, [keys.size]]).astype(np.int64)


# ---- the two plans -----------------------------------------------------------------------------------------------------
def exchange_by_key_range(hashes, key_bits, group=None, cuts=None):
    """plan "keys": all-to-all of raw k-mer hashes so that every rank ends up with exactly the keys of its own range.

    ``hashes`` (HArray int64, consumed) -> (HArray int64 of the received keys, unsorted within the range,
    (lo, hi) key range this rank owns).
    cuts: bucket boundaries if the hashes are already grouped by their top FINE_BITS bits
    (bnpk_kmers_partition); otherwise one radix level partitions them here."""
    ops = get_ops()
    coll = collectives(group)
    world, rank = coll.world, coll.rank
    key_range = key_range_of(rank, world, key_bits)
    if world == 1:
        return hashes, key_range
    if cuts is not None:
        part, cuts = hashes, np.asarray(cuts.host(), dtype=np.int64)
    else:
        part, cuts = ops.partition_by_top_bits(hashes, key_bits, FINE_BITS)
    send_counts = np.bincount(rank_of_bucket(world), weights=np.diff(cuts), minlength=world).astype(np.int64)
    recv_counts = coll.exchange_counts(send_counts)
    return coll.exchange(part, send_counts, recv_counts), key_range


def key_groups(world, groups):
    """first fine bucket of part j of rank q's range: bounds[q][j], j = 0 .. groups (the ranges of rank_of_bucket cut
    into ``groups`` nearly equal runs of fine buckets; a rank with fewer buckets than groups has empty parts)"""
    owner = rank_of_bucket(world)
    bounds = np.zeros((world, groups + 1), dtype=np.int64)
    for q in range(world):
        mine = np.flatnonzero(owner == q)
        lo, n = (int(mine[0]), mine.size) if mine.size else (0, 0)
        bounds[q] = lo + (np.arange(groups + 1, dtype=np.int64) * n) // groups
    return bounds


class _SideStream:
    """the stream the exchange steps run on (None on the CPU: the steps then run where they are called)"""

    def __init__(self, ops):
        self.torch = None
        if not getattr(ops, "host_only", False):
            import torch
            self.torch = torch
            self.stream = torch.cuda.Stream()
            self.stream.wait_stream(torch.cuda.current_stream())    # what is sent was produced on the caller's stream

    def run(self, fn):
        """fn() on the side stream -> (result, event or None)"""
        if self.torch is None:
            return fn(), None
        with self.torch.cuda.stream(self.stream):
            result = fn()
            event = self.torch.cuda.Event()
            event.record(self.stream)
        return result, event

    def ready(self, result, event):
        """the caller's stream waits for the step that produced ``result``"""
        if event is not None:
            cur = self.torch.cuda.current_stream()
            cur.wait_event(event)
            result.dev().record_stream(cur)
        return result


def count_keys_in_groups(part, cuts, key_bits, groups, group=None):
    """plan "keys" with the exchange cut into ``groups`` steps and overlapped with the counting.
    part, cuts: this rank's hashes grouped by their top FINE_BITS bits and the bucket boundaries (bnpk_kmers_partition).
    Returns (keys, counts): views of one pair of arrays holding the sorted pieces back to back."""
    ops = get_ops()
    coll = collectives(group)
    world, rank = coll.world, coll.rank
    cuts = np.asarray(cuts.host() if isinstance(cuts, HArray) else cuts, dtype=np.int64)
    bounds = key_groups(world, groups)
    send_off = cuts[bounds[:, :-1]]                                  # [peer][step]
    send_cnt = cuts[bounds[:, 1:]] - send_off
    recv_cnt = np.asarray(coll.exchange_counts(send_cnt.ravel()), dtype=np.int64).reshape(world, groups)   # [source][step]
    total = int(recv_cnt.sum())
    keys_all, counts_all = ops.empty_i64(total), ops.empty_i64(total)
    side = _SideStream(ops)
    step = lambda j: side.run(lambda: coll.exchange_slices(part, send_off[:, j], send_cnt[:, j], recv_cnt[:, j]))
    pos, pending = 0, step(0)
    for j in range(groups):
        mine = side.ready(*pending)
        pending = step(j + 1) if j + 1 < groups else None            # in flight while part j is counted
        lo_b, hi_b = int(bounds[rank, j]), int(bounds[rank, j + 1])
        if hi_b > lo_b:
            key_range = (lo_b << (key_bits - FINE_BITS), hi_b << (key_bits - FINE_BITS))
            k, _ = ops.count_sparse(mine, key_bits=key_bits, consume=True, key_range=key_range, dest=(keys_all, counts_all, pos))
            pos += k.size
        del mine
    return _slice(keys_all, 0, pos), _slice(counts_all, 0, pos)


def exchange_counted(keys, counts, key_bits, group=None):
    """plan "counts": this rank's sorted (keys, counts) -> the global (keys, counts) of the key range it owns"""
    coll = collectives(group)
    world = coll.world
    if world == 1:
        return keys, counts
    pos = _range_cuts(keys, world, key_bits)
    send_counts = np.diff(pos)
    recv_counts = coll.exchange_counts(send_counts)
    rk = coll.exchange(keys, send_counts, recv_counts)
    rc = coll.exchange(counts, send_counts, recv_counts)
    off = np.concatenate([[0], np.cumsum(recv_counts)]).astype(np.int64)
    return _merge_runs([(_slice(rk, int(off[q]), int(off[q + 1])), _slice(rc, int(off[q]), int(off[q + 1]))) for q in range(world)])


def count_sparse_distributed(hashes, key_bits, group=None, cuts=None, plan="auto", groups=None):
    """global sparse histogram, range-partitioned over the ranks: (keys, counts) of this rank's key range.
    ``hashes`` grouped by their top FINE_BITS bits with bucket boundaries ``cuts`` (bnpk_kmers_partition), or raw.
    groups: steps of the exchange of plan "keys" (default KEY_GROUPS; 1 = one exchange, then one count)."""
    ops = get_ops()
    if isinstance(hashes, list):              # [HArray]: the caller gave its only reference away
        held = hashes
        hashes = held.pop()
    coll = collectives(group)
    if cuts is None:
        hashes, cuts_np = ops.partition_by_top_bits(hashes, key_bits, FINE_BITS)
        cuts = HArray(host=np.asarray(cuts_np, dtype=np.int64))
    if plan == "auto":
        cuts_np = np.asarray(cuts.host(), dtype=np.int64)
        sizes = coll.allreduce_sum(HArray(host=np.diff(cuts_np).astype(np.int64))).host()     # which bucket: agreed by all ranks
        bucket = probe_bucket(sizes)
        d, t, sketch = probe_ratio(hashes, cuts_np, key_bits, bucket)
        both = coll.allreduce_sum(HArray(host=np.concatenate([[d, t], sketch]).astype(np.int64))).host()
        plan = choose_plan(int(both[0]), int(both[1]), coll.world, global_distinct(both[2:]))
        last["probe"] = {"bucket": bucket, "distinct_local_sum": int(both[0]), "total": int(both[1]),
                         "distinct_global_estimate": float(global_distinct(both[2:]))}
        if PROBE_LOG:
            sys.stderr.write("bionumpy_amd.parallel: probe %r -> plan %s\n" % (last["probe"], plan))
    last["plan"], last["collectives"] = plan, coll.name
    if plan == "counts":
        keys, counts = ops.count_sparse(hashes, key_bits=key_bits, consume=True, partition=(cuts, FINE_BITS))
        del hashes
        return exchange_counted(keys, counts, key_bits, group)
    groups = KEY_GROUPS if groups is None else int(groups)
    last["groups"] = groups if coll.world > 1 else 1
    if groups > 1 and coll.world > 1:
        return count_keys_in_groups(hashes, cuts, key_bits, groups, group)
    mine, key_range = exchange_by_key_range(hashes, key_bits, group, cuts)
    del hashes
    return ops.count_sparse(mine, key_bits=key_bits, consume=True, key_range=key_range)


def count_sparse_virtual(shards, key_bits, plan="auto", groups=None):
    """The N > 1 sparse path on ONE GPU: ``shards`` = [(hashes partitioned by their top FINE_BITS bits, cuts)] of N
    virtual ranks (what kmers_partitioned(FINE_BITS) leaves on every rank before the exchange).  The exchange is
    replaced by what it delivers — for destination r the slices of its range from source 0, 1, .. N-1, one after the
    other — and everything else is what count_sparse_distributed does: the probe and the choice of the plan, the
    per-range counting (plan "keys") or the local histograms, their cuts and the tree of merges (plan "counts").
    Returns ([(keys, counts)] per virtual rank — concatenated: the histogram of all shards —, the words every rank
    received, the plan)."""
    ops = get_ops()
    world = len(shards)
    owner = rank_of_bucket(world)
    cuts = [np.asarray(c.host(), dtype=np.int64) for _, c in shards]
    if plan == "auto":
        d = t = 0
        sketch = np.zeros(SKETCH_SLOTS, dtype=np.int64)
        bucket = probe_bucket(sum(np.diff(c) for c in cuts))
        for (part, _), c in zip(shards, cuts):
            dr, tr, sr = probe_ratio(part, c, key_bits, bucket)
            d, t, sketch = d + dr, t + tr, sketch + sr
        plan = choose_plan(d, t, world, global_distinct(sketch))
        last["probe"] = {"bucket": bucket, "distinct_local_sum": int(d), "total": int(t),
                         "distinct_global_estimate": float(global_distinct(sketch))}
    out, received = [], []
    if plan == "counts":
        local = [ops.count_sparse(part, key_bits=key_bits, partition=(c, FINE_BITS)) for part, c in shards]   # (c: the HArray)
        pos = [_range_cuts(k, world, key_bits) for k, _ in local]
        for r in range(world):
            runs = [(_slice(k, int(p[r]), int(p[r + 1])), _slice(c, int(p[r]), int(p[r + 1]))) for (k, c), p in zip(local, pos)]
            received.append(2 * sum(run[0].size for run in runs))
            out.append(_merge_runs(runs))
        return out, received, plan
    groups = KEY_GROUPS if groups is None else int(groups)
    bounds = key_groups(world, groups)
    for r in range(world):
        # what count_keys_in_groups does on rank r: step j delivers part j of its range from source 0, 1, .. N-1
        total = sum(int(c[bounds[r, groups]] - c[bounds[r, 0]]) for c in cuts)
        keys_all, counts_all, pos = ops.empty_i64(total), ops.empty_i64(total), 0
        for j in range(groups):
            lo_b, hi_b = int(bounds[r, j]), int(bounds[r, j + 1])
            if hi_b == lo_b:
                continue
            key_range = (lo_b << (key_bits - FINE_BITS), hi_b << (key_bits - FINE_BITS))
            recv = ops.concat([_slice(part, int(c[lo_b]), int(c[hi_b])) for (part, _), c in zip(shards, cuts)])
            k, _ = ops.count_sparse(recv, key_bits=key_bits, consume=True, key_range=key_range, dest=(keys_all, counts_all, pos))
            pos += k.size
        received.append(total)
        out.append((_slice(keys_all, 0, pos), _slice(counts_all, 0, pos)))
    return out, received, plan
