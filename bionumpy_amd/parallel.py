"""Multi-GPU: chunks shard embarrassingly (records are independent, k-mers never span records); the
only exchange is the final merge of the per-GPU k-mer histograms.

One process per GPU, ``torch.distributed`` (backend "nccl" == RCCL over xGMI on ROCm, "gloo" in the CPU
tests).

* dense histograms (k <= 13): ``all_reduce(sum)`` of the 4^k int64 bins.
* sparse histograms (k up to 31): the 62-bit key space is cut into ``world`` contiguous ranges; every
  rank stably partitions its raw hashes by the top bits of the key (one radix pass), the partitions
  are exchanged with one uneven all-to-all (each of the 7 xGMI links of a GPU carries 1/8 of its keys
  concurrently), and every rank then sorts + run-length-encodes only the range it owns.  The result
  stays distributed: rank r holds the sorted distinct keys of range r and their global counts.
"""
import numpy as np

from .device import HArray
from .ops import get_ops

FINE_BITS = 8          # partition granularity: 256 fine buckets, contiguous groups of them per rank


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


def allreduce_dense(hist, group=None):
    """sum of dense histograms over all ranks (EncodedCounts.__add__ across GPUs)"""
    ops = get_ops()
    dist = _dist()
    t = _as_tensor(hist, ops)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return _from_tensor(t, ops)


def rank_of_bucket(world):
    """fine bucket -> owning rank (contiguous, balanced)"""
    return (np.arange(1 << FINE_BITS, dtype=np.int64) * world) >> FINE_BITS


def exchange_by_key_range(hashes, key_bits, group=None, cuts=None):
    """all-to-all of raw k-mer hashes so that every rank ends up with exactly the keys of its own range.

    ``hashes`` (HArray int64, consumed) -> (HArray int64 of the received keys, unsorted within the range,
    (lo, hi) key range this rank owns).
    cuts: bucket boundaries if the hashes are already grouped by their top FINE_BITS bits
    (bnpk_kmers_partition); otherwise one radix level partitions them here."""
    ops = get_ops()
    dist = _dist()
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    owner = rank_of_bucket(world)
    mine = np.flatnonzero(owner == rank)
    key_range = (int(mine[0]) << (key_bits - FINE_BITS), (int(mine[-1]) + 1) << (key_bits - FINE_BITS))
    if world == 1:
        return hashes, key_range
    import torch
    if cuts is not None:
        part, cuts = hashes, np.asarray(cuts.host(), dtype=np.int64)
    else:
        part, cuts = ops.partition_by_top_bits(hashes, key_bits, FINE_BITS)
    bucket_sizes = np.diff(cuts)
    send_counts = np.bincount(owner, weights=bucket_sizes, minlength=world).astype(np.int64)
    send_t = _as_tensor(part, ops)
    counts_in = torch.tensor(send_counts, dtype=torch.int64, device=send_t.device)
    counts_out = torch.empty_like(counts_in)
    dist.all_to_all_single(counts_out, counts_in, group=group)
    recv_counts = counts_out.cpu().numpy()
    recv_t = torch.empty(int(recv_counts.sum()), dtype=torch.int64, device=send_t.device)
    dist.all_to_all_single(recv_t, send_t, output_split_sizes=recv_counts.tolist(),
                           input_split_sizes=send_counts.tolist(), group=group)
    return _from_tensor(recv_t, ops), key_range


def count_sparse_virtual(shards, key_bits):
    """The N > 1 sparse path on ONE GPU: ``shards`` = [(hashes partitioned by their top FINE_BITS bits, cuts)] of N
    virtual ranks (what kmers_partitioned(FINE_BITS) leaves on every rank before the exchange).  The all-to-all is
    replaced by what it delivers — for destination r the slices of its fine buckets from source 0, 1, .. N-1, one
    after the other — and every destination then counts its own key range exactly as count_sparse_distributed does.
    Returns [(keys, counts)] per virtual rank (concatenated: the histogram of all shards) and the per-rank key counts."""
    ops = get_ops()
    world = len(shards)
    owner = rank_of_bucket(world)
    cuts = [np.asarray(c.host(), dtype=np.int64) for _, c in shards]
    out, received = [], []
    for r in range(world):
        mine = np.flatnonzero(owner == r)
        lo_b, hi_b = int(mine[0]), int(mine[-1]) + 1
        key_range = (lo_b << (key_bits - FINE_BITS), hi_b << (key_bits - FINE_BITS))
        pieces = [HArray(dev=part.dev()[c[lo_b]:c[hi_b]]) if not getattr(ops, "host_only", False)
                  else HArray(host=part.host()[c[lo_b]:c[hi_b]]) for (part, _), c in zip(shards, cuts)]
        recv = ops.concat(pieces)                       # == the receive buffer of all_to_all_single on rank r
        received.append(recv.size)
        out.append(ops.count_sparse(recv, key_bits=key_bits, consume=True, key_range=key_range))
    return out, received


def count_sparse_distributed(hashes, key_bits, group=None, cuts=None):
    """global sparse histogram, range-partitioned over the ranks: (keys, counts) of this rank's key range"""
    ops = get_ops()
    if isinstance(hashes, list):              # [HArray]: the caller gave its only reference away
        held = hashes
        hashes = held.pop()
    mine, key_range = exchange_by_key_range(hashes, key_bits, group, cuts)
    del hashes
    return ops.count_sparse(mine, key_bits=key_bits, consume=True, key_range=key_range)
