"""The whole hot path over one HBM-resident FASTQ batch: decode -> 2-bit -> k-mers -> histogram.

Equivalent to, chunk by chunk in the reference (scripts/kmer_counting_example.py:4-17,
docs_source/topics/kmers.rst:42-62):

    for chunk in bnp.open(fq).read_chunks(): counts += count_kmers(change_encoding(chunk.sequence, DNA), k)

but on one batch sized for 288 GB of HBM (tens of millions of reads) instead of 5 MB chunks, and with
every intermediate released as soon as the next stage has consumed it.  For the sparse histogram the text is
decoded by the fused census + encode kernels (no newline table, field tables or row offsets) and the
k-mer hashes are never materialised in read order: ``bnpk_kmers_partition`` generates them straight
into the first level of the MSD radix partition.  bench.py times exactly this function; __graft_entry__.smoke()
and the tests check it against the oracle.
"""
from collections import namedtuple

from . import parallel
from .io.buffers import FastQBuffer
from .ops import get_ops

DENSE_MAX_K = 13            # 4^13 int64 bins = 512 MiB; above that the histogram is sparse

BatchStats = namedtuple("BatchStats", "n_reads n_bases n_kmers n_bytes")


def fastq_kmer_histogram(text, k, group=None, buffer_type=FastQBuffer, fused=True, canonical=False):
    """text: HArray uint8 holding complete FASTQ records.  Returns (histogram, BatchStats) where the
    histogram is a dense int64 HArray (k <= 13; summed over ranks if ``group``) or a (keys, counts) pair
    of HArrays (k > 13; range-partitioned over ranks if ``group``)."""
    assert 0 < k < 32, "k must be larger than 0 and smaller than 32"
    ops = get_ops()
    lpe = buffer_type.n_lines_per_entry
    distributed = group is not None or _world_size() > 1
    key_bits = 2 * k
    if k > DENSE_MAX_K and fused:                                                                     # A2-A7 fused
        packed, ends, n, n_bases = ops.fastq_encode(text, text.size, lpe, 1, ord(buffer_type.HEADER),
                                                    buffer_type._check_plus)
        starts_mask, n_kmers = ops.kmer_starts_from_ends(ends, n_bases, k)
        del ends
        stats = BatchStats(n, n_bases, n_kmers, text.size)
        if distributed:                                                                               # A9 sparse, N GPUs
            # generate the hashes already partitioned by their top 8 bits == grouped by owning rank
            part, cuts = ops.kmers_partitioned(packed, starts_mask, n_bases, n_kmers, k, parallel.FINE_BITS, canonical=canonical)
            del packed, starts_mask
            holder = [part]                   # hand the 8 B/k-mer buffer over: it is freed right after the exchange
            del part
            return parallel.count_sparse_distributed(holder, key_bits, group, cuts=cuts), stats
        # canonical k-mers are not spread evenly: min(h, rc(h)) has density 2(1 - x) over the key range, so the levels
        # are planned for twice the keys
        skew = 2.0 if canonical else 1.0
        levels = ops.radix_plan(int(n_kmers * skew), key_bits)
        if levels:                                                                                    # A8 + A9 sparse
            hashes, cuts = ops.kmers_partitioned(packed, starts_mask, n_bases, n_kmers, k, levels[0], canonical=canonical)
            del packed, starts_mask
            return ops.count_sparse(hashes, key_bits=key_bits, consume=True, partition=(cuts, levels[0]), skew=skew), stats
        hashes, _ = ops.kmers_partitioned(packed, starts_mask, n_bases, n_kmers, k, 0, canonical=canonical)
        del packed, starts_mask
        return ops.count_sparse(hashes, key_bits=key_bits, consume=True, skew=skew), stats
    scan = ops.scan_lines(text, text.size, lpe, ord(buffer_type.HEADER), buffer_type._check_plus)   # A2 + A3
    n = scan.n_records
    starts, lens = ops.field_table(text, scan.newlines, n, lpe, 1, buffer_type._line_offsets[1], scan.has_cr)  # A4
    n_bytes = scan.size
    del scan
    offsets, n_bases = ops.row_offsets(lens, 1)
    _, packed = ops.gather_encode_dna(text, starts, offsets, n, n_bases, want_codes=False, want_packed=True)  # A6+A7
    del starts
    out_offsets, n_kmers = ops.row_offsets(lens, k)
    del lens
    stats = BatchStats(n, n_bases, n_kmers, n_bytes)
    if k <= DENSE_MAX_K:                                                                              # A8 + A9 dense
        hashes = ops.kmers(packed, offsets, out_offsets, n, n_kmers, k)
        del packed, offsets, out_offsets
        if canonical and n_kmers:                            # min(h, hash of the reverse complement k-mer), in place
            hashes = ops.canonical_kmers(hashes, k)
        hist = ops.count_dense(hashes, 4 ** k)
        del hashes
        if distributed:
            hist = parallel.allreduce_dense(hist, group)
        return hist, stats
    if distributed:                                                                                   # A9 sparse, N GPUs
        # generate the hashes already partitioned by their top 8 bits == grouped by owning rank
        ends = ops.kmer_start_mask(offsets, n, n_bases, k)
        del offsets, out_offsets
        part, cuts = ops.kmers_partitioned(packed, ends, n_bases, n_kmers, k, parallel.FINE_BITS, canonical=canonical)
        del packed, ends
        holder = [part]                       # hand the 8 B/k-mer buffer over: it is freed right after the exchange
        del part
        return parallel.count_sparse_distributed(holder, key_bits, group, cuts=cuts), stats
    hashes = ops.kmers(packed, offsets, out_offsets, n, n_kmers, k)
    del packed, offsets, out_offsets
    if canonical and n_kmers:
        hashes = ops.canonical_kmers(hashes, k)
    return ops.count_sparse(hashes, key_bits=key_bits, consume=True, skew=2.0 if canonical else 1.0), stats


def fastq_kmer_histogram_virtual_ranks(texts, k, buffer_type=FastQBuffer, canonical=False, plan="auto", with_plan=False,
                                       groups=None):
    """The sparse N-GPU path with the N ranks played one after the other by a single GPU: texts[r] is rank
    r's shard of the reads.  Every virtual rank decodes its shard and generates its k-mer hashes grouped by the top
    parallel.FINE_BITS bits (the send cuts); parallel.count_sparse_virtual stands in for the exchange and counts every
    rank's key range — or, for duplicate-heavy k-mers (plan "counts"; "auto" probes the data as the ranks would), every
    rank's own histogram is cut at the range boundaries and the runs are merged.  Returns ([(keys, counts)] per rank,
    [BatchStats] per rank, int64 words received per rank[, the plan])."""
    assert k > DENSE_MAX_K
    ops = get_ops()
    lpe = buffer_type.n_lines_per_entry
    shards, stats = [], []
    for text in texts:
        packed, ends, n, n_bases = ops.fastq_encode(text, text.size, lpe, 1, ord(buffer_type.HEADER), buffer_type._check_plus)
        starts_mask, n_kmers = ops.kmer_starts_from_ends(ends, n_bases, k)
        del ends
        shards.append(ops.kmers_partitioned(packed, starts_mask, n_bases, n_kmers, k, parallel.FINE_BITS, canonical=canonical))
        stats.append(BatchStats(n, n_bases, n_kmers, text.size))
        del packed, starts_mask
    hists, received, plan = parallel.count_sparse_virtual(shards, 2 * k, plan, groups)
    return (hists, stats, received, plan) if with_plan else (hists, stats, received)


def fastq_minimizers(text, k, window_size, buffer_type=FastQBuffer):
    """BASELINE config 3 as a pipeline: text (HArray uint8, complete FASTQ records) -> the minimizers of every window
    of ``window_size`` bases of every read, ragged-flat (get_minimizers, sequence/minimizers.py:8-54), through the
    fused decode: no newline table, no field tables, no row offsets — the window starts come from the read-end mask.
    Returns (int64 HArray, BatchStats)."""
    assert 0 < k < 32 and window_size >= k
    ops = get_ops()
    if window_size - k + 1 > 26:                            # (HipOps.WINDOWS_FLAT_MAX: what bnpk_windows_flat covers)
        # wider windows: the row-lookup kernel behind get_minimizers (bnpk_minimizers), which takes any width — through
        # the API objects, since it wants row offsets and the fused decode has none
        from .encoded_array import change_encoding, DNAEncoding
        from .sequence.minimizers import get_minimizers
        buf = buffer_type.from_raw_buffer(text)
        seqs = change_encoding(buf.get_field_by_number(1), DNAEncoding)
        out = get_minimizers(seqs, k, window_size)
        out._compact()
        return out._flat_data(), BatchStats(len(seqs), seqs.total(), out.total(), buf.size)
    packed, ends, n, n_bases = ops.fastq_encode(text, text.size, buffer_type.n_lines_per_entry, 1,
                                                ord(buffer_type.HEADER), buffer_type._check_plus)
    starts_mask, n_windows = ops.kmer_starts_from_ends(ends, n_bases, window_size)
    del ends
    out = ops.windows_from_mask(packed, starts_mask, n_bases, n_windows, k, window_size)
    return out, BatchStats(n, n_bases, n_windows, text.size)


def _world_size():
    try:
        import torch.distributed as dist
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    except Exception:
        return 1
