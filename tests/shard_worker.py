"""gloo worker for tests/test_sharded_reader.py: every rank opens the SAME files with ``bnp.open(path, shard="auto")`` — the
process group makes it read its own part (io/sharding.py; opt-in: plain ``bnp.open(path)`` stays the whole file on every
rank, as the reference reads it) — and ``count_kmers`` of the stream finishes with the merge over the
ranks (dense: all-reduce; sparse: (key, count) runs to the rank that owns their key range).  Host logic + collectives on
CPU with the oracle-backed ops; the kernels are covered by -m gpu."""
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
    import bionumpy_amd as bnp
    from bionumpy_amd import ops as ops_mod, parallel
    ops_mod.set_ops(OracleOps())
    d = os.environ["BNPK_SHARD_TEST_DIR"]

    text = np.frombuffer(open(os.path.join(d, "reads.fq"), "rb").read(), dtype=np.uint8)
    res = oracle.scan_one_line_buffer(text, oracle.FASTQ)
    codes = oracle.encode_dna(oracle.gather_rows(text, res.field_starts[:, 1], res.field_lens[:, 1]))
    n_reads = res.field_lens.shape[0]

    for name in ("reads.fq", "reads.bgzf.fq.gz", "reads.plain.fq.gz"):
        path = os.path.join(d, name)
        # every read is read by exactly one rank
        mine = sum(len(c) for c in bnp.open(path, shard="auto").read_chunks(min_chunk_size=6000))
        counts = [None] * world
        dist.all_gather_object(counts, mine)
        assert sum(counts) == n_reads, (name, counts)
        if name == "reads.fq":
            assert all(c > 0 for c in counts), counts          # (a plain file of 700 reads: nobody is idle)
        # no shard given (a script written against the reference), shard=False: the whole file on every rank
        assert sum(len(c) for c in bnp.open(path).read_chunks(min_chunk_size=6000)) == n_reads
        assert sum(len(c) for c in bnp.open(path, shard=False).read_chunks(min_chunk_size=6000)) == n_reads
        # read() and read_chunk() of a sharded reader give the rank's part too — also of a gzip stream, whose shard is every
        # world-th chunk (they used to return the whole file there, a merge over the ranks counted it N times)
        if name != "reads.plain.fq.gz":
            part = len(bnp.open(path, shard="auto").read())
            assert part == mine, (name, part, mine)
        else:
            part = len(bnp.open(path, shard="auto").read())
        parts = [None] * world
        dist.all_gather_object(parts, part)
        assert sum(parts) == n_reads, (name, parts)
        reader, got_n = bnp.open(path, shard="auto"), 0
        while True:
            c = reader.read_chunk(min_chunk_size=6000)
            if len(c) == 0:
                break
            got_n += len(c)
        assert got_n == mine, (name, got_n, mine)
        # plain numpy reductions of a sharded stream are merged too: bincount / histogram of the read lengths (through a
        # mapped stream: what is mapped over a rank's part stays a rank's part), mean quality
        whole = bnp.open(path).read()
        flat_q = np.concatenate([np.asarray(r) for r in whole.quality])
        lengths_of = bnp.streamable()(lambda sequence: sequence.lengths)
        got_bc = bnp.bincount(lengths_of(bnp.open(path, shard="auto").read_chunks(min_chunk_size=5000).sequence))
        assert np.array_equal(got_bc, np.bincount(whole.sequence.lengths)), name
        got_h, _ = bnp.histogram(lengths_of(bnp.open(path, shard="auto").read_chunks(min_chunk_size=5000).sequence), bins=5, range=(0, 500))
        assert np.array_equal(got_h, np.histogram(whole.sequence.lengths, bins=5, range=(0, 500))[0]), name
        got_mean = bnp.mean(bnp.open(path, shard="auto").read_chunks(min_chunk_size=5000).quality)
        assert np.allclose(got_mean, flat_q.mean(), rtol=1e-12), (name, got_mean, flat_q.mean())

        # dense: the reference's own example (scripts/kmer_counting_example.py), k = 3 — every rank gets the counts of the file
        got = bnp.count_kmers(bnp.open(path, shard="auto").read_chunks(min_chunk_size=5000).sequence, 3)
        h, _ = oracle.get_kmers(codes, res.field_lens[:, 1], 3)
        assert np.array_equal(np.asarray(got.counts), oracle.count_dense(h, 3)), name

        # sparse: k = 31 — rank r gets the keys of its range, with the counts of the whole file
        got = bnp.count_kmers(bnp.open(path, shard="auto").read_chunks(min_chunk_size=5000).sequence, 31)
        h, _ = oracle.get_kmers(codes, res.field_lens[:, 1], 31)
        ek, ec = oracle.count_sparse(h)
        lo, hi = parallel.key_range_of(rank, world, 62)
        assert got.key_range == (lo, hi)
        sel = (ek >= lo) & (ek < hi)
        assert np.array_equal(got.keys, ek[sel]) and np.array_equal(got.counts, ec[sel]), name
        whole = got.gathered()
        assert np.array_equal(whole.keys, ek) and np.array_equal(whole.counts, ec), name

    # replicated index, sharded queries (last bullet): every rank looks up the k-mers of ITS reads in the same index
    from bionumpy_amd.sequence.indexing import KmerIndex
    ref_path = os.path.join(d, "reads.fq")
    reference = bnp.open(ref_path, shard=False).read()
    index = KmerIndex.create_index(bnp.change_encoding(reference.sequence, bnp.DNAEncoding), 31)
    hits = 0
    for chunk in bnp.open(ref_path, shard="auto").read_chunks(min_chunk_size=20000):
        kmers = bnp.get_kmers(bnp.change_encoding(chunk.sequence, bnp.DNAEncoding), 31)
        hits += int(np.sum(np.asarray(index.count_hits(kmers.raw().ravel()))))
    total = [None] * world
    dist.all_gather_object(total, hits)
    # every k-mer of every read is in the index of all reads: its hits = the rows that hold it; summed over the file this is
    # sum over distinct (kmer, row) pairs of the k-mer's multiplicity... checked against the oracle's pair list
    rows = np.repeat(np.arange(n_reads), np.maximum(res.field_lens[:, 1] - 30, 0))
    pairs = np.unique(np.stack([h, rows]), axis=1)
    per_kmer = dict(zip(*np.unique(pairs[0], return_counts=True)))
    assert sum(total) == sum(per_kmer[x] for x in h.tolist())

    dist.barrier()
    if rank == 0:
        print("SHARD_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
