"""Synthetic FASTQ of the benchmark's shape, defined as a pure function of (seed, read id, position).

Record = ``@%010d\\n`` + L bases + ``\\n+\\n`` + L * 'I' + ``\\n``  (2L + 16 bytes; 316 for L = 150).
mode 0 ("uniform"): bases i.i.d. uniform over ACGT — the distribution of the reference's own benchmark
generator (benchmarks/rules/simulation.smk:3-11).  mode 1 ("genome"): reads start at a hashed position
of a fixed pseudo-random genome of ``genome_len`` bases, so k-mers repeat like in real data.

The generator lives on the device (``bnpk_synth_fastq``, csrc/synth.hip) so that a 15.8 GB input never
crosses PCIe; this module is its numpy twin, used for small host-side cases (tests, the CPU baseline
sample) and checked byte-for-byte against the kernel in tests/test_gpu_parity.py.
"""
import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)
GENOME_SALT = np.uint64(0x67656E6F6D65)


def mix64(x):
    """splitmix64 finaliser on uint64 arrays (wrapping arithmetic)"""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def record_bytes(read_len):
    return 2 * read_len + 16


def read_codes(n_reads, read_len, seed, mode=0, genome_len=0, first_read=0):
    """(n_reads, read_len) uint8 codes 0..3 of the synthetic reads"""
    reads = np.arange(first_read, first_read + n_reads, dtype=np.uint64)
    pos = np.arange(read_len, dtype=np.uint64)
    with np.errstate(over="ignore"):
        read_key = mix64(np.uint64(seed) + reads)
        if mode == 0:
            stream = read_key[:, None]
            idx = np.broadcast_to(pos[None, :], (n_reads, read_len))
        else:
            start = read_key % np.uint64(genome_len - read_len + 1)
            stream = mix64(np.uint64(seed) ^ GENOME_SALT)
            idx = start[:, None] + pos[None, :]
        blk = mix64(stream + (idx >> np.uint64(5)))
    return ((blk >> (np.uint64(2) * (idx & np.uint64(31)))) & np.uint64(3)).astype(np.uint8)


def fastq_bytes(n_reads, read_len, seed, mode=0, genome_len=0, first_read=0):
    """the synthetic FASTQ text as a uint8 array (host twin of bnpk_synth_fastq)"""
    rec = record_bytes(read_len)
    out = np.empty((n_reads, rec), dtype=np.uint8)
    out[:, 0] = ord("@")
    ids = np.arange(first_read, first_read + n_reads, dtype=np.int64)
    for d in range(10):
        out[:, 1 + d] = ord("0") + (ids // 10 ** (9 - d)) % 10
    out[:, 11] = 10
    out[:, 12:12 + read_len] = np.frombuffer(b"ACGT", dtype=np.uint8)[
        read_codes(n_reads, read_len, seed, mode, genome_len, first_read)]
    out[:, 12 + read_len] = 10
    out[:, 13 + read_len] = ord("+")
    out[:, 14 + read_len] = 10
    out[:, 15 + read_len:rec - 1] = ord("I")
    out[:, rec - 1] = 10
    return out.reshape(-1)
