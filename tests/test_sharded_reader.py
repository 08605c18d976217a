"""A file read by N ranks (bionumpy_amd/io/sharding.py): the ranks' entries, one behind the other, are the
entries of the plain reader, in order — plain files (byte ranges cut at record starts), BGZF (member-aligned ranges of the
compressed file), other gzip streams (every rank inflates, keeps every N-th chunk); line numbers stay those of the file.

The reference's rule for what starts a record is its validation, bionumpy/io/one_line_buffer.py:156-173 +
bionumpy/io/fastq_buffer.py:39-45 (a '@' line whose line + 2 starts with '+'); its unit of work is the chunk of
bionumpy/io/parser.py:96-171."""
import gzip
import os
import struct
import subprocess
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from backends import bnp  # noqa: E402,F401


# ---- files ---------------------------------------------------------------------------------------------------------------
def fastq_text(rng, n, max_len, at_quality=0.5, eol="\n", final_newline=True, min_len=1):
    """FASTQ records whose quality lines often START WITH '@' (the case the resync rule exists for) and contain '+' / '@'"""
    parts = []
    for i in range(n):
        ln = int(rng.integers(min_len, max_len))
        seq = "".join(rng.choice(list("ACGT"), size=ln))
        qual = "".join(rng.choice(list("@+IJ#5"), size=ln))
        if rng.random() < at_quality:
            qual = "@" + qual[1:]
        plus = "+" if rng.random() < 0.7 else "+read%d" % i
        parts.append("@read%d some text%s%s%s%s%s%s%s" % (i, eol, seq, eol, plus, eol, qual, eol))
    text = "".join(parts)
    return text if final_newline else text[:-len(eol)]


def fasta_text(rng, n, max_len, width=None):
    parts = []
    for i in range(n):
        ln = int(rng.integers(1, max_len))
        seq = "".join(rng.choice(list("ACGT"), size=ln))
        if width:
            seq = "\n".join(seq[j:j + width] for j in range(0, ln, width))
        parts.append(">seq%d\n%s\n" % (i, seq))
    return "".join(parts)


def _bgzf_member(piece):
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    payload = c.compress(piece) + c.flush()
    size = 12 + 6 + len(payload) + 8
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", size - 1) + payload
            + struct.pack("<II", zlib.crc32(piece), len(piece)))


def write_bgzf(path, data, member_text=None, rng=None):
    """a BGZF file: members of at most 64 KiB of text, each with its compressed size in a BC field, + the empty end member"""
    out, pos = [], 0
    while pos < len(data):
        n = member_text or (int(rng.integers(1, 60000)) if rng is not None else 60000)
        out.append(_bgzf_member(data[pos:pos + n]))
        pos += n
    out.append(_bgzf_member(b""))
    with open(path, "wb") as f:
        f.write(b"".join(out))


def entries_of(reader, chunk=None, quality=True):
    names, seqs, quals = [], [], []
    chunks = reader.read_chunks(min_chunk_size=chunk) if chunk else [reader.read()]
    for c in chunks:
        if len(c) == 0:
            continue
        names += c.name.tolist()
        seqs += c.sequence.tolist()
        if quality:
            quals += [q.tolist() for q in c.quality]
    return names, seqs, quals


# ---- the rule ------------------------------------------------------------------------------------------------------------
def test_record_start_rule_on_quality_lines_that_look_like_headers():
    from bionumpy_amd.io.sharding import RecordRule, first_record_start, record_start_at_or_after
    rule = RecordRule("@", 2)
    text = b"@r0\nACGT\n+\n@III\n@r1\nGG\n+r1\n@@\n@r2\nT\n+\n@\n"
    starts = [0, text.index(b"@r1"), text.index(b"@r2")]
    w = np.frombuffer(text, dtype=np.uint8)
    read_at = lambda off, n: text[off:off + n]
    for x in range(len(text) + 1):
        want = next((s for s in starts if s >= x), len(text))
        for probe in (1, 3, 7, 64):
            assert record_start_at_or_after(read_at, len(text), x, rule, probe=probe) == want, (x, probe)
    # a window that ends inside a record cannot decide: None (read more) unless it is the end of the file
    cut = text.index(b"@r1") + 6
    assert first_record_start(w[text.index(b"@III"):cut], True, rule, False) is None
    assert first_record_start(w[text.index(b"@III"):cut], True, rule, True) == -1
    # FASTA: a line that starts with '>'
    fa = b">a\nAC\nGT\n>b\nT\n"
    rule = RecordRule(">")
    for x in range(len(fa) + 1):
        want = next((s for s in (0, fa.index(b">b")) if s >= x), len(fa))
        assert record_start_at_or_after(lambda off, n: fa[off:off + n], len(fa), x, rule, probe=2) == want


def test_shards_tile_the_file(tmp_path):
    """B(x) is monotone: the ranks' byte ranges follow each other without a gap, whatever N is — also when a record is
    longer than S / N (ranks with nothing to read) and when S is not a multiple of N"""
    from bionumpy_amd.io.sharding import RecordRule, Shard, plain_byte_range
    rng = np.random.default_rng(3)
    for trial in range(30):
        text = fastq_text(rng, int(rng.integers(1, 40)), int(rng.choice([5, 80, 3000])), final_newline=bool(rng.random() < 0.7)).encode()
        path = tmp_path / ("t%d.fq" % trial)
        path.write_bytes(text)
        with open(path, "rb") as f:
            for world in (1, 2, 3, 7, 16, 64):
                ranges = [plain_byte_range(f.fileno(), len(text), Shard(r, world), RecordRule("@", 2)) for r in range(world)]
                assert ranges[0][0] == 0 and ranges[-1][1] == len(text)
                for (a, b), (c, d) in zip(ranges[:-1], ranges[1:]):
                    assert a <= b == c <= d
                for a, b in ranges:
                    assert a == len(text) or text[a:a + 5] == b"@read"


# ---- the readers ---------------------------------------------------------------------------------------------------------
CASES = [
    # seed, reads, max_len, eol, final newline, chunk size
    (1, 300, 120, "\n", True, None),
    (2, 300, 120, "\n", False, 3000),
    (3, 250, 90, "\r\n", True, 2000),
    (4, 12, 9000, "\n", True, 4096),            # records larger than S / N for the larger N
    (5, 1, 50, "\n", True, None),
    (6, 997, 40, "\n", False, 1 << 16),
]


@pytest.mark.parametrize("seed,n,max_len,eol,final_newline,chunk", CASES)
def test_plain_fastq_read_by_n_ranks(bnp, tmp_path, seed, n, max_len, eol, final_newline, chunk):
    rng = np.random.default_rng(seed)
    text = fastq_text(rng, n, max_len, eol=eol, final_newline=final_newline)
    path = tmp_path / "reads.fq"
    path.write_bytes(text.encode())
    want = entries_of(bnp.open(str(path), shard=False))
    assert len(want[0]) == n
    for world in (2, 3, 5, 8):
        got = ([], [], [])
        lines = 0
        for r in range(world):
            reader = bnp.open(str(path), shard=(r, world))
            part = entries_of(reader, chunk)
            # n_lines_read counts from the start of the FILE: the lines in front of the part + the part's own
            lines += 4 * len(part[0])
            if chunk:                                        # (read() of a whole file does not count lines, as in the reference)
                assert reader._reader.n_lines_read == lines, (world, r)
            for a, b in zip(got, part):
                a += b
        assert got == want, world


def test_big_batch_path_of_a_part(bnp, tmp_path, monkeypatch):
    """the read-ahead / placed-reads path (io/parser.py:_read_chunks_ahead) stops at the end of the part"""
    from bionumpy_amd.io import parser
    monkeypatch.setattr(parser, "_BIG", 1 << 12)
    monkeypatch.setattr(parser, "_FRONT", 1 << 9)
    monkeypatch.setattr(parser, "_PIECE", 1 << 11)
    rng = np.random.default_rng(11)
    text = fastq_text(rng, 900, 70)
    path = tmp_path / "reads.fq"
    path.write_text(text)
    want = entries_of(bnp.open(str(path), shard=False))
    for threads in (4, 1):
        monkeypatch.setattr(parser, "_READ_THREADS", threads)
        for world in (2, 3):
            got = ([], [], [])
            for r in range(world):
                for a, b in zip(got, entries_of(bnp.open(str(path), shard=(r, world)), 5000)):
                    a += b
            assert got == want, (threads, world)


def test_fasta_read_by_n_ranks(bnp, tmp_path):
    rng = np.random.default_rng(5)
    two_line = tmp_path / "two.fa"
    two_line.write_text(fasta_text(rng, 200, 300))
    multi = tmp_path / "multi.fa"
    multi.write_text(fasta_text(rng, 40, 900, width=60))
    for path, buffer_type in ((two_line, bnp.TwoLineFastaBuffer), (multi, None)):
        want = entries_of(bnp.open(str(path), buffer_type=buffer_type, shard=False), quality=False)
        for world in (2, 3, 7):
            got = ([], [], [])
            for r in range(world):
                for a, b in zip(got, entries_of(bnp.open(str(path), buffer_type=buffer_type, shard=(r, world)), 4000, quality=False)):
                    a += b
            assert got == want, (path.name, world)


def test_bgzf_read_by_n_ranks(bnp, tmp_path):
    rng = np.random.default_rng(8)
    for trial, (n, max_len, member) in enumerate([(400, 150, None), (60, 5000, 700), (3, 100, 50), (200, 100, 65000)]):
        text = fastq_text(rng, n, max_len).encode()
        path = tmp_path / ("reads%d.fq.gz" % trial)
        write_bgzf(str(path), text, member_text=member, rng=rng)
        assert gzip.open(str(path)).read() == text
        want = entries_of(bnp.open(str(path), shard=False))
        assert len(want[0]) == n
        for world in (2, 3, 5):
            got = ([], [], [])
            lines = 0
            for r in range(world):
                reader = bnp.open(str(path), shard=(r, world))
                part = entries_of(reader, 3000)
                lines += 4 * len(part[0])
                assert reader._reader.n_lines_read == lines
                for a, b in zip(got, part):
                    a += b
            assert got == want, (trial, world)


def test_plain_gzip_falls_back_to_every_nth_chunk(bnp, tmp_path):
    rng = np.random.default_rng(9)
    text = fastq_text(rng, 500, 100).encode()
    path = tmp_path / "reads.fq.gz"
    with gzip.open(str(path), "wb") as f:
        f.write(text)
    chunks = [c.name.tolist() for c in bnp.open(str(path), shard=False).read_chunks(min_chunk_size=4000)]
    for world in (2, 3):
        for r in range(world):
            mine = [c.name.tolist() for c in bnp.open(str(path), shard=(r, world)).read_chunks(min_chunk_size=4000)]
            assert mine == chunks[r::world]


def test_line_numbers_of_a_part_count_from_the_start_of_the_file(bnp, tmp_path):
    """tests/test_io_exceptions.py:86-100 of the reference, with the malformed entry in the part of a later rank"""
    rng = np.random.default_rng(10)
    good = fastq_text(rng, 200, 60, at_quality=0.0)
    lines = good.split("\n")
    bad_entry = 150
    lines[4 * bad_entry + 2] = "x" + lines[4 * bad_entry + 2][1:]          # the '+' line of entry 150
    path = tmp_path / "bad.fq"
    path.write_text("\n".join(lines))
    with pytest.raises(bnp.FormatException) as whole:
        bnp.open(str(path), shard=False).read()
    assert whole.value.line_number == 2 + 4 * bad_entry
    raised = []
    for r in range(4):
        try:
            for c in bnp.open(str(path), shard=(r, 4)).read_chunks(min_chunk_size=2000):
                c.sequence
        except bnp.FormatException as e:
            raised.append((r, e.line_number))
    # the resync skips a header whose '+' line is damaged, so the entry belongs to the rank in front of the cut — one rank
    # raises, with the line number the reader of the whole file reports
    assert len(raised) == 1 and raised[0][1] == 2 + 4 * bad_entry, raised


def test_auto_shard_is_off_without_a_process_group(bnp, tmp_path):
    from bionumpy_amd.io.sharding import resolve_shard
    assert resolve_shard(None) is None and resolve_shard(False) is None and resolve_shard((0, 1)) is None
    s = resolve_shard((2, 5))
    assert (s.rank, s.world) == (2, 5)
    with pytest.raises(ValueError):
        resolve_shard((5, 5))


def test_histogram_of_a_file_read_in_parts(bnp, tmp_path):
    """virtual ranks: every part of a real file counted on its own (no process group: nothing is merged behind the
    caller's back), the parts' histograms added up == the histogram of the file == the oracle's — dense and sparse, through
    the real kernels under -m gpu"""
    import oracle
    rng = np.random.default_rng(31)
    genome = "".join(rng.choice(list("ACGT"), size=20000))
    parts = []
    for i in range(6000):
        a = int(rng.integers(0, 19800))
        seq = genome[a:a + int(rng.integers(20, 160))]
        parts.append("@r%d\n%s\n+\n%s\n" % (i, seq, "@" * len(seq)))
    text = "".join(parts).encode()
    path = tmp_path / "reads.fq"
    path.write_bytes(text)
    host = np.frombuffer(text, dtype=np.uint8)
    res = oracle.scan_one_line_buffer(host, oracle.FASTQ)
    codes = oracle.encode_dna(oracle.gather_rows(host, res.field_starts[:, 1], res.field_lens[:, 1]))
    for k in (3, 31):
        h, _ = oracle.get_kmers(codes, res.field_lens[:, 1], k)
        whole = bnp.count_kmers(bnp.open(str(path), shard=False).read_chunks(min_chunk_size=200000).sequence, k)
        for world in (2, 3, 8):
            total, n_reads = 0, 0
            for r in range(world):
                n_reads += sum(len(c) for c in bnp.open(str(path), shard=(r, world)).read_chunks(min_chunk_size=100000))
                total = total + bnp.count_kmers(bnp.open(str(path), shard=(r, world)).read_chunks(min_chunk_size=100000).sequence, k)
            assert n_reads == 6000
            assert total == whole
            if k == 3:
                assert np.array_equal(np.asarray(total.counts), oracle.count_dense(h, 3))
            else:
                ek, ec = oracle.count_sparse(h)
                assert np.array_equal(total.keys, ek) and np.array_equal(total.counts, ec)


# ---- N processes (gloo): count_kmers(bnp.open(f).read_chunks().sequence, k) reads the file once ----------------------------
@pytest.mark.parametrize("ranks,port", [(2, 29741), (3, 29743)])
def test_count_kmers_of_a_sharded_file_over_gloo(tmp_path, ranks, port):
    rng = np.random.default_rng(21)
    genome = "".join(rng.choice(list("ACGT"), size=3000))
    parts = []
    for i in range(700):
        a = int(rng.integers(0, 2900))
        seq = genome[a:a + int(rng.integers(35, 100))]
        qual = "@" + "I" * (len(seq) - 1)
        parts.append("@r%d\n%s\n+\n%s\n" % (i, seq, qual))
    plain = tmp_path / "reads.fq"
    plain.write_text("".join(parts))
    write_bgzf(str(tmp_path / "reads.bgzf.fq.gz"), "".join(parts).encode(), rng=rng)
    with gzip.open(str(tmp_path / "reads.plain.fq.gz"), "wb") as f:
        f.write("".join(parts).encode())
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", BNPK_SHARD_TEST_DIR=str(tmp_path))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % ranks,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "shard_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "SHARD_OK" in r.stdout
