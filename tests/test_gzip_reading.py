"""The ``.gz`` readers of bionumpy_amd/io/gzip_reading.py (host code; the reference's counterpart is gzip / isal.igzip,
bionumpy/io/gzip_reading.py:1-4): BGZF members inflated by a thread pool, any other gzip stream by one thread ahead of
the reader — both must hand out exactly the bytes ``gzip`` hands out, in any read pattern, and fail loudly on truncation."""
import gzip
import os
import struct
import sys
import zlib

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from backends import bnp  # noqa: E402,F401

from bionumpy_amd.io import gzip_reading as gz


def bgzf_compress(data, block=65280, eof_marker=True):
    """what ``bgzip`` writes: gzip members of at most 64 KiB with their size in a BC extra field, then an empty member"""
    out = []
    for a in list(range(0, len(data), block)) + ([None] if eof_marker else []):
        piece = b"" if a is None else data[a:a + block]
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = c.compress(piece) + c.flush()
        size = 12 + 6 + len(body) + 8
        out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, size - 1)
                   + body + struct.pack("<II", zlib.crc32(piece), len(piece)))
    return b"".join(out)


def fastq_text(n, seed=5):
    rng = np.random.default_rng(seed)
    recs = []
    for i in range(n):
        ln = int(rng.integers(20, 200))
        seq = "".join("ACGT"[j] for j in rng.integers(0, 4, ln))
        qual = "".join(chr(33 + int(q)) for q in rng.integers(0, 40, ln))
        recs.append("@read%d\n%s\n+\n%s\n" % (i, seq, qual))
    return "".join(recs).encode()


def read_in_pieces(f, sizes):
    got, i = [], 0
    while True:
        buf = bytearray(sizes[i % len(sizes)])
        n = f.readinto(buf)
        got.append(bytes(buf[:n]))
        if n < len(buf):
            break
        i += 1
    return b"".join(got)


@pytest.fixture(params=["libdeflate", "zlib"], autouse=True)
def inflater(request, monkeypatch):
    """every test runs with both inflaters of the BGZF members: the system's libdeflate (where there is one) and zlib"""
    if request.param == "libdeflate":
        if gz._libdeflate is None:
            pytest.skip("no libdeflate.so.0 on this host")
    else:
        monkeypatch.setattr(gz, "_libdeflate", None)
    return request.param


@pytest.mark.parametrize("n_reads", [0, 1, 3000, 40000])
def test_bgzf_reader_hands_out_what_gzip_does(tmp_path, n_reads):
    data = fastq_text(n_reads) if n_reads else b""
    path = str(tmp_path / "reads.fq.gz")
    open(path, "wb").write(bgzf_compress(data))
    assert gzip.open(path, "rb").read() == data                      # (the fixture itself is a valid gzip file)
    f = gz.open_gzip_for_reading(path, n_threads=4)
    assert isinstance(f, gz.BgzfReader)
    assert read_in_pieces(f, [1, 7, 65536, 1 << 20, 333]) == data
    f.close()
    with gz.open_gzip_for_reading(path) as f:
        assert f.read() == data
    with gz.open_gzip_for_reading(path) as f:
        assert f.read(10) == data[:10] and f.read() == data[10:]


def test_bgzf_without_the_empty_last_member_and_truncated(tmp_path):
    data = fastq_text(5000)
    blob = bgzf_compress(data, eof_marker=False)
    path = str(tmp_path / "a.fq.gz")
    open(path, "wb").write(blob)
    with gz.open_gzip_for_reading(path) as f:
        assert f.read() == data
    open(path, "wb").write(blob[:-100])                              # the last member is cut short
    with pytest.raises((EOFError, OSError, zlib.error)):
        with gz.open_gzip_for_reading(path) as f:
            f.read()


def test_damaged_bgzf_member_is_an_error_not_wrong_text(tmp_path):
    """a member whose trailer does not match its text (CRC32 / length, as gzip.GzipFile and isal check them): the stored
    checksum of one member is changed — the deflate stream still inflates, the text must not be handed out"""
    data = fastq_text(5000)
    blob = bytearray(bgzf_compress(data))
    size = gz._bgzf_block_size(bytes(blob[:64]))
    blob[size - 8] ^= 0x5A                                           # first member: a bit pattern of its CRC32
    path = str(tmp_path / "damaged.fq.gz")
    open(path, "wb").write(bytes(blob))
    with pytest.raises(OSError):                                     # (gzip.BadGzipFile is an OSError)
        with gz.open_gzip_for_reading(path) as f:
            f.read()
    with pytest.raises(OSError):
        gzip.open(path, "rb").read()                                 # (the standard library agrees)
    # the text length in the trailer is wrong / the deflate stream itself is damaged
    for spoil in ("isize", "stream"):
        blob = bytearray(bgzf_compress(data))
        if spoil == "isize":
            blob[size - 4] ^= 0x01
        else:
            blob[size // 2] ^= 0xFF
        open(path, "wb").write(bytes(blob))
        with pytest.raises((OSError, zlib.error)):
            with gz.open_gzip_for_reading(path) as f:
                f.read()


@pytest.mark.parametrize("members", [1, 3])
def test_plain_gzip_streams_one_member_or_several(tmp_path, members):
    data = fastq_text(20000)
    cut = [len(data) * i // members for i in range(members + 1)]
    path = str(tmp_path / "b.fq.gz")
    open(path, "wb").write(b"".join(gzip.compress(data[a:b]) for a, b in zip(cut[:-1], cut[1:])))
    f = gz.open_gzip_for_reading(path)
    assert isinstance(f, gz.AheadGzipReader)
    assert read_in_pieces(f, [5, 1 << 16, 12345]) == data
    f.close()
    with gz.open_gzip_for_reading(path) as f:
        assert f.read() == data
    # closing a reader that was never read to its end must not hang on the thread that inflates ahead
    f = gz.open_gzip_for_reading(path)
    assert f.read(100) == data[:100]
    f.close()


def test_truncated_plain_gzip_is_an_error_not_a_short_file(tmp_path):
    blob = gzip.compress(fastq_text(20000))
    path = str(tmp_path / "c.fq.gz")
    open(path, "wb").write(blob[:len(blob) // 2])
    with pytest.raises((EOFError, zlib.error)):
        with gz.open_gzip_for_reading(path) as f:
            f.read()


def test_chunks_of_a_bgzf_fastq_equal_those_of_the_plain_file(bnp, tmp_path):
    data = fastq_text(6000, seed=9)
    plain, packed = str(tmp_path / "r.fq"), str(tmp_path / "r.fq.gz")
    open(plain, "wb").write(data)
    open(packed, "wb").write(bgzf_compress(data))
    a = bnp.open(plain).read()
    names, seqs, n = [], [], 0
    for chunk in bnp.open(packed).read_chunks(min_chunk_size=200_000):
        names += chunk.name.tolist()
        seqs += chunk.sequence.tolist()
        n += 1
    assert n > 3 and names == a.name.tolist() and seqs == a.sequence.tolist()
