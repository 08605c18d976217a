"""The reader's path for big batches of a plain file (io/parser.py: parallel ``preadv`` into the staging buffer, a
background thread that reads the next batch's bytes while the current one is parsed): with the thresholds turned down it
must deliver the same entries as the serial path — long entries, batches without a complete entry, a file that ends on a
batch boundary, no final newline, an abandoned iteration."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from backends import bnp  # noqa: E402,F401


def _fastq(rng, n, max_len, final_newline=True):
    parts = []
    for i in range(n):
        ln = int(rng.integers(1, max_len))
        seq = "".join(rng.choice(list("ACGT"), size=ln))
        parts.append("@read%d\n%s\n+\n%s\n" % (i, seq, "I" * ln))
    text = "".join(parts)
    return text if final_newline else text[:-1]


@pytest.fixture(params=[(4, 2), (4, 1), (1, 2)], ids=["two-ahead", "one-ahead", "serial-reads"])
def small_thresholds(monkeypatch, request):
    """the big-batch machinery at test sizes: two batches read ahead by placed native reads (the default), one batch ahead,
    and reads through the file object (one thread: no places in the file can be handed out beforehand)"""
    from bionumpy_amd.io import parser
    threads, depth = request.param
    monkeypatch.setattr(parser, "_BIG", 1 << 12)
    monkeypatch.setattr(parser, "_FRONT", 1 << 9)
    monkeypatch.setattr(parser, "_READ_THREADS", threads)
    monkeypatch.setattr(parser, "_READ_DEPTH", depth)
    monkeypatch.setattr(parser, "_PIECE", 1 << 11)
    return parser


@pytest.mark.parametrize("seed,n,max_len,chunk,final_newline", [
    (1, 400, 60, 5000, True), (2, 400, 60, 4096, False), (3, 50, 3000, 4096, True), (4, 30, 20000, 8192, True),
    (5, 1, 50, 4096, True), (6, 300, 100, 1 << 16, True)])
def test_big_batch_path_delivers_the_entries_of_the_serial_path(bnp, small_thresholds, tmp_path, seed, n, max_len, chunk,
                                                               final_newline):
    rng = np.random.default_rng(seed)
    text = _fastq(rng, n, max_len, final_newline)
    path = tmp_path / "reads.fq"
    path.write_text(text)
    serial = bnp.open(str(path)).read()
    got_names, got_seqs, n_chunks = [], [], 0
    reader = bnp.open(str(path))
    for c in reader.read_chunks(min_chunk_size=chunk):
        got_names += c.name.tolist()
        got_seqs += c.sequence.tolist()
        n_chunks += 1
    assert got_names == serial.name.tolist() and got_seqs == serial.sequence.tolist()
    assert n_chunks >= 1 and reader._reader.n_lines_read == 4 * n
    # the file ends exactly where a batch ends: a last entry padded so that the size is a multiple of the batch size
    head = _fastq(np.random.default_rng(seed + 100), 64, 40)
    name = "@z" if (len(head) + len("@z\n\n+\n\n")) % 2 == 0 else "@zz"
    fixed = len(head) + len(name) + 5                        # name, four newlines, '+'
    pad = ((-fixed) % chunk) // 2
    pad = pad if pad > 0 else chunk // 2
    exact = head + "%s\n%s\n+\n%s\n" % (name, "A" * pad, "I" * pad)
    assert len(exact) % chunk == 0
    path2 = tmp_path / "exact.fq"
    path2.write_text(exact)
    want = bnp.open(str(path2)).read().sequence.tolist()
    got = []
    for c in bnp.open(str(path2)).read_chunks(min_chunk_size=chunk):
        got += c.sequence.tolist()
    assert got == want


def test_abandoned_iteration_keeps_the_rest_of_the_file(bnp, small_thresholds, tmp_path):
    rng = np.random.default_rng(9)
    text = _fastq(rng, 500, 80)
    path = tmp_path / "reads.fq"
    path.write_text(text)
    whole = bnp.open(str(path)).read().sequence.tolist()
    reader = bnp.open(str(path))
    seqs = []
    for i, c in enumerate(reader.read_chunks(min_chunk_size=6000)):
        seqs += c.sequence.tolist()
        if i == 1:
            break                                            # leaves a read-ahead behind
    rest = reader.read_chunks(min_chunk_size=6000)
    for c in rest:
        seqs += c.sequence.tolist()
    reader.close()
    assert seqs == whole


def test_parallel_fill_reads_what_readinto_reads(small_thresholds, tmp_path):
    parser = small_thresholds
    data = np.random.default_rng(3).integers(0, 255, size=70_001, dtype=np.uint8)
    path = tmp_path / "blob.bin"
    data.tofile(path)
    r = parser.NumpyFileReader.__new__(parser.NumpyFileReader)
    r._file_obj = open(path, "rb")
    r._stream_mode = False
    r._f_name = str(path)
    out = np.zeros(50_000, dtype=np.uint8)
    if parser._READ_THREADS < 2:
        assert r._fill_parallel(out) is None                 # (one thread: the serial path)
    assert r._fill(out) == 50_000 and np.array_equal(out, data[:50_000])
    out2 = np.zeros(50_000, dtype=np.uint8)
    assert r._fill(out2) == 20_001 and np.array_equal(out2[:20_001], data[50_000:])
    assert r._fill(out2) == 0
    r._file_obj.close()


# ---- read_chunks with the reference's small windows, cut out of big device batches (parser._cut_windows) -----------------
def _windowed(monkeypatch, on, batch=1 << 16):
    from bionumpy_amd.io import parser
    monkeypatch.setattr(parser, "_WINDOWED", on)
    monkeypatch.setattr(parser, "_WINDOW_MIN", 256)
    monkeypatch.setattr(parser, "_WINDOW_BATCH", batch)
    monkeypatch.setattr(parser, "_FRONT", 1 << 9)
    monkeypatch.setattr(parser, "_PIECE", 1 << 11)
    return parser


def _chunks(bnp, path, chunk, max_chunk=None):
    """[(names of the chunk, n_lines_read behind it)] and the exception that ended the loop, if any"""
    reader = bnp.open(str(path))
    out, err = [], None
    try:
        for c in reader.read_chunks(min_chunk_size=chunk, max_chunk_size=max_chunk):
            out.append((c.name.tolist(), c.sequence.tolist(), reader._reader.n_lines_read, reader._reader.n_bytes_read))
    except Exception as e:                                    # noqa: BLE001
        err = (type(e).__name__, getattr(e, "line_number", None), str(e)[:40])
    reader.close()
    return out, err


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,max_len,chunk,batch,final_newline,crlf", [
    (1, 2000, 60, 5000, 1 << 16, True, False), (2, 2000, 60, 4096, 1 << 15, False, False),
    (3, 300, 3000, 4096, 1 << 16, True, False),            # entries longer than the window: it grows
    (4, 40, 20000, 8192, 1 << 16, True, False),            # ... and longer than the front room of a batch
    (5, 1, 50, 4096, 1 << 16, True, False), (6, 3000, 100, 3000, 1 << 16, True, True),
    (7, 5000, 40, 1000, 1 << 14, True, False)])            # more chunks per batch than one call cuts (batches of 200 windows)
def test_windowed_chunks_are_the_plain_loops_chunks(monkeypatch, tmp_path, seed, n, max_len, chunk, batch, final_newline, crlf):
    """chunk for chunk: the same entries, n_lines_read and n_bytes_read as reading, uploading and scanning one window at a
    time (bionumpy/io/parser.py:96-171)"""
    import bionumpy_amd as bnp
    from bionumpy_amd import ops as ops_mod
    ops_mod.set_ops(None)
    text = _fastq(np.random.default_rng(seed), n, max_len, final_newline)
    if crlf:
        text = text.replace("\n", "\r\n")
    path = tmp_path / "reads.fq"
    path.write_bytes(text.encode())
    _windowed(monkeypatch, False)
    want, err0 = _chunks(bnp, path, chunk)
    _windowed(monkeypatch, True, batch)
    got, err1 = _chunks(bnp, path, chunk)
    assert err0 is None and err1 is None
    assert len(got) == len(want) and got == want


@pytest.mark.gpu
def test_windowed_reader_raises_where_the_plain_loop_raises(monkeypatch, tmp_path):
    """a malformed entry in the middle of a batch: the chunks in front of it are yielded, then FormatException with the
    line number counted from the start of the file; a window that cannot hold an entry: 'No complete entry found'"""
    import bionumpy_amd as bnp
    from bionumpy_amd import ops as ops_mod
    ops_mod.set_ops(None)
    rng = np.random.default_rng(11)
    for bad in ("header", "plus"):
        parts = _fastq(rng, 1500, 60).split("\n")
        line = 4 * 900 + (0 if bad == "header" else 2)
        parts[line] = "x" + parts[line][1:]
        path = tmp_path / ("bad_%s.fq" % bad)
        path.write_text("\n".join(parts))
        _windowed(monkeypatch, False)
        want, err0 = _chunks(bnp, path, 5000)
        _windowed(monkeypatch, True)
        got, err1 = _chunks(bnp, path, 5000)
        assert err0 is not None and err0[0] == "FormatException" and err0[1] == line
        assert err1 == err0 and got == want and len(got) > 3
    path = tmp_path / "long.fq"
    path.write_text(_fastq(rng, 50, 40) + "@long\n%s\n+\n%s\n" % ("A" * 30000, "I" * 30000) + _fastq(rng, 50, 40))
    _windowed(monkeypatch, False)
    want, err0 = _chunks(bnp, path, 4096, max_chunk=16384)
    _windowed(monkeypatch, True)
    got, err1 = _chunks(bnp, path, 4096, max_chunk=16384)
    assert err0 is not None and "No complete entry" in err0[2] and err1 == err0 and got == want


@pytest.mark.gpu
@pytest.mark.parametrize("stop_at", [0, 3, 14])
def test_windowed_reader_abandoned_keeps_the_rest_of_the_file(monkeypatch, tmp_path, stop_at):
    """the caller stops in the middle of a batch (also in the file's last batch, whose read-ahead met the end of the file)
    and goes on with read_chunk / read_chunks: nothing is lost, nothing comes twice"""
    import bionumpy_amd as bnp
    from bionumpy_amd import ops as ops_mod
    ops_mod.set_ops(None)
    text = _fastq(np.random.default_rng(12), 900, 80, final_newline=False)
    path = tmp_path / "reads.fq"
    path.write_text(text)
    whole = bnp.open(str(path)).read().sequence.tolist()
    _windowed(monkeypatch, True, 1 << 15)
    reader = bnp.open(str(path))
    seqs = []
    for i, c in enumerate(reader.read_chunks(min_chunk_size=6000)):
        seqs += c.sequence.tolist()
        if i == stop_at:
            break
    c = reader.read_chunk(min_chunk_size=6000)
    if len(c):                                               # (stop_at = the last chunk: nothing is left)
        seqs += c.sequence.tolist()
    for c in reader.read_chunks(min_chunk_size=6000):
        seqs += c.sequence.tolist()
    assert reader._reader.n_lines_read == 4 * 900
    reader.close()
    assert seqs == whole


@pytest.mark.gpu
def test_windowed_chunks_on_random_files_and_chunk_sizes(monkeypatch, tmp_path):
    """40 random combinations of entry lengths (up to several windows), window size, batch size, line ends and final newline:
    the windowed reader's chunks — entries, n_lines_read, n_bytes_read per chunk — are the plain loop's"""
    import bionumpy_amd as bnp
    from bionumpy_amd import ops as ops_mod
    ops_mod.set_ops(None)
    rng = np.random.default_rng(2026)
    for trial in range(40):
        n = int(rng.integers(1, 1500))
        max_len = int(rng.choice([8, 60, 400, 5000]))
        chunk = int(rng.choice([300, 1000, 4096, 20000]))
        batch = int(rng.choice([1 << 12, 1 << 14, 1 << 16]))
        text = _fastq(np.random.default_rng(trial), n, max_len, final_newline=bool(rng.integers(0, 2)))
        if rng.random() < 0.25:
            text = text.replace("\n", "\r\n")
        path = tmp_path / ("r%d.fq" % trial)
        path.write_bytes(text.encode())
        _windowed(monkeypatch, False)
        want, err0 = _chunks(bnp, path, chunk)
        _windowed(monkeypatch, True, batch)
        got, err1 = _chunks(bnp, path, chunk)
        assert err0 is None and err1 is None and got == want, (trial, n, max_len, chunk, batch, len(got), len(want))


@pytest.mark.gpu
@pytest.mark.parametrize("crlf,bad_at", [(False, None), (True, None), (False, 700), (False, 3)])
def test_chunks_of_a_batch_share_the_encoded_sequence_column(monkeypatch, tmp_path, crlf, bad_at):
    """as_encoded_array(chunk.sequence, DNAEncoding) on the chunks cut out of a device batch (io/buffers.py: BatchShare — the
    batch's sequence column is encoded once, a chunk takes its rows out of the packed result): the same rows, k-mers and
    counts as encoding every chunk on its own; an invalid base raises EncodingError from the chunk that holds it, with the
    offset inside THAT chunk's flat sequence, after the chunks in front of it came out right"""
    import bionumpy_amd as bnp
    from bionumpy_amd import ops as ops_mod
    from bionumpy_amd.exceptions import EncodingError
    ops_mod.set_ops(None)
    rng = np.random.default_rng(31)
    n = 1500
    lens = rng.integers(1, 90, size=n)
    seqs = ["".join(rng.choice(list("ACGTacgt"), size=l)) for l in lens]
    if bad_at is not None:
        seqs[bad_at] = seqs[bad_at][:len(seqs[bad_at]) // 2] + "N" + seqs[bad_at][len(seqs[bad_at]) // 2:]
    text = "".join("@r%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)) for i, s in enumerate(seqs))
    if crlf:
        text = text.replace("\n", "\r\n")
    path = tmp_path / "reads.fq"
    path.write_bytes(text.encode())

    def run(share):
        parser = _windowed(monkeypatch, True, 1 << 15)
        monkeypatch.setattr(parser, "_SHARE", share)
        out, err = [], None
        reader = bnp.open(str(path))
        try:
            for c in reader.read_chunks(min_chunk_size=4000):
                enc = bnp.as_encoded_array(c.sequence, bnp.DNAEncoding)
                kmers = bnp.get_kmers(enc, 5)
                mins = bnp.get_minimizers(enc, 3, 7)
                # (the counts are taken BEFORE anything reads the k-mers by row: the row offsets of a shared chunk's k-mers
                #  and the field table of its text view are only made on first use — device.py: LazyHArray)
                counts = bnp.count_encoded(kmers, axis=None).counts.tolist()
                out.append((enc.tolist(), enc.lengths.tolist(), np.asarray(kmers.raw().ravel()).tolist(), counts,
                            [np.asarray(r).tolist() for r in kmers.raw()], kmers.lengths.tolist(),
                            [np.asarray(r).tolist() for r in mins.raw()], c.sequence.lengths.tolist(),
                            c.sequence[1:3].tolist() if len(c) > 3 else None))
        except EncodingError as e:
            err = (e.offset, str(e)[:30])
        reader.close()
        return out, err

    want, err0 = run(False)
    got, err1 = run(True)
    assert len(want) > 5 or bad_at == 3
    assert got == want and err1 == err0 and (err0 is None) == (bad_at is None)
    if bad_at is None:
        assert [s for chunk in got for s in chunk[0]] == [s.upper() for s in seqs]
