"""``bnp.open``: suffix -> buffer type, gzip detection (bionumpy/io/files.py:28-72,85-182,185-227)."""
import gzip
from pathlib import PurePath

from .buffers import FastQBuffer, MultiLineFastaBuffer, TwoLineFastaBuffer  # noqa: F401
from .gzip_reading import open_gzip_for_reading
from .npdataclassreader import NpDataclassReader
from .parser import NumpyFileReader

buffer_types = {
    ".fasta": MultiLineFastaBuffer,
    ".fa": MultiLineFastaBuffer,
    ".fna": MultiLineFastaBuffer,
    ".faa": MultiLineFastaBuffer,
    ".fastq": FastQBuffer,
    ".fq": FastQBuffer,
}


def _get_buffer_type(suffix):
    if suffix in buffer_types:
        return buffer_types[suffix]
    raise RuntimeError("File format %s does not have a default buffer type on the MI355X sequence path. Specify "
                       "buffer_type (FastQBuffer, TwoLineFastaBuffer, MultiLineFastaBuffer) or use one of %s"
                       % (suffix, list(buffer_types.keys())))


def _split_suffix(filename):
    path = PurePath(filename)
    suffix = path.suffixes[-1]
    is_gzip = suffix == ".gz"
    if is_gzip:
        suffix = path.suffixes[-2]
    return suffix, is_gzip


def _file_reader(filename, buffer_type, is_gzip, shard=None):
    """NumpyFileReader over the whole file, or over one rank's part of it (io/sharding.py):
    plain file -> a byte range cut at record starts; BGZF -> members of a range of the compressed file, cut at record starts
    of the text; any other gzip stream -> the whole stream, of whose chunks this rank keeps every world-th"""
    from .sharding import RecordRule, plain_byte_range
    from .gzip_reading import is_bgzf
    import os
    if shard is None:
        reader = NumpyFileReader(open_gzip_for_reading(filename) if is_gzip else open(filename, "rb"), buffer_type)
    elif not is_gzip:
        f = open(filename, "rb")
        start, stop = plain_byte_range(f.fileno(), os.fstat(f.fileno()).st_size, shard, RecordRule.of(buffer_type))
        reader = NumpyFileReader(f, buffer_type, byte_range=(start, stop))
    elif is_bgzf(filename):
        text = open_gzip_for_reading(filename, shard=shard, rule=RecordRule.of(buffer_type))
        reader = NumpyFileReader(text, buffer_type, lines_before=text.lines_before)
    else:
        reader = NumpyFileReader(open_gzip_for_reading(filename), buffer_type, chunk_modulo=(shard.rank, shard.world))
    if is_gzip:
        reader.set_prepend_mode()
    return reader


def bnp_open(filename, mode=None, buffer_type=None, lazy=None, shard=None):
    """Open a sequence file for chunked reading (io/files.py:85-182).

    shard (extension): which part of the file this process reads when several ranks read it together.
    ``"auto"`` — under an initialised ``torch.distributed`` job with more than one rank, rank r's part (so that
    ``count_kmers(bnp.open(f, shard="auto").read_chunks().sequence, k)`` launched with torchrun reads the file ONCE and
    returns the histogram of the whole file: dense counts summed on every rank, sparse counts partitioned by key range
    over the ranks).  None (default) — the whole file on every rank, as the reference reads it, unless the environment
    says BNPK_SHARD=auto: sharding is opt-in, because only a reduction that ends in the merge over the ranks
    (``streamable`` reductions) gives the whole file's answer from a part per rank.  ``False`` — the whole file, whatever
    the environment says (a reference genome every rank indexes).  ``(rank, world)`` — that part, whatever the job is
    (no merge is attempted without a process group)."""
    from .sharding import resolve_shard
    suffix, is_gzip = _split_suffix(filename)
    open_func = gzip.open if is_gzip else open
    if buffer_type is None:
        buffer_type = _get_buffer_type(suffix)
    if mode in ("w", "write", "wb", "a", "append", "ab"):
        return NpBufferedWriter(open_func(filename, "ab" if mode in ("a", "append", "ab") else "wb"), buffer_type)
    shard = resolve_shard(shard)
    return NpDataclassReader(_file_reader(filename, buffer_type, is_gzip, shard), lazy=lazy, shard=shard)


class NpBufferedWriter:
    """File writer for chunks that still carry their text buffer (io/parser.py:209-268): the read-filter path
    ``out.write(chunk[mask])`` (scripts/small_example.py:36-46).  The selected records are gathered into one
    contiguous buffer on the device and written as they are; chunks whose fields were replaced
    (``bnp.replace(chunk, sequence=rc)``) are rebuilt from their fields (``from_data`` -> ``bnpk_join_lines``)."""

    def __init__(self, file_obj, buffer_type):
        self._file_obj = file_obj
        self._buffer_type = buffer_type

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.close()

    def close(self):
        if self._file_obj:
            self._file_obj.close()
            self._file_obj = None

    def write(self, data):
        if not hasattr(data, "get_buffer") and hasattr(data, "__iter__"):      # a stream of chunks
            for chunk in data:
                if len(chunk) > 0:
                    self.write(chunk)
            return
        if len(data) == 0:
            return
        buf = data.get_buffer() if hasattr(data, "get_buffer") else data
        if buf is not None and isinstance(buf, self._buffer_type) and hasattr(buf, "entry_bytes"):
            text = buf.entry_bytes()                           # untouched records: a compacting gather
        else:
            text = self._buffer_type.from_data(data)           # edited fields / another format: join the fields
        self._file_obj.write(text.host().tobytes())


def count_entries(filename, buffer_type=None, shard=False):
    """io/files.py:185-227 (shard: as in ``bnp_open`` — the entries of this rank's part; default: of the whole file)"""
    from .sharding import resolve_shard
    suffix, is_gzip = _split_suffix(filename)
    if buffer_type is None:
        buffer_type = _get_buffer_type(suffix)
    file_reader = _file_reader(filename, buffer_type, is_gzip, resolve_shard(shard))
    return sum(chunk.count_entries() for chunk in file_reader.read_chunks(min_chunk_size=500000))
