"""``bnp.open``: suffix -> buffer type, gzip detection (bionumpy/io/files.py:28-72,85-182,185-227)."""
import gzip
from pathlib import PurePath

from .buffers import FastQBuffer, MultiLineFastaBuffer, TwoLineFastaBuffer  # noqa: F401
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


def bnp_open(filename, mode=None, buffer_type=None, lazy=None):
    """Open a sequence file for chunked reading (io/files.py:85-182)."""
    if mode in ("w", "write", "wb", "a", "append", "ab"):
        raise NotImplementedError("writers are outside the MI355X hot path (SURVEY.md §2 row 2)")
    suffix, is_gzip = _split_suffix(filename)
    open_func = gzip.open if is_gzip else open
    if buffer_type is None:
        buffer_type = _get_buffer_type(suffix)
    file_reader = NumpyFileReader(open_func(filename, "rb"), buffer_type)
    if is_gzip:
        file_reader.set_prepend_mode()
    return NpDataclassReader(file_reader, lazy=lazy)


def count_entries(filename, buffer_type=None):
    """io/files.py:185-227"""
    suffix, is_gzip = _split_suffix(filename)
    open_func = gzip.open if is_gzip else open
    if buffer_type is None:
        buffer_type = _get_buffer_type(suffix)
    file_reader = NumpyFileReader(open_func(filename, "rb"), buffer_type)
    if is_gzip:
        file_reader.set_prepend_mode()
    return sum(chunk.count_entries() for chunk in file_reader.read_chunks(min_chunk_size=500000))
