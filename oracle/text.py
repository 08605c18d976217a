"""FASTQ / FASTA chunk decode restated in numpy (test infrastructure — see oracle/__init__.py).

Follows, function by function:
  bionumpy/io/one_line_buffer.py:45-71   OneLineBuffer.from_raw_buffer
  bionumpy/io/one_line_buffer.py:140-152 _get_buffer_extractor
  bionumpy/io/one_line_buffer.py:156-173 _validate (header byte)
  bionumpy/io/one_line_buffer.py:176-182 _modify_for_carriage_return
  bionumpy/io/fastq_buffer.py:14-45      FastQBuffer (4 lines, '+' check)
  bionumpy/io/multiline_buffer.py:33-106 MultiLineFastaBuffer
  bionumpy/io/parser.py:89-206           NumpyFileReader.read / read_chunk / _get_buffer
"""
import gzip
from collections import namedtuple

import numpy as np

from .ragged import flat_indices

NEWLINE = 10
CR = 13


class FormatException(Exception):
    """bionumpy/io/exceptions.py:4-9"""

    def __init__(self, message, line_number=None):
        super().__init__(message)
        self.line_number = line_number


class IncompleteEntryException(Exception):
    """bionumpy/io/file_buffers.py:272-273"""


class EncodingError(Exception):
    """bionumpy/encodings/exceptions.py:1-4"""

    def __init__(self, message, offset=0):
        super().__init__(message)
        self.message = message
        self.offset = offset


LineFormat = namedtuple("LineFormat", "header n_lines line_offsets plus_line")
# FastQBuffer: HEADER '@', 4 lines/entry, _line_offsets (1,0,0,0) (io/fastq_buffer.py:15-18)
FASTQ = LineFormat(ord("@"), 4, (1, 0, 0, 0), True)
# TwoLineFastaBuffer: HEADER '>', 2 lines/entry, _line_offsets (1,0) (io/one_line_buffer.py:185-192,19)
TWO_LINE_FASTA = LineFormat(ord(">"), 2, (1, 0), False)

ScanResult = namedtuple(
    "ScanResult", "size n_lines n_records new_lines field_starts field_lens entry_starts entry_ends")


def scan_one_line_buffer(chunk, fmt=FASTQ):
    """OneLineBuffer.from_raw_buffer + _validate + _get_buffer_extractor.

    chunk: uint8 array.  Returns ScanResult; ``size`` is the number of bytes
    covered by complete entries (``buffer.size`` in the reference).
    """
    chunk = np.asarray(chunk, dtype=np.uint8)
    npe = fmt.n_lines
    new_lines = np.flatnonzero(chunk == NEWLINE)              # one_line_buffer.py:63
    n_lines = new_lines.size
    if n_lines < npe:                                         # :65-66
        raise IncompleteEntryException("No complete entry in buffer. Try increasing chunk_size.")
    new_lines = new_lines[: n_lines - (n_lines % npe)]        # :67
    data = chunk[: new_lines[-1] + 1]                         # :68-69
    _validate(data, new_lines, fmt)
    # _get_buffer_extractor (:140-152)
    line_starts = np.concatenate(([0], new_lines + 1))
    field_ends = new_lines.reshape(-1, npe)
    field_ends = _carriage_return_fix(field_ends, data, npe)
    field_starts = line_starts[:-1].reshape(-1, npe) + np.array(fmt.line_offsets, dtype=np.int64)
    entry_starts = line_starts[:-1:npe]
    entry_ends = line_starts[::npe][1:]
    return ScanResult(int(data.size), int(new_lines.size), int(new_lines.size // npe), new_lines,
                      field_starts.astype(np.int64), (field_ends - field_starts).astype(np.int64),
                      entry_starts, entry_ends)


def _validate(data, new_lines, fmt):
    npe = fmt.n_lines
    if data.size == 0 and new_lines.size == 0:
        return
    header_idxs = new_lines[npe - 1:-1:npe] + 1               # one_line_buffer.py:163-166
    bad = data[header_idxs] != fmt.header
    if np.any(bad) or data[0] != fmt.header:                  # :168-173
        if data[0] != fmt.header:
            line_number = 0
        else:
            line_number = (int(np.flatnonzero(bad)[0]) + 1) * npe
        raise FormatException("Expected header line to start with %s" % chr(fmt.header),
                              line_number=line_number)
    if fmt.plus_line:                                         # fastq_buffer.py:39-45
        bad_plus = data[new_lines[1::npe] + 1] != ord("+")
        if np.any(bad_plus):
            entry = int(np.flatnonzero(bad_plus)[0])
            raise FormatException("Expected '+' at third line of entry", line_number=2 + entry * npe)


def _carriage_return_fix(field_ends, data, npe):
    # one_line_buffer.py:176-182: only look at the header-line ends of the first npe entries
    if field_ends.size == 0 or field_ends[0, 0] < 1:
        return field_ends
    last_chars = data[field_ends[:npe, 0] - 1]
    if not np.any(last_chars == CR):
        return field_ends
    return field_ends - (data[field_ends - 1] == CR)


MultiLineResult = namedtuple(
    "MultiLineResult",
    "size n_lines n_records header_starts header_lens line_starts line_lens seq_lens")


def multiline_contains_complete_entry(chunks, marker=ord(">")):
    """MultiLineFastaBuffer.contains_complete_entry (io/multiline_buffer.py:33-44)."""
    ends_with_new_line = False
    for chunk in chunks:
        chunk = np.asarray(chunk, dtype=np.uint8)
        new_lines = np.flatnonzero(chunk[:-1] == NEWLINE)
        if np.count_nonzero(chunk[new_lines + 1] == marker) >= 1:
            return True
        if ends_with_new_line and chunk[0] == marker:
            return True
        ends_with_new_line = chunk[-1] == NEWLINE
    return False


def scan_multiline_fasta(chunk, marker=ord(">")):
    """MultiLineFastaBuffer.from_raw_buffer + get_data (io/multiline_buffer.py:89-101,46-62).

    Returns the table needed to build (headers, sequences): header text views
    and, for the sequence, the list of (line_start, line_len) to concatenate
    plus the per-record sequence length.
    """
    chunk = np.asarray(chunk, dtype=np.uint8)
    assert chunk[0] == marker
    new_lines = np.flatnonzero(chunk[:-1] == NEWLINE)         # :93
    new_entries = np.flatnonzero(chunk[new_lines + 1] == marker)
    if new_entries.size == 0:
        raise RuntimeError("No complete entry found in MultiLineFastaBuffer")
    entry_starts = new_lines[new_entries] + 1
    data = chunk[: entry_starts[-1]]                          # cut_chunk
    new_lines = new_lines[: new_entries[-1]]
    new_entries = new_entries[:-1]
    # get_data (:46-62)
    line_starts = np.concatenate(([0], new_lines + 1))
    line_ends = np.concatenate((new_lines, [data.size - 1]))
    if np.any(data[line_ends[:10] - 1] == CR):                # :103-106
        line_ends = line_ends - (data[line_ends - 1] == CR)
    line_lens = line_ends - line_starts
    header_lines = np.concatenate(([0], new_entries + 1))
    n_lines_per_entry = np.diff(np.concatenate((header_lines, [new_lines.size + 1]))) - 1
    is_header = np.zeros(line_starts.size, dtype=bool)
    is_header[header_lines] = True
    seq_line_starts = line_starts[~is_header]
    seq_line_lens = line_lens[~is_header]
    line_offsets = np.concatenate(([0], np.cumsum(n_lines_per_entry)))
    csum = np.concatenate(([0], np.cumsum(seq_line_lens)))
    seq_lens = csum[line_offsets[1:]] - csum[line_offsets[:-1]]
    return MultiLineResult(int(data.size), int(new_lines.size), int(header_lines.size),
                           (line_starts[header_lines] + 1).astype(np.int64),
                           (line_lens[header_lines] - 1).astype(np.int64),
                           seq_line_starts.astype(np.int64), seq_line_lens.astype(np.int64),
                           seq_lens.astype(np.int64))


def multiline_from_data(names, name_lens, seq_ascii, seq_lens, width=80, marker=ord(">")):
    """MultiLineFastaBuffer.from_data (io/multiline_buffer.py:68-86): the text of the records — '>' name newline, then the
    sequence in lines of n_characters_per_line = 80 letters.  (Records with an empty sequence, which the reference's index
    arithmetic does not handle, are written as their header line only.)"""
    names = np.asarray(names, dtype=np.uint8)
    seq_ascii = np.asarray(seq_ascii, dtype=np.uint8)
    name_lens = np.asarray(name_lens, dtype=np.int64)
    seq_lens = np.asarray(seq_lens, dtype=np.int64)
    n_lines = (seq_lens - 1) // width + 1                                     # :71
    last_length = (seq_lens - 1) % width + 1                                  # :72
    line_lengths = np.full(int(np.sum(n_lines)) + n_lines.size, width + 1, dtype=np.int64)   # :73
    entry_starts = np.insert(np.cumsum(n_lines + 1), 0, 0)                    # :74
    nonempty = seq_lens > 0
    line_lengths[entry_starts[1:][nonempty] - 1] = last_length[nonempty] + 1  # :76
    line_lengths[entry_starts[:-1]] = name_lens + 2                           # :75
    out = np.zeros(int(line_lengths.sum()), dtype=np.uint8)
    line_off = np.insert(np.cumsum(line_lengths), 0, 0)
    out[line_off[1:] - 1] = NEWLINE                                           # :85
    head = line_off[entry_starts[:-1]]
    out[head] = marker                                                        # :80
    name_off = np.insert(np.cumsum(name_lens), 0, 0)
    for r in range(name_lens.size):                                           # :79
        out[head[r] + 1: head[r] + 1 + name_lens[r]] = names[name_off[r]:name_off[r + 1]]
    is_seq_line = np.ones(line_lengths.size, dtype=bool)
    is_seq_line[entry_starts[:-1]] = False                                    # :81
    idx = np.flatnonzero(is_seq_line)
    dst = flat_indices(line_off[idx], line_lengths[idx] - 1)                  # :84
    out[dst] = seq_ascii[:dst.size]
    return out


class ChunkReader:
    """NumpyFileReader restated (io/parser.py:36-206) for the one-line and multi-line formats.

    ``fmt`` is FASTQ / TWO_LINE_FASTA or the string "multiline_fasta".
    ``read_chunk`` returns ``(raw_bytes_of_the_complete_entries, scan_result)`` or None.
    """

    def __init__(self, file_obj, fmt=FASTQ, prepend_mode=False):
        self._file = file_obj
        self._fmt = fmt
        self._multiline = fmt == "multiline_fasta"
        self._is_finished = False
        self._do_prepend = prepend_mode
        self._prepend = np.zeros(0, dtype=np.uint8)
        self.n_bytes_read = 0
        self.n_lines_read = 0

    # -- helpers ---------------------------------------------------------
    def _scan(self, chunk):
        if self._multiline:
            return scan_multiline_fasta(chunk)
        return scan_one_line_buffer(chunk, self._fmt)

    def _add_newline_to_end(self, chunk):
        # parser.py:183-190
        if chunk[-1] != NEWLINE:
            chunk = np.append(chunk, np.uint8(NEWLINE))
        if self._multiline:
            chunk = np.append(chunk, np.uint8(ord(">")))
        return chunk

    def _get_buffer(self, min_chunk_size):
        # parser.py:192-206
        a = np.frombuffer(self._file.read(min_chunk_size), dtype=np.uint8)
        self._is_finished = a.size < min_chunk_size
        if a.size == 0:
            return None
        if self._is_finished:
            a = self._add_newline_to_end(a)
        return a

    def _contains_complete_entry(self, chunks):
        if self._multiline:
            return multiline_contains_complete_entry(chunks), None
        if len(chunks) == 1:                                   # one_line_buffer.py:36-42
            try:
                return True, self._scan(chunks[0])
            except IncompleteEntryException:
                return False, None
        n = sum(int(np.count_nonzero(c == NEWLINE)) for c in chunks)   # file_buffers.py:264-267
        return n >= self._fmt.n_lines, None

    # -- public ------------------------------------------------------------
    def read(self):
        # parser.py:89-94
        chunk = np.frombuffer(self._file.read(), dtype=np.uint8)
        if chunk.size == 0:
            return None
        chunk = self._add_newline_to_end(chunk)
        res = self._scan(chunk)
        return chunk[:res.size], res

    def read_chunk(self, min_chunk_size=5000000, max_chunk_size=None):
        # parser.py:96-171
        complete = False
        temp_chunks = []
        if len(self._prepend):
            temp_chunks.append(self._prepend)
        made = None
        chunk = None
        while not complete:
            chunk = self._get_buffer(min_chunk_size)
            if chunk is None:
                return None
            temp_chunks.append(chunk)
            if max_chunk_size is not None and sum(c.size for c in temp_chunks) > max_chunk_size:
                raise Exception("No complete entry found")
            try:
                complete, made = self._contains_complete_entry(temp_chunks)
            except FormatException as e:
                e.line_number += self.n_lines_read
                raise
        if made is None:
            chunk = temp_chunks[0] if len(temp_chunks) == 1 else np.concatenate(temp_chunks)
            try:
                made = self._scan(chunk)
            except FormatException as e:
                e.line_number += self.n_lines_read
                raise
        self._prepend = np.zeros(0, dtype=np.uint8)
        if not self._is_finished:
            if not self._do_prepend:
                self._file.seek(made.size - chunk.size, 1)
            else:
                self._prepend = chunk[made.size:]
        if chunk.size:
            self.n_bytes_read += made.size
            self.n_lines_read += made.n_lines
            return chunk[:made.size], made
        return None

    def read_chunks(self, min_chunk_size=5000000, max_chunk_size=None):
        while not self._is_finished:
            out = self.read_chunk(min_chunk_size, max_chunk_size)
            if out is None:
                break
            yield out


def open_text(filename, fmt=None):
    """bnp_open suffix rules (io/files.py:28-49,177-182) for the formats on the path."""
    name = str(filename)
    parts = name.split("/")[-1].split(".")
    suffixes = ["." + p for p in parts[1:]]
    suffix = suffixes[-1]
    is_gzip = suffix == ".gz"
    if is_gzip:
        suffix = suffixes[-2]
    if fmt is None:
        if suffix in (".fq", ".fastq"):
            fmt = FASTQ
        elif suffix in (".fa", ".fasta", ".fna", ".faa"):
            fmt = "multiline_fasta"
        else:
            raise RuntimeError("File format %s does not have a default buffer type" % suffix)
    f = gzip.open(name, "rb") if is_gzip else open(name, "rb")
    return ChunkReader(f, fmt, prepend_mode=is_gzip)


def join_fields(fields, header, line_offsets):
    """OneLineBuffer.join_fields (bionumpy/io/one_line_buffer.py:119-134): fields = list of (flat uint8 bytes, row
    lengths), one per line of an entry; line i = line_offsets[i] header bytes (the HEADER on line 0) + the field's
    row + "\n".  Returns the text as uint8."""
    n = len(fields[0][1])
    starts = [np.cumsum(np.asarray(l, dtype=np.int64)) - np.asarray(l, dtype=np.int64) for _, l in fields]
    out = bytearray()
    for r in range(n):
        for i, (flat, lens) in enumerate(fields):
            if line_offsets[i]:
                out += bytes([header]) * line_offsets[i]
            s0 = int(starts[i][r])
            out += bytes(np.asarray(flat[s0:s0 + int(lens[r])], dtype=np.uint8).tobytes())
            out += b"\n"
    return np.frombuffer(bytes(out), dtype=np.uint8)
