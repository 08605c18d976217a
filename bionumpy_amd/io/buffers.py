"""FASTQ / FASTA chunk buffers on the MI355X path.

Same classes, class attributes and method names as the reference for the sequence formats:
  bionumpy/io/file_buffers.py:80-271       FileBuffer (base)
  bionumpy/io/one_line_buffer.py:13-192    OneLineBuffer, TwoLineFastaBuffer
  bionumpy/io/fastq_buffer.py:14-61        FastQBuffer
  bionumpy/io/multiline_buffer.py:15-109   MultiLineFastaBuffer

A buffer owns the raw chunk in HBM plus the newline table produced by the device scan
(``bnpk_byte_census`` / ``bnpk_byte_positions`` / ``bnpk_validate_entries``); fields are ragged *views*
into the chunk (start / length tables from ``bnpk_field_table``), exactly like the reference's
``RaggedView2`` extraction (io/file_buffers.py:335-338), so nothing is copied until a field is
encoded or ravelled.
"""
import numpy as np

from ..datatypes import SequenceEntry, SequenceEntryWithQuality
from ..device import HArray
from ..encoded_array import EncodedArray, EncodedRaggedArray, BaseEncoding, QualityEncoding, as_encoded_array
from ..exceptions import FormatException, IncompleteEntryException, NoCompleteEntry  # noqa: F401
from ..ops import get_ops

NEWLINE = 10


def _chunk_to_harray(chunk):
    """raw chunk (numpy uint8, EncodedArray or HArray) -> HArray of bytes"""
    if isinstance(chunk, HArray):
        return chunk
    if isinstance(chunk, EncodedArray):
        return chunk._harray()
    chunk = np.asarray(chunk, dtype=np.uint8)
    ops = get_ops()
    if not getattr(ops, "host_only", False):
        from .pinned import owner_of
        owner = owner_of(chunk)
        if owner is not None:                        # page-locked staging buffer -> hipMemcpyAsync into HBM
            return ops.upload_pinned(chunk, owner)
    return HArray(host=chunk)


class FileBuffer:
    """io/file_buffers.py:80-271 (the parts a sequence reader touches)"""

    COMMENT = 0
    dataclass = None

    @property
    def size(self):
        return self._size

    @property
    def data(self):
        return EncodedArray(self._data, BaseEncoding)

    @property
    def header_data(self):
        return None

    @classmethod
    def read_header(cls, file_object):
        return None                     # COMMENT == 0 for the sequence formats (file_buffers.py:155-156)

    @classmethod
    def modify_class_with_header_data(cls, header_data):
        return cls

    @classmethod
    def contains_complete_entry(cls, chunks):
        # io/file_buffers.py:264-267 (host chunks, before they are uploaded)
        n_new_lines = sum(int(np.count_nonzero(np.asarray(c) == NEWLINE)) for c in chunks)
        return n_new_lines >= cls.n_lines_per_entry


class BatchShare:
    """What the chunks that a reader cut out of ONE device batch have in common (io/parser.py:_cut_windows): the batch's
    buffer and, computed once on first request, a field of ALL its entries encoded as DNA.  The reference's loop encodes the
    sequence column of every 5 MB chunk on its own (``as_encoded_array(chunk.sequence, DNAEncoding)``,
    scripts/kmer_counting_example.py:5): here that is the gather + 2-bit encode of the whole batch, once; the rows of a chunk
    are a contiguous run of the batch's, so its k-mers (``windows``) are a part of the batch's, computed once per k, and its
    own packed words are cut out of the packed column (``bnpk_packed_rows_slice``: one small kernel, no round trip to the
    host) only if somebody asks for them.  An invalid base anywhere in the batch switches the shortcut off for the batch:
    every chunk then encodes itself and the EncodingError comes from the chunk that holds the base, with its own offset."""

    def __init__(self, big, cut_rows):
        self.big = big
        self.cut_rows = np.asarray(cut_rows, dtype=np.int64)   # entry indices at which chunks begin / end
        self._fields = {}

    def encoded_rows(self, encoding, line, j0, j1):
        state = self._fields.get(line)
        if state is None:
            from ..exceptions import EncodingError
            try:
                whole = encoding._encode_ragged(self.big._field_view(line))
                bases = get_ops().read_i64(whole.offsets(), self.cut_rows)
                state = (whole, dict(zip(self.cut_rows.tolist(), bases.tolist())))
            except EncodingError:
                state = False
            self._fields[line] = state
        if state is False:
            return None
        whole, base_at = state
        if j0 not in base_at or j1 not in base_at:
            return None
        from ..encoded_array import _LazyPackedDna, packed_words
        from ..device import LazyHArray
        b0, b1 = base_at[j0], base_at[j1]
        made = []                                            # (packed words, offsets) of the rows: ONE small kernel, on first use

        def part(i):
            if not made:
                made.extend(get_ops().packed_rows_slice(packed_words(whole._data), whole.total(), whole.offsets(), j0, j1 - j0,
                                                        b0, b1 - b0))
            return made[i]
        lens = HArray(dev=whole._lens.dev()[j0:j1])
        rows = EncodedRaggedArray._from_parts(_LazyPackedDna(lambda: part(0), b1 - b0), None, lens,
                                              LazyHArray(j1 - j0 + 1, lambda: part(1)), j1 - j0, b1 - b0, encoding)
        rows._trim_source = (self, line, j0, j1)            # (sequence/kmers.py:_rolling asks windows() below)
        return rows

    def _window_table(self, line, window):
        """(trimmed row offsets of the whole field, their values at the cut rows) for windows of ``window`` letters"""
        key = (line, window)
        table = self._fields.get(key)
        if table is None:
            ops = get_ops()
            off, _ = ops.row_offsets(self._fields[line][0]._lens, window)
            at = ops.read_i64(off, self.cut_rows)
            table = self._fields[key] = (off, dict(zip(self.cut_rows.tolist(), at.tolist())))
        return table

    def trimmed(self, line, window, j0, j1):
        """(n_out, offsets) of rows [j0, j1) of the encoded field after the trim to windows of ``window`` letters (kmers.py:100):
        the number of windows comes from ONE table per (field, window) over the whole batch, read at the cut rows once — a
        chunk's get_kmers then needs no answer from the device; its row offsets are only computed if somebody asks."""
        off, at = self._window_table(line, window)
        from ..device import LazyHArray
        return at[j1] - at[j0], LazyHArray(j1 - j0 + 1, lambda: HArray(dev=off.dev()[j0:j1 + 1] - off.dev()[j0]))

    def windows(self, line, k, window, j0, j1):
        """the k-mer hashes (window == k) / minimizers of rows [j0, j1) as a PART of the same values of the whole batch,
        computed once per (field, k, window): the rows of a chunk are a contiguous run of the batch's and the values are in
        row order, so a chunk's get_kmers launches nothing.  None where the position-flat generator does not reach."""
        key = ("values", line, k, window)
        values = self._fields.get(key)
        if values is None:
            from ..encoded_array import packed_words
            whole = self._fields[line][0]
            counted = get_ops().windows_counted(packed_words(whole._data), whole.offsets(), len(whole), k, window, whole.total(),
                                                getattr(whole, "_row_ends", None))
            values = self._fields[key] = False if counted is None else counted[0]
        if values is False:
            return None
        _, at = self._window_table(line, window)
        from ..device import SharedSlice
        part = values.dev()[at[j0]:at[j1]]
        if part.data_ptr() & 15:                             # (kernels with 16-byte loads want their input aligned: an odd start is copied)
            return HArray(dev=part.clone())
        return SharedSlice(dev=part)


class OneLineBuffer(FileBuffer):
    n_lines_per_entry = 2
    _line_offsets = (1, 0)
    HEADER = ">"
    _check_plus = False
    dataclass = SequenceEntry

    def __init__(self, data, scan, rows=None):
        self._data = data               # HArray: the chunk cut after the last complete entry
        self._scan = scan               # ops.LineScan
        self._size = scan.size
        self._rows = rows               # optional row selection: host int64 indices, or an HArray of them (device mask)

    @property
    def n_lines(self):
        return len(self) * self.n_lines_per_entry

    def __len__(self):
        return self._scan.n_records if self._rows is None else len(self._rows)

    def count_entries(self):
        return len(self)

    @classmethod
    def contains_complete_entry(cls, chunks):
        # one_line_buffer.py:36-42
        if len(chunks) == 1:
            try:
                return True, cls.from_raw_buffer(chunks[0])
            except IncompleteEntryException:
                return False
        return super().contains_complete_entry(chunks)

    @classmethod
    def from_raw_buffer(cls, chunk, header_data=None):
        """newline scan + validation on the device (one_line_buffer.py:45-71)"""
        assert header_data is None
        data = _chunk_to_harray(chunk)
        scan = get_ops().scan_lines(data, data.size, cls.n_lines_per_entry, ord(cls.HEADER), cls._check_plus)
        return cls(data, scan)

    def _field_table(self, line):
        return get_ops().field_table(self._data, self._scan.newlines, self._scan.n_records, self.n_lines_per_entry, line,
                                     self._line_offsets[line], self._scan.has_cr)

    def _field_view(self, line):
        share = getattr(self, "_share", None) if self._rows is None else None
        if share is not None:
            # a chunk cut out of a device batch (see BatchShare): a consumer that encodes the field as DNA takes the rows of
            # the batch's shared column and never looks at this chunk's own table — it is only filled in when read
            from ..device import LazyHArray
            n, made = self._scan.n_records, []
            table = lambda i: (made or made.extend(self._field_table(line)) or made)[i]
            starts, lens = LazyHArray(n, lambda: table(0)), LazyHArray(n, lambda: table(1))
        else:
            starts, lens = self._field_table(line)
        view = EncodedRaggedArray._from_parts(self._data, starts, lens, None, self._scan.n_records, None,
                                              BaseEncoding)
        if self._rows is None:
            if share is not None:
                view._batch_rows = (share[0], line, share[1], share[2])
            return view
        return view[self._rows.host() if isinstance(self._rows, HArray) else self._rows]

    def get_text_field_by_number(self, i):
        return self._field_view(i)

    def get_field_by_number(self, i, t=None):
        text = self.get_text_field_by_number(i)
        if t is not None and t is not str:
            return t.encode(text) if hasattr(t, "encode") else t(text)
        return text

    def get_data(self):
        return SequenceEntry(self.get_field_by_number(0), self.get_field_by_number(1))

    def __getitem__(self, idx):
        if isinstance(idx, HArray):                      # the ascending row list of a device mask (DeviceVector.nonzero_rows)
            if self._rows is None:
                return self.__class__(self._data, self._scan, idx)
            idx = idx.host()
        rows = np.arange(self._scan.n_records, dtype=np.int64) if self._rows is None else \
            (self._rows.host() if isinstance(self._rows, HArray) else self._rows)
        return self.__class__(self._data, self._scan, np.atleast_1d(rows[idx]))

    @classmethod
    def _text_column(cls, column):
        """(flat ASCII bytes, offsets) of a field: names / ASCII sequences as they are, encoded DNA decoded"""
        if not isinstance(column, EncodedRaggedArray):
            column = as_encoded_array(column)
        if column.encoding == BaseEncoding and not column.is_compact():
            # a column of the chunk's own text (the names of a chunk whose sequences were replaced): joined from where it
            # lies, not gathered first (bnpk_join_lines: d_field_starts)
            return column._flat_data(), column.offsets(), 0, column._starts
        column._compact()
        if column.encoding == BaseEncoding:
            return column._flat_data(), column.offsets(), 0
        total = column.total()
        from ..encoded_array import packed_words
        return get_ops().unpack_codes(packed_words(column._data), total, to_ascii=True), column.offsets(), 0

    @classmethod
    def _columns(cls, entries):
        return [cls._text_column(entries.name), cls._text_column(entries.sequence)]

    @classmethod
    def from_data(cls, entries):
        """text of the entries built from their fields (one_line_buffer.py:99-134; fastq_buffer.py:46-61) — for chunks
        whose fields have been replaced (``bnp.replace(chunk, sequence=...)``)"""
        cols = cls._columns(entries)
        lines = []
        for i, col in enumerate(cols):
            if col is None:
                lines.append((None, None, 0, cls._line_offsets[i], ord("+")))
            else:
                data, off, add = col[:3]
                lines.append((data, off, add, cls._line_offsets[i], 0) + tuple(col[3:4]))
        return get_ops().join_lines(len(entries), lines, ord(cls.HEADER))

    def entry_bytes(self):
        """the text of the (selected) entries as one contiguous buffer — TextThroughputExtractor._make_contigous
        (io/file_buffers.py:430-440) behind ``chunk[mask]`` + ``LazyBNPDataClass.get_buffer``
        (bnpdataclass/lazybnpdataclass.py:196-214): a compacting gather of whole records on the device"""
        if self._rows is None:
            return self._data if self._data.size == self._size else HArray(dev=self._data.dev()[:self._size])
        ops = get_ops()
        rows = self._rows if isinstance(self._rows, HArray) else HArray(host=np.asarray(self._rows, dtype=np.int64))
        if rows.size == 0:
            return HArray(host=np.zeros(0, dtype=np.uint8))
        starts, lens = ops.entry_table(self._scan.newlines, self.n_lines_per_entry, rows)
        offsets, total = ops.row_offsets(lens, 1)
        return ops.gather_rows(self._data, starts, offsets, rows.size, total, 0)


class TwoLineFastaBuffer(OneLineBuffer):
    """one_line_buffer.py:185-192"""
    HEADER = ">"
    n_lines_per_entry = 2
    dataclass = SequenceEntry


class FastQBuffer(OneLineBuffer):
    """fastq_buffer.py:14-45"""
    HEADER = "@"
    n_lines_per_entry = 4
    dataclass = SequenceEntryWithQuality
    _line_offsets = (1, 0, 0, 0)
    _check_plus = True

    def get_text_field_by_number(self, i):
        if i == 2:
            return self._field_view(3)
        return super().get_text_field_by_number(i)

    def get_field_by_number(self, i, t=None):
        if i == 2:
            return QualityEncoding.encode(self.get_text_field_by_number(i))
        return super().get_field_by_number(i, t)

    def get_data(self):
        return SequenceEntryWithQuality(self.get_field_by_number(0), self.get_field_by_number(1),
                                        self.get_field_by_number(2))

    @classmethod
    def _columns(cls, entries):
        quality = entries.quality
        pending = getattr(quality, "_pending", None)
        if pending is not None:                              # the quality line of the chunk's own text, never gathered: as it lies
            base, starts, subtract = pending
            qcol = (base, quality.offsets(), (33 - subtract) & 0xff, starts)
        else:
            quality._compact()
            qcol = (quality._data, quality.offsets(), 33)
        return [cls._text_column(entries.name), cls._text_column(entries.sequence), None, qcol]


class MultiLineFastaBuffer(FileBuffer):
    """multiline_buffer.py:15-106.  Everything per line happens on the device: the newline scan, the cut at the last
    newline that is followed by '>' (bnpk_multiline_cut), the header / sequence-line / record-length tables
    (bnpk_multiline_table: classification, three scans, one scatter) and the join of a record's sequence lines (ragged
    gather).  ``from_data`` (the writer: lines of 80 letters) is one output-flat kernel (bnpk_multiline_wrap)."""

    SKIP_LAZY = True
    _new_entry_marker = ">"
    n_characters_per_line = 80
    n_lines_per_entry = 2
    dataclass = SequenceEntry

    def __init__(self, data, size, newlines, n_newlines, n_entries):
        self._data = data
        self._size = size
        self._newlines = newlines              # HArray int64: positions of the newlines of the chunk (device)
        self._n_newlines = n_newlines          # ... of which the cut chunk keeps this many
        self._n_entries = n_entries

    @property
    def n_lines(self):
        return self._n_newlines

    @property
    def size(self):
        return self._size

    def count_entries(self):
        return self._n_entries

    def __len__(self):
        return self._n_entries

    @classmethod
    def contains_complete_entry(cls, chunks):
        # multiline_buffer.py:33-44 on the host chunks
        marker = ord(cls._new_entry_marker)
        ends_with_new_line = False
        for chunk in chunks:
            chunk = np.asarray(chunk)
            new_lines = np.flatnonzero(chunk[:-1] == NEWLINE)
            if np.count_nonzero(chunk[new_lines + 1] == marker) >= 1:
                return True
            if ends_with_new_line and chunk[0] == marker:
                return True
            ends_with_new_line = chunk[-1] == NEWLINE
        return False

    @classmethod
    def from_raw_buffer(cls, chunk, header_data=None):
        assert header_data is None, header_data
        ops = get_ops()
        data = _chunk_to_harray(chunk)
        n = data.size
        marker = ord(cls._new_entry_marker)
        first = ops.take_bytes(data, HArray(host=np.zeros(1, dtype=np.int64)), 0).host()[0]
        assert first == marker, "multi-line FASTA chunk must start with '>'"
        newlines, _ = ops.newline_positions(data, n - 1, 1)          # chunk[:-1] == "\n"            (:93)
        last, count = ops.multiline_cut(data, newlines, marker)      # chunk[new_lines + 1] == ">"    (:94)
        if count == 0:
            raise NoCompleteEntry("No complete entry found in %s. This can be due to badly formatted file, or "
                                  "because the buffer_size (%d) is too low. Try increasing buffer_size"
                                  % (cls.__name__, n))
        # cut at entry_starts[-1] = new_lines[new_entries[-1]] + 1; keep new_lines[:new_entries[-1]] (:98-101)
        cut = int(ops.read_i64(newlines, [last])[0]) + 1
        return cls(data, cut, newlines, last, count)

    def get_data(self):
        # multiline_buffer.py:46-62
        ops = get_ops()
        n_nl = self._n_newlines
        ends = ops.read_i64(self._newlines, np.arange(min(10, n_nl)))                     # ends of the first ten lines (:103-106)
        if n_nl < 10:
            ends = np.append(ends, self._size - 1)
        has_cr = bool(np.any(ops.take_bytes(self._data, HArray(host=np.ascontiguousarray(ends, dtype=np.int64)), -1).host()
                             == ord("\r")))
        hs, hl, rec_lens, ss, sl, n_bytes = ops.multiline_table(self._data, self._size, self._newlines, n_nl,
                                                                ord(self._new_entry_marker), has_cr)
        n = hs.size
        headers = EncodedRaggedArray._from_parts(self._data, hs, hl, None, n, None, BaseEncoding)
        # join the sequence lines of every record: gather the line views once on the device
        lines = EncodedRaggedArray._from_parts(self._data, ss, sl, None, ss.size, n_bytes, BaseEncoding)
        lines._compact()
        sequences = EncodedRaggedArray._from_parts(lines._data, None, rec_lens, None, n, n_bytes, BaseEncoding)
        return SequenceEntry(headers, sequences)

    @classmethod
    def from_data(cls, entries):
        """the text of the entries (multiline_buffer.py:68-86): '>' name, then the sequence in lines of 80 letters"""
        names, name_off, _ = OneLineBuffer._text_column(entries.name)
        seq, seq_off, _ = OneLineBuffer._text_column(entries.sequence)
        return get_ops().multiline_wrap(names, name_off, seq, seq_off, len(entries.name), cls.n_characters_per_line,
                                        ord(cls._new_entry_marker))
