"""HipOps: the hot-path primitives on HBM-resident buffers, each a thin wrapper over the C-ABI
(include/bnpk.h).  Every method takes and returns ``HArray``s (or ints) so that the host logic
above it never touches torch or ctypes.

Host logic receives its ops object from ``get_ops()``.  There is exactly one product
implementation (this one) and it raises if no GPU / no libbnpk.so is present — no CPU fallback.
``set_ops`` exists so that the CPU-only test-suite can drive the *host* logic (chunk loop,
exception mapping, API classes) with an oracle-backed stand-in defined under tests/.
"""
import ctypes as C
from collections import namedtuple

import numpy as np

from . import _native
from ._native import lib, check, NONE
from .device import Device, HArray, ptr, torch as torch_mod
from .exceptions import EncodingError
from .exceptions import FormatException, IncompleteEntryException

NEWLINE = 10

LineScan = namedtuple("LineScan", "size n_lines n_records newlines has_cr")


class HipOps:
    def __init__(self, device=None):
        self.device = device or Device.get()
        self.ctx = self.device.ctx

    # -- small helpers -------------------------------------------------------------------------
    def _s(self):
        return self.device.stream()

    def _chk(self, status):
        check(status, self.ctx)

    def _empty(self, n, dtype):
        return self.device.empty(n, dtype)

    def device_memory_bytes(self):
        """HBM of the device this process counts on (what the batch limits of the lazy histograms are scaled to)"""
        return int(torch_mod().cuda.get_device_properties(self.device.tdev).total_memory)

    # -- host staging --------------------------------------------------------------------------------
    def _fetch(self, t, n=None):
        """the first n int64 words of a device tensor as a list of Python ints (bnpk_fetch_i64: the page-locked mailbox)"""
        n = t.numel() if n is None else n
        buf = (C.c_int64 * n)()
        self._chk(lib.bnpk_fetch_i64(self.ctx, ptr(t), n, buf, self._s()))
        return list(buf)

    def upload_pinned(self, array, pinned_buffer):
        """hipMemcpyAsync of a numpy view of a pinned staging buffer into a fresh HBM tensor"""
        n = array.size
        t = self._empty(n, np.uint8)
        src = C.c_void_p(array.__array_interface__["data"][0])
        stream = self._s()
        self._chk(lib.bnpk_copy_h2d_async(ptr(t), src, n, stream))
        pinned_buffer.in_flight = True
        pinned_buffer.stream = stream                # the reader syncs this stream before reusing the buffer
        return HArray(dev=t)

    # -- A2 + A3: newline scan and entry validation --------------------------------------------------
    def newline_positions(self, buf, n, limit_multiple=1):
        """positions of '\\n' in buf[:n], truncated to a multiple of ``limit_multiple`` lines"""
        d = buf.dev()
        tiles = lib.bnpk_scan_tiles(n)
        tile_off = self._empty(tiles + 1, np.int64)
        self._chk(lib.bnpk_byte_census(self.ctx, ptr(d), n, NEWLINE, ptr(tile_off), self._s()))
        total = self._fetch(tile_off[tiles:], 1)[0]
        n_lines = total - (total % limit_multiple)
        pos = self._empty(n_lines, np.int64)
        self._chk(lib.bnpk_byte_positions(self.ctx, ptr(d), n, NEWLINE, ptr(tile_off), n_lines, ptr(pos), self._s()))
        return HArray(dev=pos), total

    def scan_lines(self, buf, n, lines_per_entry, header, check_plus):
        """OneLineBuffer.from_raw_buffer + _validate on a device chunk (one_line_buffer.py:45-71,156-173): the newline census,
        then positions and validation in ONE pass over the text (bnpk_line_positions), one download for every host scalar."""
        d = buf.dev()
        tiles = lib.bnpk_scan_tiles(n)
        tile_off = self._empty(tiles + 1, np.int64)
        self._chk(lib.bnpk_byte_census(self.ctx, ptr(d), n, NEWLINE, ptr(tile_off), self._s()))
        total = self._fetch(tile_off[tiles:], 1)[0]
        if total < lines_per_entry:
            raise IncompleteEntryException("No complete entry in buffer. Try increasing chunk_size.")
        n_lines = total - (total % lines_per_entry)
        nl = self._empty(n_lines, np.int64)
        err = self._empty(4, np.int64)
        self._chk(lib.bnpk_line_positions(self.ctx, ptr(d), n, ptr(tile_off), n_lines, lines_per_entry, header,
                                          1 if check_plus else 0, ptr(nl), ptr(err), self._s()))
        err[3:4].copy_(nl[n_lines - 1:n_lines])          # last newline -> size, one D2H for all four
        e = self._fetch(err)
        if e[0] != NONE:
            raise FormatException("Expected header line to start with %s" % chr(header),
                                  line_number=int(e[0]) * lines_per_entry)
        if check_plus and e[1] != NONE:
            raise FormatException("Expected '+' at third line of entry",
                                  line_number=2 + int(e[1]) * lines_per_entry)
        return LineScan(int(e[3]) + 1, n_lines, n_lines // lines_per_entry, HArray(dev=nl), (e[2] & 3) == 1)

    def window_cuts(self, buf, scan, lines_per_entry, window, avail, finished, first_held, max_chunk, max_cuts=256):
        """the chunks a reader with windows of ``window`` bytes cuts out of a scanned batch (bnpk_window_cuts) and every
        chunk's newline table relative to its first byte (bnpk_rebase_lines) -> (host int64 words of bnpk_window_cuts,
        device tensor of the rebased newline positions).  One launch each and ONE download per batch."""
        out = self._empty(lib.bnpk_window_cuts_words(max_cuts), np.int64)
        nl = scan.newlines.dev()
        self._chk(lib.bnpk_window_cuts(self.ctx, ptr(buf.dev()), ptr(nl), scan.n_lines, lines_per_entry, window, avail,
                                       1 if finished else 0, first_held, max_chunk or 0, max_cuts, ptr(out), self._s()))
        rebased = self._empty(max(scan.n_lines, 1), np.int64)
        self._chk(lib.bnpk_rebase_lines(self.ctx, ptr(nl), scan.n_lines, ptr(out), lines_per_entry, ptr(rebased), self._s()))
        head = self._fetch(out, 4)                       # (chunks cut, status, ...: then only the rows that were written)
        return np.array(head + self._fetch(out[4:], 4 * head[0]), dtype=np.int64), rebased

    # -- A2-A7 fused (k-mer pipeline) -----------------------------------------------------------------------
    def fastq_encode(self, buf, n, lines_per_entry, seq_line, header, check_plus):
        """text -> (packed 2-bit reads, read-end bit mask, n_records, n_bases): newline scan, validation, sequence
        extraction and 2-bit packing in two reads of the text, with the reference's exceptions
        (one_line_buffer.py:45-71,156-173; alphabet_encoding.py:37-45)"""
        d = buf.dev()
        table = self._empty(lib.bnpk_fastq_table_words(n), np.int64)
        totals = (C.c_int64 * 4)()
        self._chk(lib.bnpk_fastq_census(self.ctx, ptr(d), n, lines_per_entry, seq_line, ptr(table), totals, self._s()))
        n_newlines, n_lines, n_bases = int(totals[0]), int(totals[1]), int(totals[2])
        if n_newlines < lines_per_entry:
            raise IncompleteEntryException("No complete entry in buffer. Try increasing chunk_size.")
        packed = self._empty(n_bases // 32 + 2, np.int64)
        ends = self._empty(n_bases // 64 + 2, np.int64)
        err = self._empty(3, np.int64)
        self._chk(lib.bnpk_fastq_encode(self.ctx, ptr(d), n, lines_per_entry, seq_line, header, 1 if check_plus else 0,
                                        ptr(table), n_lines, n_bases, ptr(packed), ptr(ends), ptr(err), self._s()))
        e = err.cpu().numpy()
        if e[0] != NONE:
            raise FormatException("Expected header line to start with %s" % chr(header),
                                  line_number=int(e[0]) * lines_per_entry)
        if check_plus and e[1] != NONE:
            raise FormatException("Expected '+' at third line of entry",
                                  line_number=2 + int(e[1]) * lines_per_entry)
        if e[2] != NONE:
            raise EncodingError("Error when encoding to AlphabetEncoding('ACGT'): invalid character at flat "
                                "offset %d" % int(e[2]), int(e[2]))
        return HArray(dev=packed), HArray(dev=ends), n_lines // lines_per_entry, n_bases

    def fastq_encode_into(self, text_t, n, lines_per_entry, seq_line, header, check_plus, packed_t, ends_t, base_offset):
        """fused decode of one chunk (torch uint8 tensor, n bytes of whole records) into the flat 2-bit stream /
        read-end mask of a batch at ``base_offset`` (a multiple of 64).  Returns (records, bases, error cell): the cell
        {bad header entry, bad '+' entry, bad base offset} is left on the device for the caller to look at later —
        nothing here waits for the encoder."""
        assert base_offset % 64 == 0
        table = self._empty(lib.bnpk_fastq_table_words(n), np.int64)
        totals = (C.c_int64 * 4)()
        self._chk(lib.bnpk_fastq_census(self.ctx, ptr(text_t), n, lines_per_entry, seq_line, ptr(table), totals, self._s()))
        n_newlines, n_lines, n_bases = int(totals[0]), int(totals[1]), int(totals[2])
        if n_lines != n_newlines or n_lines % lines_per_entry:
            raise FormatException("a chunk of the host-fed path must hold whole records", line_number=0)
        assert base_offset + n_bases + 128 <= packed_t.numel() * 32, "batch buffer too small"
        err = self._empty(3, np.int64)
        packed = packed_t[base_offset // 32:]
        ends = ends_t[base_offset // 64:]
        self._chk(lib.bnpk_fastq_encode(self.ctx, ptr(text_t), n, lines_per_entry, seq_line, header, 1 if check_plus else 0,
                                        ptr(table), n_lines, n_bases, ptr(packed), ptr(ends), ptr(err), self._s()))
        return n_lines // lines_per_entry, n_bases, err

    def kmer_starts_from_ends(self, row_ends, n_bases, k):
        """(k-mer start mask, number of k-mers) from the read-end mask"""
        mask = self._empty(n_bases // 64 + 2, np.int64)
        count = self._empty(1, np.int64)
        self._chk(lib.bnpk_kmer_starts_from_ends(self.ctx, ptr(row_ends.dev()), n_bases, k, ptr(mask), ptr(count),
                                                 self._s()))
        return HArray(dev=mask), int(count.item())

    def field_table(self, buf, newlines, n_entries, lines_per_entry, field, line_offset, strip_cr):
        starts = self._empty(n_entries, np.int64)
        lens = self._empty(n_entries, np.int64)
        self._chk(lib.bnpk_field_table(self.ctx, ptr(buf.dev()), ptr(newlines.dev()), n_entries, lines_per_entry,
                                       field, line_offset, 1 if strip_cr else 0, ptr(starts), ptr(lens), self._s()))
        return HArray(dev=starts), HArray(dev=lens)

    def entry_table(self, newlines, lines_per_entry, rows):
        """(starts, lens) of whole entries ``rows`` (header byte .. newline of the last line): index arithmetic on the
        newline table (device tensors; io/file_buffers.py:426-440 keeps entry_starts / entry_ends for this)"""
        m = rows.size
        starts, lens = self._empty(m, np.int64), self._empty(m, np.int64)
        self._chk(lib.bnpk_entry_table(self.ctx, ptr(newlines.dev()), lines_per_entry, ptr(rows.dev()), m, ptr(starts),
                                       ptr(lens), self._s()))
        return HArray(dev=starts), HArray(dev=lens)

    def take_bytes(self, buf, positions, delta):
        m = positions.size
        out = self._empty(m, np.uint8)
        self._chk(lib.bnpk_take_bytes(self.ctx, ptr(buf.dev()), ptr(positions.dev()), m, delta, ptr(out), self._s()))
        return HArray(dev=out)

    def read_i64(self, arr, indices):
        """a few elements of a device int64 array, on the host"""
        idx = np.asarray(indices, dtype=np.int64)
        if idx.size == 0:
            return np.zeros(0, dtype=np.int64)
        t = torch_mod()
        return arr.dev()[t.from_numpy(idx).to(arr.dev().device)].cpu().numpy()

    # -- A13: multi-line FASTA ------------------------------------------------------------------------
    def multiline_cut(self, buf, newlines, marker):
        """(index of the last newline followed by ``marker`` or -1, number of such newlines) — bnpk_multiline_cut"""
        last, count = C.c_int64(-1), C.c_int64(0)
        self._chk(lib.bnpk_multiline_cut(self.ctx, ptr(buf.dev()), ptr(newlines.dev()), newlines.size, marker,
                                         C.byref(last), C.byref(count), self._s()))
        return int(last.value), int(count.value)

    def multiline_table(self, buf, size, newlines, n_newlines, marker, strip_cr):
        """header views, record lengths and sequence-line views of a cut multi-line FASTA chunk (bnpk_multiline_table):
        (header_starts, header_lens, record_lens, seq_line_starts, seq_line_lens) HArrays + total sequence bytes"""
        m = n_newlines + 2
        outs = [self._empty(m, np.int64) for _ in range(5)]
        totals = (C.c_int64 * 3)()
        self._chk(lib.bnpk_multiline_table(self.ctx, ptr(buf.dev()), size, ptr(newlines.dev()) if n_newlines else None,
                                           n_newlines, marker, 1 if strip_cr else 0, *[ptr(o) for o in outs], totals,
                                           self._s()))
        n_rec, n_seq, n_bytes = int(totals[0]), int(totals[1]), int(totals[2])
        hs, hl, rl, ss, sl = outs
        return (HArray(dev=hs[:n_rec]), HArray(dev=hl[:n_rec]), HArray(dev=rl[:n_rec]), HArray(dev=ss[:n_seq]),
                HArray(dev=sl[:n_seq]), n_bytes)

    def multiline_wrap(self, names, name_offsets, seq, seq_offsets, n_records, width, marker):
        """the text of multi-line FASTA records (bnpk_multiline_wrap): '>' name, then the sequence in lines of ``width``"""
        out_off = self._empty(n_records + 1, np.int64)
        total = C.c_int64(0)
        args = (ptr(names.dev()), ptr(name_offsets.dev()), ptr(seq.dev()), ptr(seq_offsets.dev()), n_records, width, marker,
                ptr(out_off))
        self._chk(lib.bnpk_multiline_wrap(self.ctx, *args, None, 0, C.byref(total), self._s()))
        out = self._empty(int(total.value), np.uint8)
        if total.value:
            self._chk(lib.bnpk_multiline_wrap(self.ctx, *args, ptr(out), out.numel(), C.byref(total), self._s()))
        return HArray(dev=out)

    # -- ragged offsets ------------------------------------------------------------------------------
    def row_offsets(self, lens, window=1):
        """(offsets[n+1], total) of rows trimmed by window-1 (RaggedShape; kmers.py:100)"""
        n = lens.size
        off = self._empty(n + 1, np.int64)
        self._chk(lib.bnpk_row_offsets(self.ctx, ptr(lens.dev()) if n else None, n, window, ptr(off), self._s()))
        return HArray(dev=off), self._fetch(off[n:], 1)[0]

    def exclusive_scan(self, values):
        n = values.size
        out = self._empty(n + 1, np.int64)
        self._chk(lib.bnpk_exclusive_scan_i64(self.ctx, ptr(values.dev()) if n else None, n, ptr(out), self._s()))
        return HArray(dev=out)

    # -- A6 + A7 ----------------------------------------------------------------------------------------
    def _err_cell(self):
        cell = self._empty(1, np.int64)
        self._chk(lib.bnpk_fill_i64(self.ctx, ptr(cell), 1, NONE, self._s()))
        return cell

    def _raise_if_bad(self, cell):
        off = self._fetch(cell, 1)[0]
        if off != NONE:
            raise EncodingError("Error when encoding to AlphabetEncoding('ACGT'): invalid character at flat "
                                "offset %d" % off, off)

    def gather_encode_dna(self, buf, starts, offsets, n_rows, total, want_codes=False, want_packed=True, want_ends=False):
        """-> (codes, packed[, row-end bits: what row_end_mask would make of ``offsets``])"""
        codes = self._empty(total, np.uint8) if want_codes else None
        packed = self._empty(total // 32 + 2, np.int64) if want_packed else None
        ends = self._empty(total // 64 + 2, np.int64) if want_ends else None
        cell = self._err_cell()
        self._chk(lib.bnpk_gather_encode_dna(self.ctx, ptr(buf.dev()), buf.size, ptr(starts.dev()), ptr(offsets.dev()), n_rows,
                                             total, ptr(codes), ptr(packed), ptr(ends), ptr(cell), self._s()))
        self._raise_if_bad(cell)
        out = (HArray(dev=codes) if want_codes else None, HArray(dev=packed) if want_packed else None)
        return out + (HArray(dev=ends),) if want_ends else out

    def packed_rows_slice(self, packed, n_bases_in, offsets, first_row, n_rows, first_base, n_bases):
        """rows [first_row, first_row + n_rows) of a compact packed ragged array as one of their own (bnpk_packed_rows_slice)
        -> (packed words, offsets)"""
        out = self._empty(n_bases // 32 + 2, np.int64)
        off = self._empty(n_rows + 1, np.int64)
        self._chk(lib.bnpk_packed_rows_slice(self.ctx, ptr(packed.dev()), n_bases_in, ptr(offsets.dev()), first_row, n_rows,
                                             first_base, n_bases, ptr(out), ptr(off), self._s()))
        return HArray(dev=out), HArray(dev=off)

    def gather_rows(self, buf, starts, offsets, n_rows, total, subtract=0):
        out = self._empty(total, np.uint8)
        self._chk(lib.bnpk_gather_rows(self.ctx, ptr(buf.dev()), ptr(starts.dev()), ptr(offsets.dev()), n_rows, total,
                                       subtract, ptr(out), self._s()))
        return HArray(dev=out)

    def encode_dna_flat(self, ascii_bytes, want_codes=True, want_packed=True):
        n = ascii_bytes.size
        codes = self._empty(n, np.uint8) if want_codes else None
        packed = self._empty(n // 32 + 2, np.int64) if want_packed else None
        cell = self._err_cell()
        self._chk(lib.bnpk_encode_dna_flat(self.ctx, ptr(ascii_bytes.dev()), n, ptr(codes), ptr(packed), ptr(cell),
                                           self._s()))
        self._raise_if_bad(cell)
        return (HArray(dev=codes) if want_codes else None, HArray(dev=packed) if want_packed else None)

    def pack_codes(self, codes):
        n = codes.size
        packed = self._empty(n // 32 + 2, np.int64)
        self._chk(lib.bnpk_pack_codes(self.ctx, ptr(codes.dev()), n, ptr(packed), self._s()))
        return HArray(dev=packed)

    def unpack_codes(self, packed, n, to_ascii=False):
        out = self._empty(n, np.uint8)
        self._chk(lib.bnpk_unpack_codes(self.ctx, ptr(packed.dev()), n, 1 if to_ascii else 0, ptr(out), self._s()))
        return HArray(dev=out)

    # -- A8 / A11 ------------------------------------------------------------------------------------------
    WINDOWS_FLAT_MAX = 26         # k-mers per window the row-lookup-free generator covers
    GATHER_GIVES_ROW_ENDS = True  # gather_encode_dna(want_ends=True): the rows' end bits as a by-product

    def _windows_flat(self, packed, in_offsets, n_rows, n_out, k, window_size, total=None):
        """hashes (window_size == k) / minimizers through the position-flat generator: start mask + ranks"""
        out = self._empty(n_out, np.int64)
        if n_out == 0:
            return HArray(dev=out)
        if total is None:                                # (callers that know the number of bases say so: one round trip less)
            total = self._fetch(in_offsets.dev()[n_rows:], 1)[0]
        mask = self.kmer_start_mask(in_offsets, n_rows, total, window_size)
        self._chk(lib.bnpk_windows_flat(self.ctx, ptr(packed.dev()), ptr(mask.dev()), total, k, window_size - k + 1,
                                        n_out, ptr(out), self._s()))
        return HArray(dev=out)

    def windows_counted(self, packed, in_offsets, n_rows, k, window_size, total, row_ends=None):
        """(hashes (window_size == k) / minimizers of every window of window_size letters, their number) — the number comes
        back with the start mask (bnpk_kmer_starts_from_ends counts the bits it sets), so nobody has to scan the trimmed row
        lengths for it before the output can be allocated.  row_ends: the rows' end bits if somebody has them already
        (bnpk_gather_encode_dna).  None where the position-flat generator does not reach."""
        if window_size > 64 or window_size - k + 1 > self.WINDOWS_FLAT_MAX:
            return None
        if total == 0:
            return HArray(dev=self._empty(0, np.int64)), 0
        if row_ends is None:
            ends = self._empty(total // 64 + 2, np.int64)
            self._chk(lib.bnpk_row_end_mask(self.ctx, ptr(in_offsets.dev()), n_rows, total, ptr(ends), self._s()))
        else:
            ends = row_ends.dev()
        mask = self._empty(total // 64 + 2, np.int64)
        count = self._empty(1, np.int64)
        self._chk(lib.bnpk_kmer_starts_from_ends(self.ctx, ptr(ends), total, window_size, ptr(mask), ptr(count), self._s()))
        n_out = self._fetch(count, 1)[0]
        out = self._empty(n_out, np.int64)
        if n_out:
            self._chk(lib.bnpk_windows_flat(self.ctx, ptr(packed.dev()), ptr(mask), total, k, window_size - k + 1, n_out,
                                            ptr(out), self._s()))
        return HArray(dev=out), n_out

    def windows_from_mask(self, packed, start_mask, n_bases, n_out, k, window_size):
        """hashes (window_size == k) / minimizers of the windows marked in ``start_mask`` (bnpk_windows_flat)"""
        out = self._empty(n_out, np.int64)
        if n_out:
            self._chk(lib.bnpk_windows_flat(self.ctx, ptr(packed.dev()), ptr(start_mask.dev()), n_bases, k,
                                            window_size - k + 1, n_out, ptr(out), self._s()))
        return HArray(dev=out)

    def kmers(self, packed, in_offsets, out_offsets, n_rows, n_out, k, total=None):
        return self._windows_flat(packed, in_offsets, n_rows, n_out, k, k, total)

    def kmers_generic(self, codes, in_offsets, out_offsets, n_rows, n_out, k, alphabet_size):
        """hashes sum_j code[p + j] * alphabet_size^j of every window of k codes (bnpk_kmers_generic): the k-mers of
        alphabets that are not 4 letters wide"""
        out = self._empty(n_out, np.int64)
        if n_out:
            self._chk(lib.bnpk_kmers_generic(self.ctx, ptr(codes.dev()), ptr(in_offsets.dev()), ptr(out_offsets.dev()),
                                             n_rows, n_out, k, alphabet_size, ptr(out), self._s()))
        return HArray(dev=out)

    def minimizers_generic(self, codes, in_offsets, out_offsets, n_rows, n_out, k, window_size, alphabet_size):
        """the smallest (signed) generic k-mer hash of every window of window_size codes (bnpk_minimizers_generic)"""
        out = self._empty(n_out, np.int64)
        if n_out:
            self._chk(lib.bnpk_minimizers_generic(self.ctx, ptr(codes.dev()), ptr(in_offsets.dev()), ptr(out_offsets.dev()),
                                                  n_rows, n_out, k, window_size, alphabet_size, ptr(out), self._s()))
        return HArray(dev=out)

    def lut_bytes(self, data, lut, what="AlphabetEncoding"):
        """lut[data] on the device (bnpk_lut_bytes); EncodingError(offset of the first byte that maps to 255)"""
        n = data.size
        out = self._empty(n, np.uint8)
        if n:
            table = np.ascontiguousarray(lut, dtype=np.uint8)
            assert table.size == 256
            cell = self._err_cell()
            self._chk(lib.bnpk_lut_bytes(self.ctx, ptr(data.dev()), n, table.ctypes.data_as(C.c_void_p), ptr(out), ptr(cell),
                                         self._s()))
            off = int(cell.item())
            if off != NONE:
                raise EncodingError("Error when encoding to %s: invalid character at flat offset %d" % (what, off), off)
        return HArray(dev=out)

    def minimizers(self, packed, in_offsets, out_offsets, n_rows, n_out, k, window_size):
        if window_size - k + 1 <= self.WINDOWS_FLAT_MAX:
            return self._windows_flat(packed, in_offsets, n_rows, n_out, k, window_size)
        out = self._empty(n_out, np.int64)
        self._chk(lib.bnpk_minimizers(self.ctx, ptr(packed.dev()), ptr(in_offsets.dev()), ptr(out_offsets.dev()),
                                      n_rows, n_out, k, window_size, ptr(out), self._s()))
        return HArray(dev=out)

    # -- match_string ---------------------------------------------------------------------------------
    def match_windows(self, data, offsets, n_rows, total, n_out, pattern, packed):
        """uint8 0/1 per window of len(pattern) symbols, ragged-flat.  packed: ``data`` = 2-bit words and ``pattern``
        = codes (first symbol in the low bits of the hash); else ``data`` = bytes and ``pattern`` = bytes"""
        m = len(pattern)
        out = self._empty(n_out, np.uint8)
        if n_out == 0:
            return HArray(dev=out)
        mask = self.kmer_start_mask(offsets, n_rows, total, m)
        if packed:
            h = 0
            for j, c in enumerate(pattern):
                h |= int(c) << (2 * j)
            self._chk(lib.bnpk_match_windows_packed(self.ctx, ptr(data.dev()), ptr(mask.dev()), total, m, h, n_out,
                                                    ptr(out), self._s()))
        else:
            pat = (C.c_uint8 * m)(*[int(c) for c in pattern])
            self._chk(lib.bnpk_match_windows_bytes(self.ctx, ptr(data.dev()), ptr(mask.dev()), total, m, pat, n_out,
                                                   ptr(out), self._s()))
        return HArray(dev=out)

    def match_rows(self, packed, offsets, n_rows, total, pattern):
        """int64 per row: its windows of len(pattern) bases that equal the pattern, from the 2-bit words — the flags of
        match_windows summed per row without being written (bnpk_match_rows_packed); len(pattern) <= 31"""
        out = self._empty(n_rows, np.int64)
        h = 0
        for j, c in enumerate(pattern):
            h |= int(c) << (2 * j)
        if n_rows:
            self._chk(lib.bnpk_match_rows_packed(self.ctx, ptr(packed.dev()) if total else None, total, ptr(offsets.dev()), n_rows,
                                                 len(pattern), h, ptr(out), self._s()))
        return HArray(dev=out)

    def pwm_scores(self, packed, offsets, n_rows, total, n_out, matrix):
        """float64 motif score of every window of matrix.shape[1] bases, ragged-flat; matrix[code][position]"""
        width = matrix.shape[1]
        out = self._empty(n_out, np.float64)
        if n_out == 0:
            return HArray(dev=out)
        mask = self.kmer_start_mask(offsets, n_rows, total, width)
        m = np.ascontiguousarray(np.asarray(matrix, dtype=np.float64).T)          # [position][code]
        self._chk(lib.bnpk_pwm_scores(self.ctx, ptr(packed.dev()), ptr(mask.dev()), total, width,
                                      m.ctypes.data_as(C.c_void_p), n_out, ptr(out), self._s()))
        return HArray(dev=out)

    def col_sums_u8(self, data, offsets, n_rows, total, n_cols):
        """(sums int64[n_cols], counts int64[n_cols]) over the columns of ragged uint8 rows"""
        sums, counts = self._empty(n_cols, np.int64), self._empty(n_cols, np.int64)
        self._chk(lib.bnpk_col_sums_u8(self.ctx, ptr(data.dev()), ptr(offsets.dev()), n_rows, total, n_cols, ptr(sums),
                                       ptr(counts), self._s()))
        return HArray(dev=sums), HArray(dev=counts)

    # -- join_fields: record text from fields ------------------------------------------------------
    def join_lines(self, n_rows, lines, header):
        """lines: per line (data HArray | None, offsets HArray | None, add, prefix, fill byte[, starts HArray | None: the rows
        lie at data[starts[r]], not back to back]).  Returns the text (HArray uint8) of all entries: every line = prefix header
        bytes + field row (+ add) + newline."""
        n = len(lines)
        view_starts = [(l[5] if len(l) > 5 else None) for l in lines]
        lines = [l[:5] for l in lines]
        starts = (C.c_void_p * n)(*[ptr(v.dev()) if v is not None else None for v in view_starts])
        sizes = (C.c_int64 * n)(*[(d.size if (d is not None and v is not None) else 0) for (d, _, _, _, _), v in zip(lines, view_starts)])
        offs = (C.c_void_p * n)(*[ptr(o.dev()) if o is not None else None for _, o, _, _, _ in lines])
        prefixes = (C.c_int * n)(*[int(p) for _, _, _, p, _ in lines])
        lens = self._empty(n_rows, np.int64)
        self._chk(lib.bnpk_join_line_lens(self.ctx, n_rows, n, offs, prefixes, ptr(lens), self._s()))
        entry_off, total = self.row_offsets(HArray(dev=lens), 1)
        out = self._empty(total, np.uint8)
        datas = (C.c_void_p * n)(*[ptr(d.dev()) if d is not None else None for d, _, _, _, _ in lines])
        adds = (C.c_int * n)(*[int(a) for _, _, a, _, _ in lines])
        fills = (C.c_uint8 * n)(*[int(f) for _, _, _, _, f in lines])
        self._chk(lib.bnpk_join_lines(self.ctx, n_rows, n, datas, offs, starts, sizes, adds, prefixes, fills, header,
                                      ptr(entry_off.dev()), total, ptr(out), self._s()))
        return HArray(dev=out)

    # -- per-row reductions of ragged uint8 data --------------------------------------------------
    def row_reduce_u8(self, data, offsets, n_rows, want=("sum",)):
        """{name: HArray} for name in want ⊆ {sum (int64), min, max (uint8)}: one value per row"""
        outs = {"sum": self._empty(n_rows, np.int64) if "sum" in want else None,
                "min": self._empty(n_rows, np.uint8) if "min" in want else None,
                "max": self._empty(n_rows, np.uint8) if "max" in want else None}
        self._chk(lib.bnpk_row_reduce_u8(self.ctx, ptr(data.dev()), ptr(offsets.dev()), n_rows, ptr(outs["sum"]),
                                         ptr(outs["min"]), ptr(outs["max"]), self._s()))
        return {k: HArray(dev=v) for k, v in outs.items() if v is not None}

    def row_reduce_u8_view(self, data, starts, offsets, n_rows, subtract, want=("sum",)):
        """row_reduce_u8 over rows data[starts[r] .. + offsets[r+1] - offsets[r]) with ``subtract`` taken off every byte (uint8
        wrap-around) — the reductions of a column nobody gathered (bnpk_row_reduce_u8_view)"""
        outs = {"sum": self._empty(n_rows, np.int64) if "sum" in want else None,
                "min": self._empty(n_rows, np.uint8) if "min" in want else None,
                "max": self._empty(n_rows, np.uint8) if "max" in want else None}
        self._chk(lib.bnpk_row_reduce_u8_view(self.ctx, ptr(data.dev()), data.size, ptr(starts.dev()), ptr(offsets.dev()), n_rows,
                                              subtract, ptr(outs["sum"]), ptr(outs["min"]), ptr(outs["max"]), self._s()))
        return {k: HArray(dev=v) for k, v in outs.items() if v is not None}

    def row_reduce_wide(self, data, offsets, n_rows, want=("sum",)):
        """{name: HArray} for name in want ⊆ {sum, min, max} over ragged int64 / float64 rows (bnpk_row_reduce_wide)"""
        dtype = np.dtype(data.dtype)
        assert dtype in (np.dtype(np.int64), np.dtype(np.float64))
        outs = {name: self._empty(n_rows, dtype) if name in want else None for name in ("sum", "min", "max")}
        self._chk(lib.bnpk_row_reduce_wide(self.ctx, ptr(data.dev()), 1 if dtype == np.float64 else 0, ptr(offsets.dev()), n_rows,
                                           ptr(outs["sum"]), ptr(outs["min"]), ptr(outs["max"]), self._s()))
        return {k: HArray(dev=v) for k, v in outs.items() if v is not None}

    # -- per-row values that stay in HBM (device_vector.py) ---------------------------------------------------------
    def vec_ratio_rows(self, sums, offsets, n):
        """sums[i] / (offsets[i+1] - offsets[i]) as float64 — np.mean(ragged, axis=-1)"""
        out = self._empty(n, np.float64)
        self._chk(lib.bnpk_vec_ratio_rows(self.ctx, ptr(sums.dev()), ptr(offsets.dev()), n, ptr(out), self._s()))
        return HArray(dev=out)

    def vec_compare(self, x, op, scalar):
        """x OP scalar as a 0/1 uint8 mask; x float64 / int64 / uint8, op one of < <= > >= == !="""
        code = {"<": 0, "<=": 1, ">": 2, ">=": 3, "==": 4, "!=": 5}[op]
        dtype = {np.dtype(np.float64): 0, np.dtype(np.int64): 1, np.dtype(np.uint8): 2}[np.dtype(x.dtype)]
        out = self._empty(x.size, np.uint8)
        self._chk(lib.bnpk_vec_compare(self.ctx, ptr(x.dev()), x.size, dtype, code, float(scalar) if dtype == 0 else 0.0,
                                       int(scalar) if dtype != 0 else 0, ptr(out), self._s()))
        return HArray(dev=out)

    def mask_logic(self, a, b, op):
        """a AND / OR / XOR b, or NOT a (b None): 0/1 uint8 masks"""
        code = {"and": 0, "or": 1, "xor": 2, "not": 3}[op]
        out = self._empty(a.size, np.uint8)
        self._chk(lib.bnpk_mask_logic(self.ctx, ptr(a.dev()), ptr(b.dev()) if b is not None else None, a.size, code,
                                      ptr(out), self._s()))
        return HArray(dev=out)

    def mask_fill(self, mask, start, step, count, value):
        """mask[start + i * step] = value for i < count, in place"""
        self._chk(lib.bnpk_mask_fill(self.ctx, ptr(mask.dev()), mask.size, start, step, count, 1 if value else 0, self._s()))

    def mask_rows(self, mask):
        """(indices of the set bytes of a 0/1 mask as int64, ascending; their number) — np.flatnonzero(mask)"""
        d = mask.dev()
        n = mask.size
        tiles = lib.bnpk_scan_tiles(n)
        tile_off = self._empty(tiles + 1, np.int64)
        self._chk(lib.bnpk_byte_census(self.ctx, ptr(d), n, 1, ptr(tile_off), self._s()))
        total = int(tile_off[tiles].item())
        pos = self._empty(total, np.int64)
        self._chk(lib.bnpk_byte_positions(self.ctx, ptr(d), n, 1, ptr(tile_off), total, ptr(pos), self._s()))
        return HArray(dev=pos), total

    def take_i64(self, arr, idx):
        """arr[idx] for device int64 arrays"""
        out = self._empty(idx.size, np.int64)
        self._chk(lib.bnpk_take_i64(self.ctx, ptr(arr.dev()), ptr(idx.dev()), idx.size, ptr(out), self._s()))
        return HArray(dev=out)

    def dense_to_sparse(self, hist):
        """(indices of the non-zero bins, their counts): a dense histogram in the sorted (key, count) form"""
        keys, _ = self.mask_rows(self.vec_compare(hist, "!=", 0))
        return keys, self.take_i64(hist, keys)

    def slice_copy(self, x, start, stop):
        """x[start:stop] as its own buffer"""
        return HArray(dev=x.dev()[start:stop].clone())

    # -- reverse complement / canonical k-mers ---------------------------------------------------
    def reverse_complement_packed(self, packed, offsets, n_rows, total):
        out = self._empty(total // 32 + 2, np.int64)
        self._chk(lib.bnpk_reverse_complement_packed(self.ctx, ptr(packed.dev()), ptr(offsets.dev()), n_rows, total,
                                                     ptr(out), self._s()))
        return HArray(dev=out)

    def reverse_complement_bytes(self, flat, offsets, n_rows, total):
        out = self._empty(total, np.uint8)
        self._chk(lib.bnpk_reverse_complement_bytes(self.ctx, ptr(flat.dev()), ptr(offsets.dev()), n_rows, total,
                                                    ptr(out), self._s()))
        return HArray(dev=out)

    def reverse_complement_rows(self, buf, starts, offsets, n_rows, total):
        """reverse_complement_bytes over rows buf[starts[r] .. + offsets[r+1] - offsets[r]) that nobody gathered
        (bnpk_reverse_complement_rows); the result is compact"""
        out = self._empty(total, np.uint8)
        self._chk(lib.bnpk_reverse_complement_rows(self.ctx, ptr(buf.dev()), buf.size, ptr(starts.dev()), ptr(offsets.dev()), n_rows,
                                                   total, ptr(out), self._s()))
        return HArray(dev=out)

    def canonical_kmers(self, hashes, k):
        """h = min(h, hash of the reverse complement k-mer); overwrites the device buffer of ``hashes``"""
        t = hashes.dev()
        self._chk(lib.bnpk_canonical_kmers(self.ctx, ptr(t), t.numel(), k, self._s()))
        return HArray(dev=t)

    def kmers_by_rows(self, packed, in_offsets, out_offsets, n_rows, n_out, k):
        """bnpk_kmers: the output-flat kernel with per-lane row lookups (kept for comparison / as the reference form)"""
        out = self._empty(n_out, np.int64)
        self._chk(lib.bnpk_kmers(self.ctx, ptr(packed.dev()), ptr(in_offsets.dev()), ptr(out_offsets.dev()), n_rows,
                                 n_out, k, ptr(out), self._s()))
        return HArray(dev=out)

    def minimizers_by_rows(self, packed, in_offsets, out_offsets, n_rows, n_out, k, window_size):
        out = self._empty(n_out, np.int64)
        self._chk(lib.bnpk_minimizers(self.ctx, ptr(packed.dev()), ptr(in_offsets.dev()), ptr(out_offsets.dev()),
                                      n_rows, n_out, k, window_size, ptr(out), self._s()))
        return HArray(dev=out)

    # -- A9 ---------------------------------------------------------------------------------------------------
    def count_dense(self, values, n_bins, hist=None):
        if hist is None:
            hist = HArray(dev=self.device.zeros(n_bins, np.int64))
        self._chk(lib.bnpk_count_dense(self.ctx, ptr(values.dev()), values.size, n_bins, ptr(hist.dev()), self._s()))
        return hist

    def count_bytes(self, values, n_bins, hist=None):
        """np.bincount(values, minlength=n_bins)[:n_bins] over uint8 codes where they lie (bnpk_count_bytes)"""
        if hist is None:
            hist = HArray(dev=self.device.zeros(n_bins, np.int64))
        self._chk(lib.bnpk_count_bytes(self.ctx, ptr(values.dev()), values.size, n_bins, ptr(hist.dev()), self._s()))
        return hist

    def count_packed(self, packed, n_bases, hist=None):
        """the same over DNA packed 2 bits per base: four bins (bnpk_count_packed2)"""
        if hist is None:
            hist = HArray(dev=self.device.zeros(4, np.int64))
        self._chk(lib.bnpk_count_packed2(self.ctx, ptr(packed.dev()), n_bases, ptr(hist.dev()), self._s()))
        return hist

    COUNT_BYTES_ROWS_MAX_BINS = 8

    def count_bytes_rows(self, values, offsets, n_rows, total, n_bins):
        """one histogram per row of ragged uint8 codes, [n_rows * n_bins] int64 (bnpk_count_bytes_rows; n_bins <= 8)"""
        hist = self._empty(n_rows * n_bins, np.int64)
        self._chk(lib.bnpk_count_bytes_rows(self.ctx, ptr(values.dev()), ptr(offsets.dev()), n_rows, total, n_bins, ptr(hist),
                                            self._s()))
        return HArray(dev=hist)

    def count_dense_rows(self, values, offsets, n_rows, n_bins):
        hist = self.device.zeros(n_rows * n_bins, np.int64)
        self._chk(lib.bnpk_count_dense_rows(self.ctx, ptr(values.dev()), ptr(offsets.dev()), n_rows, values.size,
                                            n_bins, ptr(hist), self._s()))
        return HArray(dev=hist)

    def count_weighted(self, values, weights, n, n_rows, value_stride, weight_stride, n_bins):
        """bnpk_count_weighted: hist[r, values[r * value_stride + i]] += weights[r * weight_stride + i].  weights: an HArray
        of int64 (exact) or float64; returns the [n_rows, n_bins] histogram of the same dtype."""
        f64 = weights.dtype == np.float64
        assert f64 or weights.dtype == np.int64
        hist = self.device.zeros(n_rows * n_bins, np.float64 if f64 else np.int64)
        bad = C.c_int(0)
        self._chk(lib.bnpk_count_weighted(self.ctx, ptr(values.dev()), ptr(weights.dev()), 1 if f64 else 0, n, n_rows,
                                          value_stride, weight_stride, n_bins, ptr(hist), C.byref(bad), self._s()))
        if bad.value:
            raise ValueError("count_encoded: a value outside the alphabet")
        return HArray(dev=hist)

    def sort_keys(self, keys_t, key_bits, begin_bit=0):
        """sorts a torch int64 tensor on bits [begin_bit, key_bits); returns the tensor holding the result
        and the other (free) one"""
        n = keys_t.numel()
        alt = self._empty(n, np.int64)
        in_alt = C.c_int(0)
        self._chk(lib.bnpk_sort_keys(self.ctx, ptr(keys_t), ptr(alt), n, begin_bit, key_bits, C.byref(in_alt),
                                     self._s()))
        return (alt, keys_t) if in_alt.value else (keys_t, alt)

    def partition_by_top_bits(self, values, key_bits, top_bits):
        """partition of the keys by their top ``top_bits`` bits (of ``key_bits``), one MSD radix level.
        Returns the partitioned keys and the bucket boundaries (2^top_bits + 1 offsets, host numpy)."""
        out, child = self.radix_partition(values.dev(), None, 1, key_bits - top_bits, top_bits)
        return HArray(dev=out), child.cpu().numpy()

    def _runs(self, sorted_t, second_t=None):
        """run boundaries of a sorted tensor (optionally of (sorted, second) pairs): n_runs, tile offsets"""
        n = sorted_t.numel()
        tiles = lib.bnpk_run_tiles(n)
        tile_off = self._empty(tiles + 1, np.int64)
        n_runs = C.c_int64(0)
        self._chk(lib.bnpk_run_census(self.ctx, ptr(sorted_t), ptr(second_t), n, ptr(tile_off), C.byref(n_runs),
                                      self._s()))
        return int(n_runs.value), tile_off

    # -- sparse histogram: MSD radix partition (write-combining scatter) + in-LDS finishing sort -------------
    FINISH_TARGET = 7000          # average bucket size the plan aims for (csrc/sparse.hip FINISH_TARGET: the same number, and why)

    @classmethod
    def radix_plan(cls, n, key_bits, done=0):
        """digit widths of the MSD levels still to run so that the buckets average <= FINISH_TARGET keys,
        given that the top ``done`` bits are already resolved"""
        need = 0
        while need < key_bits and (n >> need) > cls.FINISH_TARGET:
            need += 1
        rest = max(0, need - done)
        if rest == 0:
            return []
        max_bits = 10                                    # bnpk_radix_max_bits() is 11; 10-bit digits flush whole 128-B lines
        if -(-rest // 11) < -(-rest // 10):              # ... but an 11-bit digit is cheaper than one more level
            max_bits = 11
        levels = -(-rest // max_bits)
        base, extra = divmod(rest, levels)
        return [base + (1 if i < extra else 0) for i in range(levels)]

    def kmer_start_mask(self, offsets, n_rows, total, k):
        """one bit per base, set where a k-mer starts (bnpk_kmer_start_mask)"""
        # the rows' end bits (one atomic per row), then the bit-parallel windowed OR the fused decode uses: a third of the
        # time of bnpk_kmer_start_mask, which walks the words of every row
        mask = self._empty(total // 64 + 2, np.int64)
        if k > 64:                                       # (windows longer than the windowed OR looks ahead)
            self._chk(lib.bnpk_kmer_start_mask(self.ctx, ptr(offsets.dev()), n_rows, total, k, ptr(mask), self._s()))
            return HArray(dev=mask)
        ends = self._empty(total // 64 + 2, np.int64)
        count = self._empty(1, np.int64)
        self._chk(lib.bnpk_row_end_mask(self.ctx, ptr(offsets.dev()), n_rows, total, ptr(ends), self._s()))
        self._chk(lib.bnpk_kmer_starts_from_ends(self.ctx, ptr(ends), total, k, ptr(mask), ptr(count), self._s()))
        return HArray(dev=mask)

    def kmer_start_mask_by_rows(self, offsets, n_rows, total, k):
        """bnpk_kmer_start_mask itself (a thread walks the words of its row): any k, kept as the reference form"""
        mask = self._empty(total // 64 + 2, np.int64)
        self._chk(lib.bnpk_kmer_start_mask(self.ctx, ptr(offsets.dev()), n_rows, total, k, ptr(mask), self._s()))
        return HArray(dev=mask)

    def kmers_partitioned(self, packed, starts_mask, n_bases, n_out, k, bits, canonical=False):
        """bnpk_kmers_partition: the k-mer hashes written once, partitioned by their top ``bits`` bits.
        Returns (hashes, bucket offsets[2^bits + 1])"""
        out = self._empty(n_out, np.int64)
        child = self._empty((1 << bits) + 1, np.int64)
        self._chk(lib.bnpk_kmers_partition(self.ctx, ptr(packed.dev()), ptr(starts_mask.dev()), n_bases, k,
                                           1 if canonical else 0, 2 * k - bits,
                                           bits, ptr(out), ptr(child), self._s()))
        return HArray(dev=out), HArray(dev=child)

    def radix_partition(self, keys_t, seg_offsets_t, n_seg, shift, bits, out_t=None):
        """one MSD level over torch tensors: returns (partitioned keys, child offsets[n_seg * 2^bits + 1])"""
        n = keys_t.numel()
        out = out_t if out_t is not None else self._empty(n, np.int64)
        child = self._empty(n_seg * (1 << bits) + 1, np.int64)
        self._chk(lib.bnpk_radix_partition(self.ctx, ptr(keys_t), n, ptr(seg_offsets_t), n_seg, shift, bits, ptr(out),
                                           ptr(child), self._s()))
        return out, child

    def count_sparse(self, values, key_bits=62, consume=False, partition=None, key_range=None, fast=True, skew=1.0, dest=None):
        """np.unique(values, return_counts=True) on the device -> (keys, counts) HArrays (sorted keys): ONE call of
        bnpk_count_sparse (csrc/sparse.hip plans the levels, the claiming level, the census, the heavy buckets and the
        fall-back; rounds 1-5 had that planner here).

        partition: (bucket_offsets, bits) if ``values`` is already grouped by its top ``bits`` bits
        (kmers_partitioned).  key_range: (lo, hi) if all values are known to lie in [lo, hi) (the key range a
        rank owns after the multi-GPU exchange) — the shared leading bits are then skipped by the partition.
        fast=False forces the fallback (rocPRIM sort + run kernels) that heavy-hitter buckets take.
        skew: densest / average density of the keys over their range, where it is known (canonical k-mers: 2) —
        the levels are then planned for the densest part instead of discovering it from over-full buckets.
        dest: (keys HArray, counts HArray, pos) — the result is written to [pos, pos + distinct) of these device arrays,
        which have room for pos + len(values) entries, and views of them are returned (the pieces of a histogram that is
        counted key range by key range land next to each other without a copy)."""
        t = values.dev()
        n = t.numel()
        if dest is not None:
            out_keys, out_counts, pos = dest[0].dev(), dest[1].dev(), int(dest[2])
            assert pos + n <= out_keys.numel() and pos + n <= out_counts.numel()
        if n == 0:
            if dest is not None:
                return HArray(dev=out_keys[pos:pos]), HArray(dev=out_counts[pos:pos])
            z = self._empty(0, np.int64)
            return HArray(dev=z), HArray(dev=z.clone())
        work = t if consume else t.clone()                   # (the call uses its input as workspace: never the caller's array)
        if not fast or key_bits > 62:
            keys_out, counts = self._count_by_sorting(work, key_bits)
            if dest is not None:
                d = keys_out.numel()
                out_keys[pos:pos + d].copy_(keys_out)
                out_counts[pos:pos + d].copy_(counts)
                keys_out, counts = out_keys[pos:pos + d], out_counts[pos:pos + d]
            return HArray(dev=keys_out), HArray(dev=counts)
        skip, n_plan = 0, int(n * skew)
        if key_range is not None and partition is None:
            lo, hi = int(key_range[0]), int(key_range[1])
            skip = key_bits - (lo ^ (hi - 1)).bit_length()
            n_plan = int(n * skew * (1 << (key_bits - skip)) / max(hi - lo, 1))
        offsets, done = (partition[0].dev(), int(partition[1])) if partition is not None else (None, 0)
        if dest is not None:
            keys_out, counts = out_keys[pos:pos + n], out_counts[pos:pos + n]
        else:
            keys_out, counts = self._empty(n, np.int64), self._empty(n, np.int64)
        # the workspace: enough for any input where the device has it to spare (heavy-hitter buckets, the library sort: ~6 n
        # words), else what the claiming level takes (~1.4 n: the 31-mer batch that fills the HBM), else the plain levels'
        tm = torch_mod()
        free, _ = tm.cuda.mem_get_info(self.device.tdev)
        room = free + tm.cuda.memory_reserved(self.device.tdev) - tm.cuda.memory_allocated(self.device.tdev)
        sizes = [int(lib.bnpk_count_sparse_workspace(n, key_bits, skip, n_plan, done, mode)) for mode in (0, 1, 2)]
        if not self.claim_last_level:
            sizes[1] = sizes[0]
        lib.bnpk_set_option(self.ctx, b"sparse_claim", 1 if self.claim_last_level else 0)
        work_bytes = sizes[2] if sizes[2] < room // 2 else (sizes[1] if sizes[1] + (64 << 20) < room else sizes[0])
        space = self._empty(work_bytes, np.uint8)
        n_unique = C.c_int64(0)
        info = (C.c_int64 * 5)()
        status = lib.bnpk_count_sparse(self.ctx, ptr(work), n, key_bits, skip, n_plan, ptr(offsets), done, ptr(space), work_bytes,
                                       ptr(keys_out), ptr(counts), C.byref(n_unique), info, self._s())
        if status == -4 and work_bytes < sizes[2]:
            # BNPK_ERR_NOMEM: the keys needed a path the workspace had no room for (heavy-hitter buckets on an input too large to
            # be given the any-input workspace up front).  The call consumed its input; where that was a copy, the caller's keys
            # are intact and the call is repeated with everything the allocator can give back
            del space
            tm.cuda.empty_cache()
            if consume:
                raise MemoryError("bnpk_count_sparse: %d keys with over-full buckets need a workspace of %.1f GB (the call was given "
                                  "%.1f GB and has used its input up); count a copy (consume=False) or smaller batches"
                                  % (n, sizes[2] / 1e9, work_bytes / 1e9))
            work = t.clone()
            work_bytes = sizes[2]
            space = self._empty(work_bytes, np.uint8)
            status = lib.bnpk_count_sparse(self.ctx, ptr(work), n, key_bits, skip, n_plan, ptr(offsets), done, ptr(space), work_bytes,
                                           ptr(keys_out), ptr(counts), C.byref(n_unique), info, self._s())
        self._chk(status)
        self.last_sparse_info = {"path": int(info[0]), "levels": int(info[1]), "round_trips": int(info[2]), "bag": int(info[3]),
                                 "precounted": int(info[4]), "workspace": work_bytes}
        if info[0] == 1 or info[3] > 0:                      # the claiming level ran (and, if the path is not 1, was given up: its bag overflowed)
            self.last_claimed = {"n": n, "bag": int(info[3]), "kept": info[0] == 1}
        d = n_unique.value
        return HArray(dev=keys_out[:d]), HArray(dev=counts[:d])

    claim_last_level = True       # may the last partition level claim its buckets' places (bnpk_count_sparse's claiming level)?
    last_claimed = None           # {"n", "bag"} if the last call of count_sparse took the claiming level (experiments, tests)
    last_sparse_info = None       # what bnpk_count_sparse reported about its last call

    def _count_by_sorting(self, work, key_bits):
        """(sorted distinct keys, counts) of a torch int64 tensor (consumed): rocPRIM radix sort + run kernels"""
        n = work.numel()
        sorted_t, free_t = self.sort_keys(work, key_bits)
        n_runs, tile_off = self._runs(sorted_t)
        keys_out = free_t[:n_runs]                       # the ping-pong buffer is free after the sort
        starts = self._empty(n_runs + 1, np.int64)
        self._chk(lib.bnpk_run_heads(self.ctx, ptr(sorted_t), None, n, ptr(tile_off), n_runs, ptr(keys_out), None,
                                     ptr(starts), self._s()))
        counts = self._empty(n_runs, np.int64)
        self._chk(lib.bnpk_run_sums(self.ctx, ptr(starts), n_runs, None, ptr(counts), self._s()))
        return keys_out, counts

    def merge_add(self, a_keys, a_counts, b_keys, b_counts):
        """(keys, counts) of the sum of two sparse histograms (sorted distinct keys each): bnpk_merge_add"""
        na, nb = a_keys.size, b_keys.size
        out_k, out_c = self._empty(na + nb, np.int64), self._empty(na + nb, np.int64)
        n_out = C.c_int64(0)
        self._chk(lib.bnpk_merge_add(self.ctx, ptr(a_keys.dev()) if na else None, ptr(a_counts.dev()) if na else None, na,
                                     ptr(b_keys.dev()) if nb else None, ptr(b_counts.dev()) if nb else None, nb,
                                     ptr(out_k), ptr(out_c), C.byref(n_out), self._s()))
        return HArray(dev=out_k[:n_out.value]), HArray(dev=out_c[:n_out.value])

    def reduce_by_key(self, keys, weights, key_bits=62):
        """sum of weights per distinct key (merge of sparse histograms; EncodedCounts.__add__ analogue)."""
        t = self.device.torch_cat([k.dev() for k in keys]) if isinstance(keys, (list, tuple)) else keys.dev().clone()
        w = self.device.torch_cat([x.dev() for x in weights]) if isinstance(weights, (list, tuple)) \
            else weights.dev().clone()
        n = t.numel()
        if n == 0:
            return HArray(dev=t), HArray(dev=w)
        t_alt = self._empty(n, np.int64)
        w_alt = self._empty(n, np.int64)
        in_alt = C.c_int(0)
        self._chk(lib.bnpk_sort_pairs(self.ctx, ptr(t), ptr(t_alt), ptr(w), ptr(w_alt), n, key_bits, C.byref(in_alt),
                                      self._s()))
        if in_alt.value:
            t, t_alt, w, w_alt = t_alt, t, w_alt, w
        n_runs, tile_off = self._runs(t)
        keys_out = t_alt[:n_runs]
        starts = self._empty(n_runs + 1, np.int64)
        self._chk(lib.bnpk_run_heads(self.ctx, ptr(t), None, n, ptr(tile_off), n_runs, ptr(keys_out), None,
                                     ptr(starts), self._s()))
        prefix = self._empty(n + 1, np.int64)
        self._chk(lib.bnpk_exclusive_scan_i64(self.ctx, ptr(w), n, ptr(prefix), self._s()))
        sums = self._empty(n_runs, np.int64)
        self._chk(lib.bnpk_run_sums(self.ctx, ptr(starts), n_runs, ptr(prefix), ptr(sums), self._s()))
        return HArray(dev=keys_out), HArray(dev=sums)

    # -- A12 -------------------------------------------------------------------------------------------------------
    def row_ids(self, offsets, n_rows, n):
        rows = self._empty(n, np.int64)
        self._chk(lib.bnpk_row_ids(self.ctx, ptr(offsets.dev()), n_rows, n, ptr(rows), self._s()))
        return HArray(dev=rows)

    def unique_pairs(self, keys, values, key_bits=62, n_values=None, with_counts=False):
        """sorted distinct (key, value) pairs, values in [0, n_values) — KmerIndex.create_index
        (bionumpy/sequence/indexing/kmer_indexing.py:24-47) as ONE call of bnpk_index_build (csrc/sparse.hip): up to 1024
        values ONE partition of (key's low bits : value : tag) words behind a first level over the key's top bits; more values
        (or option "index_pairs" 0, or words that only a sort of everything could count) the distinct values of
        id = rank(key) * n_values + value — two runs of the sparse counting path around a rank kernel.  No library sort on
        either way (round 5 sorted (k-mer, row) with rocprim's radix_sort_pairs for indices under 2^26 pairs)."""
        t, v = keys.dev(), values.dev()
        n = t.numel()
        if n == 0:
            empty = (HArray(dev=t.clone()), HArray(dev=v.clone()))
            return empty + (HArray(dev=t.clone()),) if with_counts else empty
        if n_values is None:
            n_values = int(v.max().item()) + 1
        work_bytes = int(lib.bnpk_index_build_workspace(n, key_bits, n_values))
        keys_out, vals_out = self._empty(n, np.int64), self._empty(n, np.int64)
        counts = self._empty(n, np.int64) if with_counts else None
        m = C.c_int64(0)
        for attempt in range(2):                             # (the inputs are left alone: a workspace that was too small is asked for again, larger)
            space = self._empty(work_bytes, np.uint8)
            status = lib.bnpk_index_build(self.ctx, ptr(t), ptr(v), n, key_bits, n_values, ptr(space), work_bytes, ptr(keys_out),
                                          ptr(vals_out), ptr(counts), C.byref(m), self._s())
            del space
            if status != -4:                                 # BNPK_ERR_NOMEM
                break
            work_bytes *= 3
        if status == -6:                                     # BNPK_ERR_RANGE: distinct keys x values over 62 bits
            raise NotImplementedError("index too large: %d pairs x %d rows" % (n, n_values))
        self._chk(status)
        d = m.value
        if with_counts:
            return HArray(dev=keys_out[:d]), HArray(dev=vals_out[:d]), HArray(dev=counts[:d])
        return HArray(dev=keys_out[:d]), HArray(dev=vals_out[:d])

    def search_sorted(self, sorted_keys, queries, upper=False):
        m = queries.size
        out = self._empty(m, np.int64)
        self._chk(lib.bnpk_search_sorted(self.ctx, ptr(sorted_keys.dev()), sorted_keys.size, ptr(queries.dev()), m,
                                         1 if upper else 0, ptr(out), self._s()))
        return HArray(dev=out)

    # -- misc --------------------------------------------------------------------------------------------------------
    def empty_i64(self, n):
        """uninitialised int64 array in HBM"""
        return HArray(dev=self._empty(n, np.int64))

    def concat(self, arrays):
        return HArray(dev=self.device.torch_cat([a.dev() for a in arrays]))

    def concat_words(self, parts, pad=2):
        """the first n words of every (int64 HArray, n) one behind the other, plus ``pad`` zero words — the packed reads /
        k-mer start masks of several chunks as one array (the pieces are cut at whole words: SparseKmerCounts._count_reads)"""
        pieces = [a.dev()[:n] for a, n in parts]
        return HArray(dev=self.device.torch_cat(pieces + [self.device.zeros(pad, np.int64)]))

    def add_i64(self, a, b):
        return HArray(dev=a.dev() + b.dev())

    def synth_fastq(self, n_reads, read_len, seed, mode=0, genome_len=0, first_read=0):
        total = n_reads * lib.bnpk_synth_record_bytes(read_len)
        out = self._empty(total, np.uint8)
        self._chk(lib.bnpk_synth_fastq(self.ctx, ptr(out), first_read, n_reads, read_len, seed, mode, genome_len,
                                       self._s()))
        return HArray(dev=out)


_ops = None


def get_ops():
    global _ops
    if _ops is None:
        _ops = HipOps()
    return _ops


def set_ops(ops):
    """Replace the ops object (tests only: CPU-side host-logic tests inject an oracle-backed stand-in)."""
    global _ops
    _ops = ops
