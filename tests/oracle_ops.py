"""TEST-ONLY stand-in for bionumpy_amd.ops.HipOps backed by the CPU oracle.

Lets the CPU-only suite (-m "not gpu") drive the *host logic* of bionumpy_amd (chunk loop, lazy
chunk objects, ragged/encoded array classes, exception mapping, API surface) without a GPU.  It is
never importable from the product package and never used on the GPU box: the -m gpu tests run the
same API tests through the real HIP ops and compare against the oracle's outputs.
"""
from collections import namedtuple

import numpy as np

import oracle
from oracle import text as otext
from bionumpy_amd.device import HArray
from bionumpy_amd.exceptions import FormatException, IncompleteEntryException, EncodingError

LineScan = namedtuple("LineScan", "size n_lines n_records newlines has_cr")
NEWLINE = 10


def _h(a):
    return HArray(host=np.ascontiguousarray(a))


def _unpack(packed, n):
    words = packed.host().view(np.uint64)
    i = np.arange(n, dtype=np.int64)
    return ((words[i >> 5] >> (2 * (i & 31)).astype(np.uint64)) & np.uint64(3)).astype(np.uint8)


def _pack(codes):
    n = codes.size
    out = np.zeros(n // 32 + 2, dtype=np.uint64)
    w = oracle.pack_2bit(codes)
    out[:w.size] = w
    return _h(out.view(np.int64))


class OracleOps:
    host_only = True          # parallel.py then moves numpy-backed tensors (gloo) instead of HBM tensors

    def partition_by_top_bits(self, values, key_bits, top_bits):
        v = values.host()
        bucket = v >> (key_bits - top_bits)
        order = np.argsort(bucket, kind="stable")
        cuts = np.searchsorted(bucket[order], np.arange((1 << top_bits) + 1))
        return _h(v[order]), cuts.astype(np.int64)

    @staticmethod
    def radix_plan(n, key_bits, done=0):
        from bionumpy_amd.ops import HipOps
        return HipOps.radix_plan(n, key_bits, done)

    def fastq_encode(self, buf, n, lines_per_entry, seq_line, header, check_plus):
        scan = self.scan_lines(buf, n, lines_per_entry, header, check_plus)
        starts, lens = self.field_table(buf, scan.newlines, scan.n_records, lines_per_entry, seq_line, 0, scan.has_cr)
        offsets, n_bases = self.row_offsets(lens, 1)
        _, packed = self.gather_encode_dna(buf, starts, offsets, scan.n_records, n_bases)
        off = offsets.host()
        bits = np.zeros((n_bases // 64 + 2) * 64, dtype=np.uint8)
        bits[off[1:][np.diff(off) > 0] - 1] = 1
        return packed, _h(np.packbits(bits, bitorder="little").view(np.int64)), scan.n_records, n_bases

    def kmer_starts_from_ends(self, row_ends, n_bases, k):
        flags = np.unpackbits(row_ends.host().view(np.uint8), bitorder="little")
        ends = np.flatnonzero(flags[:n_bases]) + 1
        out = np.zeros(flags.size, dtype=np.uint8)
        for s0, e in zip(np.concatenate([[0], ends[:-1]]), ends):
            if e - (k - 1) > s0:
                out[s0:e - (k - 1)] = 1
        return _h(np.packbits(out, bitorder="little").view(np.int64)), int(out.sum())

    def kmer_start_mask(self, offsets, n_rows, total, k):
        off = offsets.host()
        bits = np.zeros((total // 64 + 2) * 64, dtype=np.uint8)
        for s, e in zip(off[:-1], off[1:]):
            if e - (k - 1) > s:
                bits[s:e - (k - 1)] = 1
        return _h(np.packbits(bits, bitorder="little").view(np.int64))

    def kmers_partitioned(self, packed, starts_mask, n_bases, n_out, k, bits, canonical=False):
        flags = np.unpackbits(starts_mask.host().view(np.uint8), bitorder="little")[:n_bases]
        pos = np.flatnonzero(flags)
        codes = _unpack(packed, n_bases).astype(np.int64)
        h = np.zeros(pos.size, dtype=np.int64)
        for j in range(k):
            h |= codes[pos + j] << (2 * j)
        assert h.size == n_out
        if canonical:
            h = oracle.canonical_kmers(h, k)
        digit = h >> (2 * k - bits)
        order = np.argsort(digit, kind="stable")
        cuts = np.searchsorted(digit[order], np.arange((1 << bits) + 1)).astype(np.int64)
        return _h(h[order]), _h(cuts)

    # -- decode -----------------------------------------------------------------------------------
    def newline_positions(self, buf, n, limit_multiple=1):
        pos = np.flatnonzero(buf.host()[:n] == NEWLINE).astype(np.int64)
        total = pos.size
        return _h(pos[:total - total % limit_multiple]), total

    def scan_lines(self, buf, n, lines_per_entry, header, check_plus):
        fmt = otext.LineFormat(header, lines_per_entry, (0,) * lines_per_entry, check_plus)
        try:
            res = oracle.scan_one_line_buffer(buf.host()[:n], fmt)
        except otext.IncompleteEntryException as e:
            raise IncompleteEntryException(str(e))
        except otext.FormatException as e:
            raise FormatException(str(e), line_number=e.line_number)
        data = buf.host()
        nl = res.new_lines
        ends = nl.reshape(-1, lines_per_entry)
        has_cr = bool(ends[0, 0] >= 1 and np.any(data[ends[:lines_per_entry, 0] - 1] == 13))
        return LineScan(res.size, res.n_lines, res.n_records, _h(nl.astype(np.int64)), has_cr)

    def field_table(self, buf, newlines, n_entries, lines_per_entry, field, line_offset, strip_cr):
        nl = newlines.host()
        data = buf.host()
        line_starts = np.concatenate(([0], nl + 1))[:-1].reshape(-1, lines_per_entry)[:, field] + line_offset
        ends = nl.reshape(-1, lines_per_entry)[:, field]
        if strip_cr:
            ends = ends - ((ends >= 1) & (data[np.maximum(ends - 1, 0)] == 13))
        return _h(line_starts.astype(np.int64)), _h((ends - line_starts).astype(np.int64))

    def entry_table(self, newlines, lines_per_entry, rows):
        nl, r = newlines.host(), rows.host()
        first = r * lines_per_entry
        starts = np.where(first > 0, nl[np.maximum(first - 1, 0)] + 1, 0)
        ends = nl[first + lines_per_entry - 1] + 1
        return _h(starts.astype(np.int64)), _h((ends - starts).astype(np.int64))

    def take_bytes(self, buf, positions, delta):
        return _h(buf.host()[positions.host() + delta])

    def read_i64(self, arr, indices):
        return arr.host()[np.asarray(indices, dtype=np.int64)]

    # -- multi-line FASTA ----------------------------------------------------------------------------
    def multiline_cut(self, buf, newlines, marker):
        hits = np.flatnonzero(buf.host()[newlines.host() + 1] == marker)
        return (int(hits[-1]) if hits.size else -1), int(hits.size)

    def multiline_table(self, buf, size, newlines, n_newlines, marker, strip_cr):
        data = buf.host()[:size]
        nl = newlines.host()[:n_newlines]
        line_starts = np.concatenate(([0], nl + 1))
        line_ends = np.concatenate((nl, [size - 1]))
        if strip_cr:
            line_ends = line_ends - (data[line_ends - 1] == 13)
        is_header = data[line_starts] == marker
        is_header[0] = True
        line_lens = line_ends - line_starts
        rec_of_line = np.cumsum(is_header) - 1
        seq = ~is_header
        rec_lens = np.bincount(rec_of_line[seq], weights=line_lens[seq], minlength=int(is_header.sum())).astype(np.int64)
        return (_h(line_starts[is_header] + 1), _h(line_lens[is_header] - 1), _h(rec_lens), _h(line_starts[seq]),
                _h(line_lens[seq]), int(line_lens[seq].sum()))

    def multiline_wrap(self, names, name_offsets, seq, seq_offsets, n_records, width, marker):
        no, so = name_offsets.host(), seq_offsets.host()
        return _h(oracle.multiline_from_data(names.host()[:int(no[-1])], np.diff(no), seq.host()[:int(so[-1])],
                                             np.diff(so), width, marker))

    # -- offsets -----------------------------------------------------------------------------------
    def row_offsets(self, lens, window=1):
        l = lens.host().astype(np.int64)
        if window > 1:
            l = np.maximum(l - (window - 1), 0)
        off = np.concatenate(([0], np.cumsum(l))).astype(np.int64)
        return _h(off), int(off[-1])

    def exclusive_scan(self, values):
        return _h(np.concatenate(([0], np.cumsum(values.host()))).astype(np.int64))

    # -- encode ------------------------------------------------------------------------------------
    def _encode(self, ascii_):
        try:
            return oracle.encode_dna(ascii_)
        except otext.EncodingError as e:
            raise EncodingError(e.message, e.offset)

    def gather_encode_dna(self, buf, starts, offsets, n_rows, total, want_codes=False, want_packed=True):
        lens = np.diff(offsets.host())
        codes = self._encode(oracle.gather_rows(buf.host(), starts.host(), lens))
        return (_h(codes) if want_codes else None, _pack(codes) if want_packed else None)

    def gather_rows(self, buf, starts, offsets, n_rows, total, subtract=0):
        lens = np.diff(offsets.host())
        out = oracle.gather_rows(buf.host(), starts.host(), lens)
        return _h((out.astype(np.int64) - subtract).astype(np.uint8))

    def encode_dna_flat(self, ascii_bytes, want_codes=True, want_packed=True):
        codes = self._encode(ascii_bytes.host())
        return (_h(codes) if want_codes else None, _pack(codes) if want_packed else None)

    def pack_codes(self, codes):
        return _pack(codes.host())

    def unpack_codes(self, packed, n, to_ascii=False):
        codes = _unpack(packed, n)
        return _h(oracle.decode_dna(codes) if to_ascii else codes)

    # -- k-mers ------------------------------------------------------------------------------------
    def match_windows(self, data, offsets, n_rows, total, n_out, pattern, packed):
        flat = _unpack(data, total) if packed else data.host()[:total]
        hit, _ = oracle.match_string(flat, np.diff(offsets.host()), np.asarray(list(pattern), dtype=np.uint8))
        assert hit.size == n_out
        return _h(hit)

    def match_rows(self, packed, offsets, n_rows, total, pattern):
        lens = np.diff(offsets.host())
        hit, out_lens = oracle.match_string(_unpack(packed, total), lens, np.asarray(list(pattern), dtype=np.uint8))
        ends = np.cumsum(out_lens)
        sums = np.concatenate([[0], np.cumsum(hit.astype(np.int64))])
        return _h((sums[ends] - sums[ends - out_lens]).astype(np.int64))

    def pwm_scores(self, packed, offsets, n_rows, total, n_out, matrix):
        scores, _ = oracle.pwm_scores(_unpack(packed, total), np.diff(offsets.host()), matrix)
        assert scores.size == n_out
        return _h(scores)

    def join_lines(self, n_rows, lines, header):
        fields = []
        for line in lines:
            data, off, add, prefix, fill = line[:5]
            starts = line[5] if len(line) > 5 else None          # rows that lie at data[starts[r]], not back to back
            if data is None:
                fields.append((np.full(n_rows, fill, dtype=np.uint8), np.ones(n_rows, dtype=np.int64)))
            else:
                lens = np.diff(off.host())
                flat = data.host()[:int(off.host()[-1])] if starts is None else oracle.gather_rows(data.host(), starts.host(), lens)
                fields.append(((flat.astype(np.int64) + add).astype(np.uint8), lens))
        return _h(oracle.join_fields(fields, header, [line[3] for line in lines]))

    def col_sums_u8(self, data, offsets, n_rows, total, n_cols):
        sums, counts = oracle.col_sums(data.host()[:total], np.diff(offsets.host()))
        return _h(sums[:n_cols]), _h(counts[:n_cols])

    def row_reduce_u8(self, data, offsets, n_rows, want=("sum",)):
        sums, mins, maxs = oracle.row_reduce(data.host(), np.diff(offsets.host()))
        full = {"sum": sums, "min": mins, "max": maxs}
        return {k: _h(full[k]) for k in want}

    def row_reduce_wide(self, data, offsets, n_rows, want=("sum",)):
        flat, off = data.host(), offsets.host()
        full = {name: np.zeros(n_rows, dtype=flat.dtype) for name in ("sum", "min", "max")}
        for r in range(n_rows):
            row = flat[off[r]:off[r + 1]]
            if row.size:
                full["sum"][r], full["min"][r], full["max"][r] = row.sum(), row.min(), row.max()
        return {k: _h(full[k]) for k in want}

    def reverse_complement_packed(self, packed, offsets, n_rows, total):
        lens = np.diff(offsets.host())
        return _pack(oracle.reverse_complement(_unpack(packed, total), lens))

    def reverse_complement_bytes(self, flat, offsets, n_rows, total):
        return _h(oracle.reverse_complement(flat.host()[:total], np.diff(offsets.host()), ascii_bytes=True))

    def canonical_kmers(self, hashes, k):
        return _h(oracle.canonical_kmers(hashes.host(), k))

    def windows_from_mask(self, packed, start_mask, n_bases, n_out, k, window_size):
        flags = np.unpackbits(start_mask.host().view(np.uint8), bitorder="little")[:n_bases]
        pos = np.flatnonzero(flags)
        codes = _unpack(packed, n_bases).astype(np.int64)
        per = window_size - k + 1
        best = None
        for i in range(per):
            h = np.zeros(pos.size, dtype=np.int64)
            for j in range(k):
                h |= codes[pos + i + j] << (2 * j)
            best = h if best is None else np.minimum(best, h)
        assert pos.size == n_out
        return _h(best if best is not None else np.zeros(0, dtype=np.int64))

    def kmers(self, packed, in_offsets, out_offsets, n_rows, n_out, k, total=None):
        off = in_offsets.host()
        h, _ = oracle.get_kmers(_unpack(packed, int(off[-1])), np.diff(off), k)
        assert h.size == n_out
        return _h(h)

    def kmers_generic(self, codes, in_offsets, out_offsets, n_rows, n_out, k, alphabet_size):
        off = in_offsets.host()
        h, _ = oracle.get_kmers_generic(codes.host()[:int(off[-1])], np.diff(off), k, alphabet_size)
        assert h.size == n_out
        return _h(h)

    def minimizers_generic(self, codes, in_offsets, out_offsets, n_rows, n_out, k, window_size, alphabet_size):
        off = in_offsets.host()
        m, _ = oracle.get_minimizers(codes.host()[:int(off[-1])], np.diff(off), k, window_size, alphabet_size)
        assert m.size == n_out
        return _h(m)

    def lut_bytes(self, data, lut, what="AlphabetEncoding"):
        try:
            return _h(oracle.encode_dna(data.host(), np.asarray(lut, dtype=np.uint8), 255))
        except otext.EncodingError as e:
            raise EncodingError(e.message, e.offset)

    def minimizers(self, packed, in_offsets, out_offsets, n_rows, n_out, k, window_size):
        off = in_offsets.host()
        m, _ = oracle.get_minimizers(_unpack(packed, int(off[-1])), np.diff(off), k, window_size)
        assert m.size == n_out
        return _h(m)

    # -- counting ----------------------------------------------------------------------------------
    def count_dense(self, values, n_bins, hist=None):
        c = np.bincount(values.host(), minlength=n_bins).astype(np.int64)
        if hist is not None:
            c = c + hist.host()
        return _h(c)

    def count_bytes(self, values, n_bins, hist=None):
        v = values.host()
        c = np.bincount(v[v < n_bins], minlength=n_bins).astype(np.int64)
        return _h(c if hist is None else c + hist.host())

    def count_packed(self, packed, n_bases, hist=None):
        c = np.bincount(_unpack(packed, n_bases), minlength=4).astype(np.int64)
        return _h(c if hist is None else c + hist.host())

    COUNT_BYTES_ROWS_MAX_BINS = 8

    def count_bytes_rows(self, values, offsets, n_rows, total, n_bins):
        off, v = offsets.host(), values.host()
        return _h(np.array([np.bincount(v[off[r]:off[r + 1]], minlength=n_bins)[:n_bins] for r in range(n_rows)],
                           dtype=np.int64).reshape(-1))

    def count_dense_rows(self, values, offsets, n_rows, n_bins):
        off = offsets.host()
        v = values.host()
        return _h(np.array([np.bincount(v[off[r]:off[r + 1]], minlength=n_bins) for r in range(n_rows)],
                           dtype=np.int64).reshape(-1))

    def count_weighted(self, values, weights, n, n_rows, value_stride, weight_stride, n_bins):
        v, w = values.host(), weights.host()
        out = np.zeros((n_rows, n_bins), dtype=w.dtype)
        for r in range(n_rows):
            out[r] = np.bincount(v[r * value_stride:r * value_stride + n], weights=w[r * weight_stride:r * weight_stride + n],
                                 minlength=n_bins).astype(w.dtype)
        return _h(out.reshape(-1))

    def count_sparse(self, values, key_bits=62, consume=False, partition=None, key_range=None, fast=True, skew=1.0, dest=None):
        k, c = oracle.count_sparse(values.host())
        if dest is not None:                             # (keys, counts, pos): written in place, views returned
            ok, oc, pos = dest[0].host(), dest[1].host(), int(dest[2])
            assert pos + values.size <= ok.size
            ok[pos:pos + k.size], oc[pos:pos + k.size] = k, c
            return _h(ok[pos:pos + k.size]), _h(oc[pos:pos + k.size])
        return _h(k), _h(c)

    def empty_i64(self, n):
        return _h(np.zeros(n, dtype=np.int64))

    def reduce_by_key(self, keys, weights, key_bits=62):
        if not isinstance(keys, (list, tuple)):
            keys, weights = [keys], [weights]
        k, c = oracle.merge_sparse([(a.host(), b.host()) for a, b in zip(keys, weights)])
        return _h(k), _h(c)

    def row_ids(self, offsets, n_rows, n):
        return _h(np.repeat(np.arange(n_rows, dtype=np.int64), np.diff(offsets.host())))

    def vec_ratio_rows(self, sums, offsets, n):
        off = offsets.host()
        with np.errstate(divide="ignore", invalid="ignore"):
            return _h(sums.host()[:n].astype(np.float64) / (off[1:n + 1] - off[:n]).astype(np.float64))

    def vec_compare(self, x, op, scalar):
        import operator
        f = {"<": operator.lt, "<=": operator.le, ">": operator.gt, ">=": operator.ge, "==": operator.eq, "!=": operator.ne}[op]
        return _h(f(x.host(), scalar).astype(np.uint8))

    def mask_logic(self, a, b, op):
        x = a.host().astype(bool)
        y = b.host().astype(bool) if b is not None else None
        out = {"and": lambda: x & y, "or": lambda: x | y, "xor": lambda: x ^ y, "not": lambda: ~x}[op]()
        return _h(out.astype(np.uint8))

    def mask_fill(self, mask, start, step, count, value):
        mask.host()[start:start + count * step:step] = 1 if value else 0

    def mask_rows(self, mask):
        rows = np.flatnonzero(mask.host()).astype(np.int64)
        return _h(rows), int(rows.size)

    def take_i64(self, arr, idx):
        return _h(arr.host()[idx.host()])

    def dense_to_sparse(self, hist):
        keys = np.flatnonzero(hist.host()).astype(np.int64)
        return _h(keys), _h(hist.host()[keys])

    def slice_copy(self, x, start, stop):
        return _h(x.host()[start:stop].copy())

    def unique_pairs(self, keys, values, key_bits=62, n_values=None, with_counts=False):
        if keys.size:
            pairs, counts = np.unique(np.stack([keys.host(), values.host()], axis=1), axis=0, return_counts=True)
        else:
            pairs, counts = np.zeros((0, 2), dtype=np.int64), np.zeros(0, dtype=np.int64)
        if with_counts:
            return _h(pairs[:, 0]), _h(pairs[:, 1]), _h(counts.astype(np.int64))
        return _h(pairs[:, 0]), _h(pairs[:, 1])

    def merge_add(self, a_keys, a_counts, b_keys, b_counts):
        k, c = oracle.merge_sparse([(a_keys.host(), a_counts.host()), (b_keys.host(), b_counts.host())])
        return _h(k), _h(c)

    def search_sorted(self, sorted_keys, queries, upper=False):
        return _h(np.searchsorted(sorted_keys.host(), queries.host(), side="right" if upper else "left")
                  .astype(np.int64))

    def concat(self, arrays):
        return _h(np.concatenate([a.host() for a in arrays]))

    def concat_words(self, parts, pad=2):
        return _h(np.concatenate([a.host()[:n] for a, n in parts] + [np.zeros(pad, dtype=np.int64)]))

    def synth_fastq(self, n_reads, read_len, seed, mode=0, genome_len=0, first_read=0):
        from bionumpy_amd import synth
        return _h(synth.fastq_bytes(n_reads, read_len, seed, mode, genome_len, first_read))
