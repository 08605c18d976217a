"""RaggedArray: rows of different length over one flat HBM buffer.

Mirrors the parts of ``npstructures.RaggedArray`` the sequence path uses
(SURVEY.md Appendix A): ``len``, ``.lengths``, ``.shape == (n_rows, lengths)``, ``.size``,
``.ravel()``, row / row-slice / column-slice / mask indexing, iteration, ``tolist``.

A ragged array is (flat data, row starts, row lengths).  It is *compact* when the rows tile
the flat buffer (starts == cumsum(lengths) - lengths); otherwise it is a view (the reference's
``RaggedView``: e.g. the sequence lines inside a raw FASTQ chunk, io/file_buffers.py:335-338)
and ``ravel()`` gathers it into a compact buffer on the device (copy A6 of SURVEY §8a).

The flat data and the row tables live in HBM (``HArray``); indexing and printing are host
conveniences that pull what they need.
"""
import numpy as np

from .device import HArray, as_harray, as_u8
from .ops import get_ops


def _row_starts(lengths):
    out = np.zeros(len(lengths), dtype=np.int64)
    if len(lengths) > 1:
        np.cumsum(lengths[:-1], out=out[1:])
    return out


class RaggedArray:
    def __init__(self, data, shape=None, dtype=None, safe_mode=True):
        if shape is None:                                   # list of rows
            rows = [np.asarray(r) for r in data]
            lengths = np.array([r.size for r in rows], dtype=np.int64)
            flat = np.concatenate(rows) if rows else np.zeros(0, dtype=dtype or np.int64)
            if dtype is not None:
                flat = flat.astype(dtype)
            self._init(as_harray(flat), None, as_harray(lengths), None, len(rows), int(lengths.sum()))
            return
        if isinstance(shape, RaggedArray):
            shape = shape._shape
        if isinstance(shape, RaggedShape):
            self._init(as_harray(data), shape.starts, shape.lens, shape.offsets, shape.n_rows, shape.total)
            return
        if isinstance(shape, tuple):                       # (n_rows, lengths) as returned by .shape
            shape = shape[-1]
        lengths = np.asarray(shape, dtype=np.int64)
        data = as_harray(data)
        total = int(lengths.sum())
        if safe_mode:
            assert data.size == total, (data.size, total)
        self._init(data, None, as_harray(lengths), None, lengths.size, total)

    def _init(self, data, starts, lens, offsets, n_rows, total):
        self._data = data          # HArray, flat storage
        self._starts = starts      # HArray int64[n] or None (compact)
        self._lens = lens          # HArray int64[n]
        self._offsets = offsets    # HArray int64[n+1] compact offsets (lazy)
        self._n_rows = int(n_rows)
        self._total = total        # int or None (lazy)

    # -- construction helpers -------------------------------------------------------------------------
    @classmethod
    def _from_parts(cls, data, starts, lens, offsets, n_rows, total):
        obj = cls.__new__(cls)
        obj._init(data, starts, lens, offsets, n_rows, total)
        return obj

    def _like(self, data, starts, lens, offsets, n_rows, total):
        return RaggedArray._from_parts(data, starts, lens, offsets, n_rows, total)

    @property
    def _shape(self):
        return RaggedShape(self._starts, self._lens, self._offsets, self._n_rows, self._total)

    # -- basic properties ------------------------------------------------------------------------------------
    def __len__(self):
        return self._n_rows

    @property
    def lengths(self):
        return self._lens.host()

    @property
    def shape(self):
        return (self._n_rows, self.lengths)

    @property
    def size(self):
        return self.total()

    @property
    def dtype(self):
        return self._data.dtype

    def total(self):
        if self._total is None:
            self.offsets()
        return self._total

    def is_compact(self):
        return self._starts is None

    def offsets(self):
        """compact row offsets (n+1) as an HArray; computed on the device on first use"""
        if self._offsets is None:
            self._offsets, self._total = get_ops().row_offsets(self._lens, 1)
        return self._offsets

    def _flat_data(self):
        """the flat storage, materialised (EncodedRaggedArray overrides this for packed DNA)"""
        return self._data

    # -- ravel: gather a view into a compact buffer (device) ------------------------------------------------------
    def _compact(self):
        if self._starts is None:
            return
        data = self._flat_data()
        total = self.total()
        if data.dtype == np.uint8:
            new = get_ops().gather_rows(data, self._starts, self.offsets(), self._n_rows, total, 0)
        else:
            # column/row slices of numeric results: small presentation-side gathers
            starts, lens = self._starts.host(), self._lens.host()
            idx = np.repeat(starts - _row_starts(lens), lens) + np.arange(total, dtype=np.int64)
            new = HArray(host=data.host()[idx])
        self._data, self._starts = new, None

    def ravel(self):
        self._compact()
        return self._flat_data().host()

    def raw(self):
        return self

    # -- indexing (host side) ------------------------------------------------------------------------------------------
    def _host_starts(self):
        if self._starts is not None:
            return self._starts.host()
        return self.offsets().host()[:-1]

    def _row(self, i):
        n = self._n_rows
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError("row %d out of range for %d rows" % (i, n))
        s = int(self._host_starts()[i])
        return self._flat_data().host()[s:s + int(self.lengths[i])]

    def _wrap_row(self, row):
        return row

    def _select_rows(self, idx):
        starts = self._host_starts()[idx]
        lens = self.lengths[idx]
        return self._like(self._flat_data(), HArray(host=np.ascontiguousarray(starts)),
                          HArray(host=np.ascontiguousarray(lens)), None, len(lens), int(lens.sum()))

    def _col_slice(self, sl):
        starts, lens = self._host_starts(), self.lengths
        if not isinstance(sl, slice):
            raise IndexError("ragged columns are indexed by an int or a slice")
        if sl.step not in (None, 1):
            # a strided column slice is not a (start, length) view of the flat data: the selected elements are gathered
            # (npstructures RaggedArray does the same: index arithmetic, then one take)
            picks = [np.arange(int(n))[sl] for n in lens]
            new_lens = np.array([p.size for p in picks], dtype=np.int64)
            flat_idx = np.concatenate([s0 + p for s0, p in zip(starts, picks)]) if len(picks) else np.zeros(0, dtype=np.int64)
            data = as_harray(np.ascontiguousarray(self._flat_data().host()[flat_idx.astype(np.int64)]))
            return self._like(data, None, as_harray(new_lens), None, len(new_lens), int(new_lens.sum()))
        lo = 0 if sl.start is None else sl.start
        lo_abs = np.minimum(lo, lens) if lo >= 0 else np.maximum(lens + lo, 0)
        if sl.stop is None:
            hi_abs = lens
        else:
            hi = sl.stop
            hi_abs = np.minimum(hi, lens) if hi >= 0 else np.maximum(lens + hi, 0)
        new_lens = np.maximum(hi_abs - lo_abs, 0)
        return self._like(self._flat_data(), HArray(host=starts + lo_abs), HArray(host=new_lens), None, len(lens),
                          int(new_lens.sum()))

    def __getitem__(self, idx):
        if isinstance(idx, tuple):
            if len(idx) != 2:
                raise IndexError("ragged arrays are two-dimensional")
            rows, cols = idx
            if rows is Ellipsis:
                rows = slice(None)
            if isinstance(rows, (int, np.integer)):
                row = self._row(int(rows))
                return self._wrap_row(row[cols])
            sub = self if (isinstance(rows, slice) and rows == slice(None)) else self[rows]
            if isinstance(cols, (int, np.integer)):
                starts, lens = sub._host_starts(), sub.lengths
                c = np.where(cols >= 0, cols, lens + cols)
                if np.any((c < 0) | (c >= lens)):
                    raise IndexError("column index out of range")
                return self._wrap_row(sub._flat_data().host()[starts + c])
            return sub._col_slice(cols)
        if isinstance(idx, (int, np.integer)):
            return self._wrap_row(self._row(int(idx)))
        if isinstance(idx, slice):
            return self._select_rows(idx)
        idx = np.asarray(idx)
        if idx.dtype == bool:
            idx = np.flatnonzero(idx)
        return self._select_rows(idx.astype(np.int64))

    # -- per-row reductions (npstructures RaggedArray.sum/mean/min/max(axis=-1); scripts/small_example.py:36-46) ----
    def _row_reduce(self, what, as_float=False):
        self._compact()
        from .device_vector import DeviceVector             # one value per row, left in HBM (device_vector.py)
        dtype = np.dtype(self.dtype)
        if dtype in (np.dtype(np.uint8), np.dtype(np.bool_)):
            out = get_ops().row_reduce_u8(as_u8(self._data), self.offsets(), self._n_rows, want=(what,))[what]
            if dtype == np.bool_ and what != "sum":
                return DeviceVector(out, np.bool_)
            return DeviceVector(out)
        # everything else is reduced as int64 or float64 (k-mer hashes, motif scores): bnpk_row_reduce_wide
        data = self._flat_data()
        data = data._unpacked() if hasattr(data, "_unpacked") else data
        if as_float and dtype != np.dtype(np.float64):      # (np.mean of integers accumulates in float64: no wrap-around)
            data = HArray(dev=data.dev().double()) if data.on_device else HArray(host=data.host().astype(np.float64))
        elif dtype not in (np.dtype(np.int64), np.dtype(np.float64)):
            wide = np.float64 if np.issubdtype(dtype, np.floating) else np.int64
            data = HArray(host=data.host().astype(wide))
        out = get_ops().row_reduce_wide(data, self.offsets(), self._n_rows, want=(what,))[what]
        if what != "sum" and out.dtype != dtype:            # min / max keep the element type (sums widen, as numpy's do)
            return out.host().astype(dtype)
        return DeviceVector(out)

    def _host_columns(self):
        """(values, column index of every value) of the compact flat data, on the host: the column-wise reductions of
        anything but uint8 / bool, and axis=None — index arithmetic for the rare shapes, not a hot path"""
        self._compact()
        data = self._flat_data()
        data = data._unpacked() if hasattr(data, "_unpacked") else data
        lens = self.lengths
        cols = np.arange(int(lens.sum()), dtype=np.int64) - np.repeat(np.cumsum(lens) - lens, lens)
        return data.host()[:int(lens.sum())], cols

    def _col_sums(self):
        """(sums, counts) per column: the rows that are long enough contribute (axis=0)"""
        n_cols = int(self.lengths.max()) if self._n_rows else 0
        if self.dtype not in (np.uint8, np.bool_):
            values, cols = self._host_columns()
            sums = np.zeros(n_cols, dtype=np.float64 if np.issubdtype(values.dtype, np.floating) else np.int64)
            np.add.at(sums, cols, values)
            return sums, np.bincount(cols, minlength=n_cols).astype(np.int64)
        self._compact()
        sums, counts = get_ops().col_sums_u8(as_u8(self._data), self.offsets(), self._n_rows, self.total(), n_cols)
        return sums.host(), counts.host()

    def _flat_values(self):
        self._compact()
        data = self._flat_data()
        data = data._unpacked() if hasattr(data, "_unpacked") else data
        return data.host()[:self.total()]

    @staticmethod
    def _check_axis(axis):
        if axis not in (None, 0, 1, -1):
            raise ValueError("axis %r of a ragged (two-dimensional) array" % (axis,))

    def _n_true(self):
        """number of set flags of a compact bool array, counted on the device (bnpk_byte_census)"""
        self._compact()
        return int(get_ops().mask_rows(as_u8(self._data))[1]) if self.total() else 0

    def _any_empty_row(self):
        """is there a row without an element?  (asked on the device where the lengths live there)"""
        lens = self._lens
        if getattr(lens, "on_device", False) and self._n_rows:
            from .device_vector import DeviceVector
            d = lens.dev()
            return bool((DeviceVector(HArray(dev=d)) == 0).any())
        return bool(np.any(self.lengths == 0))

    # The reduction METHODS default to ``axis=None`` — the whole array, one scalar — as npstructures' RaggedArray methods do
    # (pinned by the reference's doctest docs_source/source/reading_files.rst:46-54: ``chunk.quality.mean()`` prints one
    # number per chunk); per row is ``axis=-1``, per column ``axis=0``.
    def any(self, axis=None):
        """np.any: over everything (None), per row (axis -1; an empty row gives False) or per column (0) — flags are
        reduced where they are (what the reference's callers do with match_string's result: string_matcher.py:16-55)"""
        self._check_axis(axis)
        if np.dtype(self.dtype) == np.bool_:
            if axis is None:
                return self._n_true() > 0
            if axis != 0:
                return self._row_reduce("sum") > 0
        if axis is None:
            return bool(self._flat_values().any())
        if axis == 0:
            return self._col_sums()[0] > 0 if np.dtype(self.dtype) == np.bool_ else self._nonzero()._col_sums()[0] > 0
        return self._nonzero()._row_reduce("sum") > 0

    def all(self, axis=None):
        """np.all: over everything, per row (an empty row gives True) or per column"""
        self._check_axis(axis)
        flags = self if np.dtype(self.dtype) == np.bool_ else self._nonzero()
        if axis is None:
            return flags._n_true() == flags.total()
        if axis == 0:
            sums, counts = flags._col_sums()
            return sums == counts
        return ~(flags._negated()._row_reduce("sum") > 0)

    def _nonzero(self):
        """x != 0 as a bool ragged array of the same rows"""
        self._compact()
        values = self._flat_values()
        return RaggedArray._from_parts(HArray(host=values != 0), None, self._lens, self._offsets, self._n_rows, self._total)

    def _negated(self):
        """~flags of a compact bool array, on the device"""
        self._compact()
        from .device import as_bool
        flipped = as_bool(get_ops().mask_logic(as_u8(self._data), None, "not"))
        return RaggedArray._from_parts(flipped, None, self._lens, self._offsets, self._n_rows, self._total)

    def _device_total(self):
        """sum of all elements of uint8 rows as a Python int: the per-row sums (one pass where the rows lie, also for a
        text column that was never gathered) added up on the device — exact, so ``/ size`` is numpy's float64 mean"""
        if not self.total():
            return 0
        sums = self._row_reduce("sum").harray()
        return int(sums.dev().sum().item()) if sums.on_device else int(sums.host().sum())

    def sum(self, axis=None):
        self._check_axis(axis)
        if axis is None:
            if np.dtype(self.dtype) == np.bool_:             # np.sum(sequence == "G"): counted on the device (README.rst:38-42)
                return self._n_true()
            if np.dtype(self.dtype) == np.uint8:             # (numpy adds uint8 up in uint64)
                return np.uint64(self._device_total())
            return self._flat_values().sum()
        if axis == 0:
            return self._col_sums()[0]
        return self._row_reduce("sum")

    def mean(self, axis=None):
        self._check_axis(axis)
        if axis is None:
            if np.dtype(self.dtype) == np.uint8 and self.total():
                return np.float64(self._device_total()) / self.total()      # (integers below 2^53: the exact sum, as numpy's)
            return self._flat_values().mean()
        if axis == 0:
            if self.dtype not in (np.uint8, np.bool_):                      # (accumulated in float64, as np.mean does)
                values, cols = self._host_columns()
                n_cols = int(self.lengths.max()) if self._n_rows else 0
                return np.bincount(cols, weights=values.astype(np.float64), minlength=n_cols) / np.bincount(cols, minlength=n_cols)
            sums, counts = self._col_sums()
            return sums / counts                                            # every column < max length has a row
        with np.errstate(invalid="ignore", divide="ignore"):
            from .device_vector import DeviceVector
            if np.dtype(self.dtype) in (np.dtype(np.uint8), np.dtype(np.bool_)):
                sums = self._row_reduce("sum")
                return DeviceVector(get_ops().vec_ratio_rows(sums.harray(), self.offsets(), self._n_rows))   # an empty row gives nan, as in numpy
            return np.asarray(self._row_reduce("sum", as_float=True)) / self.lengths

    def _extreme(self, what, axis):
        if axis is None:
            values = self._flat_values()
            return values.min() if what == "min" else values.max()
        if axis == 0:                                                       # per column, over the rows that reach it
            values, cols = self._host_columns()
            n_cols = int(self.lengths.max()) if self._n_rows else 0
            if values.dtype == np.bool_:                                    # (np.iinfo knows no bool: min = all, max = any per column)
                as_u8 = RaggedArray._from_parts(HArray(host=values.view(np.uint8)), None, as_harray(self.lengths.astype(np.int64)), None,
                                                self._n_rows, values.size)
                return as_u8._extreme(what, 0).astype(np.bool_)
            if what == "min":
                out = np.full(n_cols, np.inf if np.issubdtype(values.dtype, np.floating) else np.iinfo(values.dtype).max, dtype=values.dtype)
                np.minimum.at(out, cols, values)
            else:
                out = np.full(n_cols, -np.inf if np.issubdtype(values.dtype, np.floating) else np.iinfo(values.dtype).min, dtype=values.dtype)
                np.maximum.at(out, cols, values)
            return out
        if axis not in (-1, 1):
            raise ValueError("axis %r of a ragged (two-dimensional) array" % (axis,))
        if self._any_empty_row():
            raise ValueError("zero-size row in a reduction which has no identity")
        return self._row_reduce(what)

    def min(self, axis=None):
        return self._extreme("min", axis)

    def max(self, axis=None):
        return self._extreme("max", axis)

    def std(self, axis=None):
        """np.std (population, ddof 0): over everything, or per row — presentation-side, on the host"""
        self._check_axis(axis)
        values = self._flat_values().astype(np.float64)
        if axis is None:
            return values.std()
        if axis == 0:
            raise NotImplementedError("std along the columns of a ragged array")
        lens = self.lengths
        with np.errstate(invalid="ignore", divide="ignore"):
            row_of = np.repeat(np.arange(self._n_rows), lens)
            means = np.bincount(row_of, weights=values, minlength=self._n_rows) / lens
            dev2 = (values - means[row_of]) ** 2
            return np.sqrt(np.bincount(row_of, weights=dev2, minlength=self._n_rows) / lens)

    @staticmethod
    def _concatenate(arrays):
        """np.concatenate of ragged arrays == the rows of all of them, in order (npstructures RaggedArray
        __array_function__; what np.concatenate(chunks) does to every field): flat data joined on the device."""
        for a in arrays:
            a._compact()
        ops = get_ops()
        flats = [a._flat_data() for a in arrays]
        flats = [f._unpacked() if hasattr(f, "_unpacked") else f for f in flats]
        lens = np.concatenate([a.lengths for a in arrays]) if arrays else np.zeros(0, dtype=np.int64)
        data = ops.concat(flats) if sum(f.size for f in flats) else flats[0]
        return arrays[0]._like(data, None, as_harray(lens.astype(np.int64)), None, lens.size, int(lens.sum()))

    def __array_function__(self, func, types, args, kwargs):
        if func is np.concatenate:
            arrays = list(args[0])
            if not all(isinstance(a, RaggedArray) for a in arrays) or kwargs.get("axis", 0) != 0:
                return NotImplemented
            return self._concatenate(arrays)
        name = {np.sum: "sum", np.mean: "mean", np.min: "min", np.max: "max", np.amin: "min", np.amax: "max", np.any: "any",
                np.all: "all", np.std: "std"}.get(func)
        if func is np.count_nonzero and args and args[0] is self and np.dtype(self.dtype) == np.bool_ and \
                kwargs.get("axis", args[1] if len(args) > 1 else None) is None:
            return self._n_true()
        if name is None or not args or args[0] is not self:
            return NotImplemented
        axis = kwargs.get("axis", args[1] if len(args) > 1 else None)
        return getattr(self, name)(axis=axis)

    def get_column_values(self, col):
        """the values in column ``col`` of the rows that reach it (npstructures RaggedArray.get_column_values; the
        reference's docs_source/source/sequences.rst:115 counts the first letters of its sequences with it)"""
        lens = self.lengths
        reach = lens > col if col >= 0 else lens >= -col
        return (self if bool(np.all(reach)) else self[reach])[:, col]

    def __iter__(self):
        return (self[i] for i in range(self._n_rows))

    def tolist(self):
        return [np.asarray(r).tolist() for r in self]

    def __repr__(self):
        """the rows as numpy prints them, one per line (sequence/string_matcher.py:34-36, position_weight_matrix.py:186-189
        of the reference: ``ragged_array([ True False False]\n[False  True False False  True])``); the first 20 rows"""
        rows = [str(np.asarray(self._row(i))) for i in range(min(self._n_rows, 20))]
        if self._n_rows > 20:
            rows.append("...")
        return "ragged_array(%s)" % "\n".join(rows)

    def __eq__(self, other):
        if not isinstance(other, RaggedArray):
            return NotImplemented
        return (np.array_equal(self.lengths, other.lengths)
                and np.array_equal(np.asarray(self.ravel()), np.asarray(other.ravel())))


class _DeferredRows(RaggedArray):
    """uint8 rows of another buffer minus a constant, not gathered yet: a column of a text chunk under a DigitEncoding
    (``chunk.quality``: io/file_buffers.py:426-440 + encodings/__init__.py:15-16,26).  Per-row sums / means / minima / maxima
    are taken straight from the text (``bnpk_row_reduce_u8_view``) — ``np.mean(chunk.quality, axis=1) > 30`` of a read filter
    (scripts/small_example.py:36-46) never needs the 7.5 GB copy per 50 M reads that gathering the column costs.  Anything else
    that looks at the values gathers them first, transparently: ``_data`` is a property."""

    @classmethod
    def _defer(cls, base, starts, lens, offsets, n_rows, total, subtract):
        obj = cls.__new__(cls)
        obj._pending = (base, starts, int(subtract))
        obj._real = None
        obj._init(None, None, lens, offsets, n_rows, total)
        return obj

    @property
    def _data(self):
        if self._pending is not None:
            base, starts, subtract = self._pending
            self._real = get_ops().gather_rows(base, starts, self.offsets(), self._n_rows, self.total(), subtract)
            self._pending = None
        return self._real

    @_data.setter
    def _data(self, value):
        self._real = value
        if value is not None:
            self._pending = None

    @property
    def dtype(self):
        return np.dtype(np.uint8)

    def _row_reduce(self, what, as_float=False):
        if self._pending is None:
            return RaggedArray._row_reduce(self, what, as_float)
        base, starts, subtract = self._pending
        from .device_vector import DeviceVector
        return DeviceVector(get_ops().row_reduce_u8_view(base, starts, self.offsets(), self._n_rows, subtract, want=(what,))[what])


class RaggedShape:
    """row layout of a RaggedArray: (starts | None, lens, offsets | None, n_rows, total)"""

    def __init__(self, starts, lens, offsets, n_rows, total):
        self.starts, self.lens, self.offsets, self.n_rows, self.total = starts, lens, offsets, n_rows, total

    @property
    def lengths(self):
        return self.lens.host()

    def __eq__(self, other):
        if isinstance(other, RaggedShape):
            return np.array_equal(self.lengths, other.lengths)
        return NotImplemented
