"""Ragged-layout helpers (test infrastructure — see oracle/__init__.py).

Restates the parts of ``npstructures.RaggedShape`` the hot path relies on
(SURVEY.md Appendix A): row r of a contiguous ragged array occupies the flat
range [starts[r], starts[r] + lengths[r]) with starts = cumsum(lengths) - lengths.
"""
import numpy as np


def row_starts(lengths):
    """starts of each row in the compact flat layout (int64)."""
    lengths = np.asarray(lengths, dtype=np.int64)
    out = np.zeros(lengths.size, dtype=np.int64)
    if lengths.size > 1:
        np.cumsum(lengths[:-1], out=out[1:])
    return out


def flat_indices(starts, lengths):
    """Flat indices of every element of the rows (starts[r] + 0..lengths[r]).

    This is the index vector a ``RaggedView`` expands to when it is ravelled
    (npstructures ``_flatten_myself``; reference call site
    bionumpy/encoded_array.py:688-690 -> ``encoded_array.ravel()``).
    """
    starts = np.asarray(starts, dtype=np.int64)
    lengths = np.asarray(lengths, dtype=np.int64)
    total = int(lengths.sum())
    if total == 0:
        return np.zeros(0, dtype=np.int64)
    compact = row_starts(lengths)
    # offset of every flat element relative to its row start, then add the view start
    shift = np.repeat(starts - compact, lengths)
    return np.arange(total, dtype=np.int64) + shift


def row_reduce(flat, lengths):
    """np.sum / np.min / np.max(ragged, axis=-1) of uint8 rows (npstructures RaggedArray reductions as used in
    scripts/small_example.py:36-46) -> (sums int64, mins uint8, maxs uint8); an empty row gives (0, 255, 0), the
    identities of the three reductions over uint8."""
    flat = np.asarray(flat, dtype=np.uint8)
    lengths = np.asarray(lengths, dtype=np.int64)
    n = lengths.size
    sums = np.zeros(n, dtype=np.int64)
    mins = np.full(n, 255, dtype=np.uint8)
    maxs = np.zeros(n, dtype=np.uint8)
    nz = np.flatnonzero(lengths > 0)
    if nz.size:
        starts = (np.cumsum(lengths) - lengths)[nz]
        data = flat[:int(lengths.sum())]
        sums[nz] = np.add.reduceat(data.astype(np.int64), starts)
        mins[nz] = np.minimum.reduceat(data, starts)
        maxs[nz] = np.maximum.reduceat(data, starts)
    return sums, mins, maxs


def col_sums(flat, lengths):
    """np.sum(ragged, axis=0) and the number of rows that reach every column (npstructures RaggedArray.sum/mean over
    axis 0, scripts/small_example.py:20-22,49-52) -> (sums int64[max_len], counts int64[max_len])"""
    flat = np.asarray(flat)
    lengths = np.asarray(lengths, dtype=np.int64)
    n_cols = int(lengths.max()) if lengths.size else 0
    starts = np.cumsum(lengths) - lengths
    cols = np.arange(int(lengths.sum()), dtype=np.int64) - np.repeat(starts, lengths)
    sums = np.bincount(cols, weights=flat[:cols.size].astype(np.float64), minlength=n_cols).astype(np.int64)
    counts = np.bincount(cols, minlength=n_cols).astype(np.int64)
    return sums, counts
