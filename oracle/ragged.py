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
