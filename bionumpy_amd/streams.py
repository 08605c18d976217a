"""The slice of bionumpy.streams the count path uses (bionumpy/streams/decorators.py:9-110,
bionumpy/streams/stream.py:1-53): ``@streamable(reduction)`` maps a function over a stream of
chunks (or chunk fields) and reduces the per-chunk results, e.g. ``sum`` of k-mer histograms."""
import functools
import types

import numpy as np


class BnpStream:
    def __init__(self, stream, first_buffer=None):
        self._stream = stream
        self.first_buffer = first_buffer

    def __iter__(self):
        return self

    def __next__(self):
        return next(self._stream)


class NpDataclassStream(BnpStream):
    """stream of chunk objects (streams/stream.py:40-53); attribute access maps over the chunks.

    rebatch: what a reader hands along — a function (min_chunk_size) -> a fresh generator of chunks of that size over the same
    entries, usable while nothing has been taken from this stream.  A reduction that does not depend on where the chunks
    are cut (``streamable(sum)`` over histograms) asks for batches sized for the device instead of the reference's 5 MB."""

    def __init__(self, stream, dataclass=None, rebatch=None, shard=None):
        super().__init__(stream)
        self.dataclass = dataclass
        self._rebatch = rebatch
        self._started = False
        self._shard = shard                # io.sharding.Shard: the chunks are ONE rank's part of a file read by several ranks

    def __next__(self):
        self._started = True
        return next(self._stream)

    def _coalesced(self, min_chunk_size):
        """this stream again, cut into chunks of at least min_chunk_size bytes — or itself, if it cannot be cut anew"""
        if self._rebatch is None or self._started:
            return self
        self._started = True                                 # (the entries are handed out through the new stream from now on)
        return NpDataclassStream(self._rebatch(min_chunk_size), self.dataclass, shard=self._shard)

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return _FieldStream(self, name)


class _FieldStream(BnpStream):
    """one field of every chunk of a chunk stream (what NpDataclassStream.__getattr__ gives)"""

    def __init__(self, parent, name):
        super().__init__(getattr(chunk, name) for chunk in parent)
        self._parent, self._name = parent, name
        self._shard = getattr(parent, "_shard", None)

    def _coalesced(self, min_chunk_size):
        again = self._parent._coalesced(min_chunk_size)
        return self if again is self._parent else _FieldStream(again, self._name)


def _is_stream(x):
    return isinstance(x, (BnpStream, types.GeneratorType))


COALESCED_CHUNK = 256 << 20        # bytes per chunk a chunk-boundary-independent reduction asks a file stream for


class streamable:
    """streams/decorators.py:9-110.  coalesce=True: the reduced result does not depend on where the stream is cut into chunks
    (a sum of histograms) — a stream that comes straight from a file reader is then cut into COALESCED_CHUNK-byte chunks
    instead of the size the caller named (the reference's default is 5 MB, a few hundredths of what one launch of the
    counting kernels is sized for)."""

    def __init__(self, reduction=None, coalesce=False, merge=None):
        self._reduction = reduction
        self._coalesce = coalesce
        self._merge = merge                # (list of every rank's reduced value) -> the job's: plain numpy results of sharded streams

    def __call__(self, func):
        reduction = self._reduction
        coalesce = self._coalesce and reduction is not None
        merge = self._merge

        @functools.wraps(func)
        def wrapped(*args, **kwargs):
            stream_args = [i for i, a in enumerate(args) if _is_stream(a)]
            stream_keys = [k for k, v in kwargs.items() if _is_stream(v)]
            if not stream_args and not stream_keys:
                return func(*args, **kwargs)
            if coalesce and len(stream_args) + len(stream_keys) == 1:
                args = tuple(a._coalesced(COALESCED_CHUNK) if hasattr(a, "_coalesced") else a for a in args)
                kwargs = {k: (v._coalesced(COALESCED_CHUNK) if hasattr(v, "_coalesced") else v) for k, v in kwargs.items()}

            def results():
                iters = {i: iter(args[i]) for i in stream_args}
                kiters = {k: iter(kwargs[k]) for k in stream_keys}
                while True:
                    try:
                        a = [next(iters[i]) if i in iters else v for i, v in enumerate(args)]
                        kw = {k: (next(kiters[k]) if k in kiters else v) for k, v in kwargs.items()}
                    except StopIteration:
                        return
                    yield func(*a, **kw)

            shards = [v._shard for v in list(args) + list(kwargs.values()) if getattr(v, "_shard", None) is not None]
            if reduction is None:
                out = BnpStream(results())
                out._shard = shards[0] if shards else None       # what is mapped over a rank's part is a rank's part
                return out
            reduced = reduction(results())
            # the stream was one rank's part of a file that several ranks read (bnp.open(..., shard=...), io/sharding.py): what
            # the ranks reduced is put together — the sum of the chunks' histograms of ALL ranks is what the reference's loop
            # over the whole file returns (EncodedCounts.__add__ across GPUs: SURVEY §8e)
            if shards:
                reduced = _merged_over_ranks(reduced, shards[0], merge)
            return reduced

        return wrapped


def _merged_over_ranks(reduced, shard, merge=None):
    """what every rank reduced from its part of a file, put together (collective: every rank of the shard's group gets here,
    also one whose part held no entry — its ``sum`` of nothing is the int 0, so the ranks first tell each other what they
    hold, and a rank without a result joins the merge with an empty one of the others' kind)"""
    from . import parallel
    if not parallel.group_is_up():
        return reduced                                       # (a shard given by hand, no process group: the caller merges)
    mine = reduced._merge_descriptor() if hasattr(reduced, "_merge_descriptor") else None
    proto = next((d for d in parallel.all_gather_objects(mine, shard.group) if d is not None), None)
    if proto is None:
        # nothing that knows how to merge itself on any rank: plain numbers and numpy arrays.  The reductions of this module
        # say how their values add up (bincounts, histograms, sums); anybody else's reduction of a sharded stream stays the
        # rank's partial result — said aloud, because the reference's caller expects the whole file's
        if merge is not None:
            return merge(parallel.all_gather_objects(reduced, shard.group))
        import warnings
        warnings.warn("the reduction of a sharded stream (bnp.open(..., shard=...)) returned %s, which this package cannot merge "
                      "over the ranks: every rank holds the result of ITS part of the file" % type(reduced).__name__, stacklevel=3)
        return reduced
    if mine is None:
        reduced = proto[0]._empty_like_descriptor(proto)
    return reduced._merged_over_ranks(shard)


# ---- streamable numpy reductions (bionumpy/streams/reductions.py:1-63): per chunk, then joined ---------------------------------
def _join_bincounts(parts):
    total = None
    for part in parts:
        part = np.asarray(part)
        if total is None:
            total = part.copy()
        elif total.size >= part.size:
            total[:part.size] += part
        else:
            part = part.copy()
            part[:total.size] += total
            total = part
    return total


def _join_histograms(parts):
    first = next(parts, None)
    if first is None:                                    # (no chunk: a rank of a sharded job whose part held no entry)
        return None
    hist, edges = first
    hist = np.asarray(hist).copy()
    for h, _ in parts:                                   # (the caller gives explicit bins / range, or the edges would differ)
        hist += h
    return hist, edges


def _merge_histograms(parts):
    parts = [p for p in parts if isinstance(p, tuple)]
    return _join_histograms(iter(parts)) if parts else None


def _merge_sums(parts):
    parts = [p for p in parts if not (np.ndim(p) == 0 and p == 0)]      # (sum of no chunk is the int 0)
    return _join_bincounts(parts) if parts else 0


bincount = streamable(_join_bincounts, merge=lambda parts: _join_bincounts([p for p in parts if p is not None]))(np.bincount)
histogram = streamable(_join_histograms, merge=_merge_histograms)(np.histogram)


@streamable(sum, merge=_merge_sums)
def _sum_and_n(array, axis=None):
    n = array.size if axis is None else len(array)
    return np.append(np.asarray(np.sum(array, axis=axis), dtype=np.float64), n)


@streamable()
def _row_mean(array, axis=None):
    return np.mean(array, axis=axis)


def mean(array, axis=None):
    """np.mean over an array or a stream of chunks: sums and counts per chunk, one division at the end (axis None / 0); along
    the rows (axis 1 / -1) every chunk gives its own means (reductions.py:41-57)"""
    if axis is not None and axis != 0:
        return _row_mean(array, axis)
    t = _sum_and_n(array, axis=axis)
    return t[:-1] / t[-1]                                  # (a 1-element array for axis=None, as reductions.py:41-57 returns it)


def quantile(array, quantiles, axis=None):
    """the values below which the given fractions of a stream of small non-negative integers lie, from its bincount
    (reductions.py:60-66)"""
    cumulative = np.cumsum(bincount(array))
    return np.searchsorted(cumulative, np.asarray(quantiles) * cumulative[-1])

