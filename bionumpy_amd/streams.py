"""The slice of bionumpy.streams the count path uses (bionumpy/streams/decorators.py:9-110,
bionumpy/streams/stream.py:1-53): ``@streamable(reduction)`` maps a function over a stream of
chunks (or chunk fields) and reduces the per-chunk results, e.g. ``sum`` of k-mer histograms."""
import functools
import types


class BnpStream:
    def __init__(self, stream, first_buffer=None):
        self._stream = stream
        self.first_buffer = first_buffer

    def __iter__(self):
        return self

    def __next__(self):
        return next(self._stream)


class NpDataclassStream(BnpStream):
    """stream of chunk objects (streams/stream.py:40-53); attribute access maps over the chunks"""

    def __init__(self, stream, dataclass=None):
        super().__init__(stream)
        self.dataclass = dataclass

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return BnpStream(getattr(chunk, name) for chunk in self)


def _is_stream(x):
    return isinstance(x, (BnpStream, types.GeneratorType))


class streamable:
    def __init__(self, reduction=None):
        self._reduction = reduction

    def __call__(self, func):
        reduction = self._reduction

        @functools.wraps(func)
        def wrapped(*args, **kwargs):
            stream_args = [i for i, a in enumerate(args) if _is_stream(a)]
            stream_keys = [k for k, v in kwargs.items() if _is_stream(v)]
            if not stream_args and not stream_keys:
                return func(*args, **kwargs)

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

            if reduction is None:
                return BnpStream(results())
            return reduction(results())

        return wrapped
