"""Chunk containers of the sequence path: SequenceEntry{name, sequence} and
SequenceEntryWithQuality{+quality} (bionumpy/datatypes/__init__.py:40-47), with the lazy-field
behaviour of the reader's chunk objects (bionumpy/bnpdataclass/lazybnpdataclass.py:52-225): a field is
decoded from the chunk buffer the first time it is accessed."""
import numpy as np


class SequenceEntry:
    _fields = ("name", "sequence")

    def __init__(self, *values, **kwargs):
        values = dict(zip(self._fields, values), **kwargs)
        assert set(values) == set(self._fields), (values.keys(), self._fields)
        self._values = values
        self._buffer = None
        self._line_offset = 0

    # -- lazy construction from a chunk buffer (ItemGetter, lazybnpdataclass.py:19-49) -----------------
    @classmethod
    def _lazy(cls, buffer, n_lines_read=0):
        obj = cls.__new__(cls)
        obj._values = {}
        obj._buffer = buffer
        obj._line_offset = n_lines_read
        return obj

    def __getattr__(self, name):
        if name.startswith("_") or name not in self._fields:
            raise AttributeError(name)
        if name not in self._values:
            from .exceptions import FormatException
            try:
                self._values[name] = self._buffer.get_field_by_number(self._fields.index(name))
            except FormatException as e:
                e.line_number += self._line_offset
                raise
        return self._values[name]

    def __len__(self):
        if self._buffer is not None:
            return len(self._buffer)
        return len(self._values[self._fields[0]])

    def __getitem__(self, idx):
        if isinstance(idx, (int, np.integer)):
            idx = [int(idx)]
        if self._buffer is not None:      # stays lazy: the selection is a view of the chunk's text buffer, so
            from .device_vector import DeviceVector                   # that it can be written back (get_buffer)
            if isinstance(idx, DeviceVector) and idx.dtype == np.bool_:
                idx = idx.nonzero_rows()                              # the row list of a device mask never leaves HBM
            elif isinstance(idx, np.ndarray) and idx.dtype == bool:
                idx = np.flatnonzero(idx)
            return self.__class__._lazy(self._buffer[idx], self._line_offset)
        return self.__class__(**{f: getattr(self, f)[idx] for f in self._fields})

    @classmethod
    def empty(cls):
        obj = cls.__new__(cls)
        obj._values = {f: [] for f in cls._fields}
        obj._buffer = None
        obj._line_offset = 0
        return obj

    def get_buffer(self):
        return self._buffer

    def _replace(self, **kwargs):
        """bnp.replace (bnpdataclass/bnpdataclassfunction.py:14-47): same entries, some fields exchanged"""
        unknown = set(kwargs) - set(self._fields)
        assert not unknown, unknown
        return self.__class__(**{f: kwargs.get(f, None) if f in kwargs else getattr(self, f) for f in self._fields})

    def __repr__(self):
        """the entry table the reference prints (npstructures' npdataclass ``__str__``, shown in bionumpy/io/files.py:116-170,
        docs_source/source/sequences.rst:165-170): a count line, the field names and the first ten entries in columns 25
        wide; text longer than 20 letters is cut there and marked ``...``, a row of numbers is numpy's print of it, every
        cell cut at 23 characters"""
        lines = ["%s with %d entries" % (self.__class__.__name__, len(self)),
                 "".join("%25s" % f for f in self._fields)]
        head = self[:10] if len(self) > 10 else self
        columns = [_cells(getattr(head, f), min(len(self), 10)) for f in self._fields]
        lines.extend("".join("%25s" % cell for cell in row) for row in zip(*columns))
        return "\n".join(lines)

    __str__ = __repr__

    def __array_function__(self, func, types, args, kwargs):
        """np.concatenate(chunks) (bnpdataclass/lazybnpdataclass.py:178-196, bnpdataclass.py:477-493): the entries of
        all chunks, every field joined (on the device) in chunk order"""
        if func is not np.concatenate:
            return NotImplemented
        chunks = list(args[0])
        if not chunks or not all(isinstance(c, SequenceEntry) and c._fields == self._fields for c in chunks):
            return NotImplemented
        return self.__class__(**{f: np.concatenate([getattr(c, f) for c in chunks]) for f in self._fields})


def _cells(field, n):
    """the table cells of the first n values of a field"""
    out = []
    for i in range(n):
        value = field[i]
        if hasattr(value, "to_string"):
            text = value.to_string()
            text = text[:20] + "..." if len(text) > 20 else text
        elif isinstance(value, str):
            text = value[:20] + "..." if len(value) > 20 else value
        else:
            text = str(np.asarray(value))
        out.append(text[:23])
    return out


class SequenceEntryWithQuality(SequenceEntry):
    _fields = ("name", "sequence", "quality")


def replace(obj, **kwargs):
    """Replace fields of a chunk / dataclass by new values (bnpdataclass/bnpdataclassfunction.py:14-47)."""
    return obj._replace(**kwargs)
