"""EncodedArray / EncodedRaggedArray / encodings for the sequence path.

Same names, arguments and error behaviour as the reference for this path:
  bionumpy/encoded_array.py:16-157   Encoding, OneToOneEncoding, ASCIIEncoding (BaseEncoding)
  bionumpy/encoded_array.py:161-232  EncodedRaggedArray
  bionumpy/encoded_array.py:239-500  EncodedArray
  bionumpy/encoded_array.py:547-613  as_encoded_array
  bionumpy/encoded_array.py:655-695  change_encoding
  bionumpy/encodings/alphabet_encoding.py:8-107  AlphabetEncoding, DNAEncoding = ACGTEncoding
  bionumpy/encodings/__init__.py:11-26           QualityEncoding (byte - 33)

The numeric work (ASCII -> code LUT, ragged gather, 2-bit packing) runs in the HIP kernels behind
``ops``; DNA-encoded ragged arrays keep the npstructures-BitArray-compatible packed words in HBM and
only materialise the 1-byte-per-base codes when ``.raw()`` / ``.ravel()`` is asked for.
Only the 'ACGT' alphabet has a device encoder in this round; other alphabets raise.
"""
import numpy as np

from .device import HArray, as_harray
from .exceptions import EncodingError  # noqa: F401  (re-export)
from .ops import get_ops
from .ragged import RaggedArray, RaggedShape


# ------------------------------------------------------------------------------------------ encodings
class Encoding:
    def encode(self, *args, **kwargs):
        return NotImplemented

    def get_labels(self):
        pass

    def __call__(self, *args, **kwargs):
        return self.encode(*args, **kwargs)

    def is_base_encoding(self):
        return False

    def is_one_to_one_encoding(self):
        return False

    def is_numeric(self):
        return False


def _ascii_bytes(s):
    return np.frombuffer(bytes(s, encoding="ascii"), dtype=np.uint8)


class OneToOneEncoding(Encoding):
    """encoded_array.py:38-116: str / list[str] / base-encoded arrays -> encoded arrays"""

    def is_one_to_one_encoding(self):
        return True

    def encode(self, data):
        if isinstance(data, (EncodedArray, EncodedRaggedArray)):
            assert data.encoding.is_base_encoding(), \
                "Data is already encoded. Can only encode already encoded data if it is base encoded."
            if isinstance(data, EncodedRaggedArray):
                return self._encode_ragged(data)
            return self._wrap_flat(self._encode_flat(data._harray()), data.shape)
        if isinstance(data, str):
            return self._wrap_flat(self._encode_flat(HArray(host=_ascii_bytes(data))), None)
        if isinstance(data, list):
            flat = _ascii_bytes("".join(data))
            lens = np.array([len(s) for s in data], dtype=np.int64)
            ragged = EncodedRaggedArray(EncodedArray(flat, BaseEncoding), lens)
            return self._encode_ragged(ragged)
        if isinstance(data, RaggedArray):
            ragged = EncodedRaggedArray._from_parts(data._data, data._starts, data._lens, data._offsets,
                                                    data._n_rows, data._total, BaseEncoding)
            return self._encode_ragged(ragged)
        if isinstance(data, np.ndarray):
            return self._wrap_flat(self._encode_flat(HArray(host=data.astype(np.uint8, copy=False).ravel())),
                                   data.shape)
        assert False, "Wrong input type for encode: %s" % type(data)

    def _wrap_flat(self, harray, shape):
        if self.is_numeric():
            out = harray.host()
            return out if shape is None or len(shape) == 1 else out.reshape(shape)
        return EncodedArray(harray, self, shape=shape)

    def _encode_ragged(self, ragged):
        """encode every row of a base-encoded ragged array (gather fused with the LUT on the device)"""
        shared = getattr(ragged, "_batch_rows", None)
        if shared is not None and getattr(self, "_is_dna", lambda: False)():
            rows = shared[0].encoded_rows(self, shared[1], shared[2], shared[3])    # (io/buffers.py: BatchShare)
            if rows is not None:
                return rows
        data = ragged._flat_data()
        if ragged.is_compact():
            enc = self._encode_flat(data)
        else:
            enc = self._encode_gather(data, ragged._starts, ragged.offsets(), ragged._n_rows, ragged.total())
        if self.is_numeric():
            return RaggedArray._from_parts(enc, None, ragged._lens, ragged._offsets, ragged._n_rows, ragged._total)
        out = EncodedRaggedArray._from_parts(enc, None, ragged._lens, ragged._offsets, ragged._n_rows, ragged._total, self)
        ends = enc.__dict__.pop("_gather_ends", None) if hasattr(enc, "__dict__") else None
        if ends is not None:
            out._row_ends = ends                             # (of THIS row layout: sequence/kmers.py:_rolling)
        return out

    def decode(self, data):
        raise NotImplementedError


class ASCIIEncoding(OneToOneEncoding):
    """encoded_array.py:119-137"""

    def _encode_flat(self, harray):
        return harray

    def _encode_gather(self, data, starts, offsets, n_rows, total):
        return get_ops().gather_rows(data, starts, offsets, n_rows, total, 0)

    def decode(self, data):
        return data

    def __repr__(self):
        return "ASCIIEncoding()"

    def __hash__(self):
        return hash(repr(self))

    def is_base_encoding(self):
        return True

    def __eq__(self, other):
        return isinstance(other, ASCIIEncoding)


BaseEncoding = ASCIIEncoding()


class NumericEncoding(OneToOneEncoding):
    def is_numeric(self):
        return True


class DigitEncodingFactory(NumericEncoding):
    """encodings/__init__.py:11-22; QualityEncoding = DigitEncodingFactory('!')"""

    def __init__(self, min_code):
        self._min_code = ord(min_code)

    def _encode_flat(self, harray):
        off = HArray(host=np.array([0, harray.size], dtype=np.int64))
        starts = HArray(host=np.zeros(1, dtype=np.int64))
        return get_ops().gather_rows(harray, starts, off, 1, harray.size, self._min_code)

    def _encode_gather(self, data, starts, offsets, n_rows, total):
        return get_ops().gather_rows(data, starts, offsets, n_rows, total, self._min_code)

    def _encode_ragged(self, ragged):
        if not ragged.is_compact() and hasattr(get_ops(), "row_reduce_u8_view"):
            # a column of a text chunk: left where it lies until somebody needs the values (ragged.py: _DeferredRows)
            from .ragged import _DeferredRows
            return _DeferredRows._defer(ragged._flat_data(), ragged._starts, ragged._lens, ragged._offsets, ragged._n_rows,
                                        ragged._total, self._min_code)
        return super()._encode_ragged(ragged)

    def __repr__(self):
        return "DigitEncoding(min_code=%d)" % self._min_code


QualityEncoding = DigitEncodingFactory("!")


class AlphabetEncoding(OneToOneEncoding):
    """encodings/alphabet_encoding.py:8-99 (case-insensitive, EncodingError(offset) on other bytes)"""

    def __init__(self, alphabet):
        self._raw_alphabet = [c.upper() for c in alphabet]
        self._alphabet = np.array([ord(c) for c in self._raw_alphabet], dtype=np.uint8)

    @property
    def alphabet_size(self):
        return self._alphabet.size

    def get_alphabet(self):
        return [chr(c) for c in self._alphabet]

    def get_labels(self):
        return self.get_alphabet()

    def _is_dna(self):
        return "".join(self._raw_alphabet) == "ACGT"

    def _lookup(self):
        """the reference's 256-entry table (alphabet_encoding.py:19-32): both cases of a letter -> its code, else 255"""
        lut = np.full(256, 255, dtype=np.uint8)
        lut[self._alphabet] = np.arange(self._alphabet.size)
        lut[self._alphabet + (ord("a") - ord("A"))] = np.arange(self._alphabet.size)
        return lut

    def _encode_flat(self, harray):
        if self._is_dna():                                   # fused LUT + 2-bit packing
            codes, packed = get_ops().encode_dna_flat(harray, want_codes=False, want_packed=True)
            return _PackedDna(packed, harray.size)
        return get_ops().lut_bytes(harray, self._lookup(), str(self))

    def _encode_gather(self, data, starts, offsets, n_rows, total):
        if self._is_dna():
            ops = get_ops()
            if getattr(ops, "GATHER_GIVES_ROW_ENDS", False):  # (the kernel knows the rows' last bases as it goes: kept for
                codes, packed, ends = ops.gather_encode_dna(data, starts, offsets, n_rows, total, want_codes=False,
                                                            want_packed=True, want_ends=True)   # get_kmers / get_minimizers)
                out = _PackedDna(packed, total)
                out._gather_ends = ends
                return out
            codes, packed = ops.gather_encode_dna(data, starts, offsets, n_rows, total, want_codes=False, want_packed=True)
            return _PackedDna(packed, total)
        return get_ops().lut_bytes(get_ops().gather_rows(data, starts, offsets, n_rows, total, 0), self._lookup(),
                                   str(self))

    def _decode_codes(self, codes):
        return self._alphabet[np.asarray(codes)]

    def decode(self, data):
        if isinstance(data, EncodedRaggedArray):
            data._compact()
            flat = data._data
            if isinstance(flat, _PackedDna):
                ascii_ = get_ops().unpack_codes(flat.packed, flat.size, to_ascii=True)
            else:
                ascii_ = HArray(host=self._decode_codes(flat.host()))
            return EncodedRaggedArray._from_parts(ascii_, None, data._lens, data._offsets, data._n_rows,
                                                  data._total, BaseEncoding)
        if isinstance(data, EncodedArray):
            return EncodedArray(self._decode_codes(data.raw()), BaseEncoding)
        if isinstance(data, (int, np.integer)):
            return EncodedArray(self._decode_codes(np.atleast_1d(data)), BaseEncoding)
        raise Exception("Not able to decode %s with %s" % (data, self))

    def __str__(self):
        return "AlphabetEncoding('%s')" % "".join(self.get_alphabet())

    __repr__ = __str__

    def __eq__(self, other):
        return isinstance(other, AlphabetEncoding) and self._raw_alphabet == other._raw_alphabet

    def __hash__(self):
        return hash(repr(self))


ACGTEncoding = AlphabetEncoding("ACGT")
DNAEncoding = ACGTEncoding
# the other named alphabets of the reference (encodings/alphabet_encoding.py:102-111); everything but ACGT takes the
# generic look-up / dot-product kernels (bnpk_lut_bytes, bnpk_kmers_generic)
ACTGEncoding = AlphabetEncoding("ACTG")
ACTGnEncoding = AlphabetEncoding("ACTGn")
ACGTnEncoding = AlphabetEncoding("ACGTn")
DigitEncoding = AlphabetEncoding("0123456789")
ACUGEncoding = AlphabetEncoding("ACUG")
RNAENcoding = ACUGEncoding
AminoAcidEncoding = AlphabetEncoding("ACDEFGHIKLMNPQRSTVWY*")
BamEncoding = AlphabetEncoding("=ACMGRSVTWYHKDBN")


class _PackedDna:
    """DNA codes stored 2 bits per base in HBM (npstructures BitArray layout); behaves like an
    HArray of uint8 codes, unpacking on the device the first time the codes are needed."""

    def __init__(self, packed, n):
        self.packed = packed
        self._n = int(n)
        self._codes = None

    @property
    def size(self):
        return self._n

    @property
    def dtype(self):
        return np.dtype(np.uint8)

    def _unpacked(self):
        if self._codes is None:
            self._codes = get_ops().unpack_codes(self.packed, self._n, to_ascii=False)
        return self._codes

    def host(self):
        return self._unpacked().host()

    def dev(self):
        return self._unpacked().dev()


class _LazyPackedDna(_PackedDna):
    """packed DNA whose words are only produced (``make() -> HArray``) when somebody reads them: the rows a chunk takes out
    of its batch's encoded column (io/buffers.py: BatchShare) — the k-mers of the chunk are a slice of the batch's and never
    look at the chunk's own words"""

    def __init__(self, make, n):
        self._make = make
        _PackedDna.__init__(self, None, n)

    @property
    def packed(self):
        if self._make is not None:
            self._words, self._make = self._make(), None
        return self._words

    @packed.setter
    def packed(self, value):
        self._words = value
        if value is not None:
            self._make = None


def packed_words(flat):
    """packed 2-bit words (HArray int64) of a flat DNA code buffer, packing it on the device if needed"""
    if isinstance(flat, _PackedDna):
        return flat.packed
    return get_ops().pack_codes(flat)


# ------------------------------------------------------------------------------------------ arrays
class EncodedArray:
    """ndarray of codes + the encoding they are in (encoded_array.py:239-500)."""

    def __init__(self, data, encoding, shape=None):
        if isinstance(data, EncodedArray):
            assert data.encoding == encoding
            data, shape = data._store, data._shp
        if isinstance(data, (HArray, _PackedDna)):
            self._store = data
            self._shp = tuple(shape) if shape is not None else (data.size,)
        else:
            arr = np.asarray(data) if hasattr(data, "dtype") else np.asarray(data, dtype=np.uint8)
            self._store = HArray(host=arr.ravel())
            self._shp = arr.shape
        self.encoding = encoding

    # -- storage ------------------------------------------------------------------------------------
    def _harray(self):
        return self._store

    @property
    def data(self):
        return self.raw()

    def raw(self):
        return self._store.host().reshape(self._shp)

    # -- ndarray-like -------------------------------------------------------------------------------
    def __len__(self):
        return self._shp[0] if self._shp else 0

    @property
    def size(self):
        return int(np.prod(self._shp)) if self._shp else 1

    @property
    def shape(self):
        return self._shp

    @property
    def ndim(self):
        return len(self._shp)

    @property
    def dtype(self):
        return self._store.dtype

    def ravel(self):
        return EncodedArray(self._store, self.encoding, shape=(self.size,))

    def copy(self):
        return EncodedArray(self.raw().copy(), self.encoding)

    def reshape(self, *args):
        return EncodedArray(self.raw().reshape(*args), self.encoding)

    def __getitem__(self, idx):
        return EncodedArray(self.raw()[idx], self.encoding)

    def __iter__(self):
        return (EncodedArray(e, self.encoding) for e in self.raw())

    # -- text ------------------------------------------------------------------------------------------
    def to_string(self):
        if not self.encoding.is_one_to_one_encoding():
            return self.encoding.to_string(self.raw())
        raw = self.raw() if self.encoding.is_base_encoding() else self.encoding._decode_codes(self.raw())
        return bytes(np.atleast_1d(np.asarray(raw, dtype=np.uint8))).decode("ascii")

    def tolist(self):
        return self.to_string()

    def __str__(self):
        if not self.encoding.is_one_to_one_encoding():
            if self.ndim == 0:
                return self.encoding.to_string(self.raw())
            return "[" + ", ".join(self.encoding.to_string(e).strip() for e in self.raw().ravel()) + "]"
        if self.ndim <= 1:
            return self.to_string()
        return str(np.array([str(EncodedArray(r, self.encoding)) for r in self.raw().reshape(-1, self._shp[-1])]))

    def __repr__(self):
        quotes = "'" if self.encoding.is_one_to_one_encoding() else ""
        if self.encoding.is_base_encoding():
            return "encoded_array(%s%s%s)" % (quotes, str(self), quotes)
        return "encoded_array(%s%s%s, %s)" % (quotes, str(self), quotes, self.encoding)

    def __hash__(self):
        return hash(self.to_string())

    # -- comparisons (encoded_array.py:438-452: only == / != are defined) --------------------------------
    def _other_raw(self, other):
        if isinstance(other, (str, list)):
            other = as_encoded_array(other, self.encoding)
        if isinstance(other, (EncodedArray, EncodedRaggedArray)):
            return other.raw() if isinstance(other, EncodedArray) else other.raw().ravel()
        return other

    def __eq__(self, other):
        return self.raw() == self._other_raw(other)

    def __ne__(self, other):
        return self.raw() != self._other_raw(other)

    def __array__(self, dtype=None, copy=None):
        return self.raw() if dtype is None else self.raw().astype(dtype)

    def _n_codes(self):
        """number of distinct codes of the encoding, where it says (letters, k-mers), else 0"""
        enc = self.encoding
        if hasattr(enc, "alphabet_size"):
            return int(enc.alphabet_size)
        if hasattr(enc, "_alphabet_encoding"):
            return int(enc._alphabet_encoding.alphabet_size) ** int(enc.k)
        return 0

    def __array_function__(self, func, types, args, kwargs):
        """the numpy functions the reference answers for EncodedArrays (encoded_array.py:454-486).  np.concatenate joins
        1-D arrays in HBM, np.bincount with minlength is the device histogram; the others (index arithmetic on small
        arrays in the reference's own use) run on the host copy of the codes."""
        cls = self.__class__
        if func is np.concatenate:
            parts = list(args[0])
            if not all(isinstance(e, EncodedArray) for e in parts):
                return NotImplemented
            assert all(e.encoding == self.encoding for e in parts), "arrays of different encodings"
            if all(e.ndim == 1 for e in parts) and kwargs.get("axis", 0) == 0:
                stores = [e._store._unpacked() if isinstance(e._store, _PackedDna) else e._store for e in parts]
                return cls(get_ops().concat(stores), self.encoding)
            return cls(func([e.raw() for e in parts], *args[1:], **kwargs), self.encoding)
        if func is np.bincount:
            minlength = int(kwargs.get("minlength", args[1] if len(args) > 1 else 0) or 0)
            n_codes = self._n_codes()
            if n_codes and n_codes <= (1 << 26) and len(args) <= 2 and set(kwargs) <= {"minlength"} and self.ndim == 1:
                store = self._store._unpacked() if isinstance(self._store, _PackedDna) else self._store
                if store.dtype == np.uint8:                  # letters: counted as the bytes they are
                    hist = get_ops().count_bytes(store, min(256, max(n_codes, minlength))).host()
                    if hist.size < max(n_codes, minlength):
                        hist = np.concatenate([hist, np.zeros(max(n_codes, minlength) - hist.size, dtype=hist.dtype)])
                else:
                    if store.dtype != np.int64:
                        store = HArray(host=store.host().astype(np.int64))
                    hist = get_ops().count_dense(store, max(n_codes, minlength)).host()
                used = np.flatnonzero(hist)
                return hist[:max(minlength, int(used[-1]) + 1 if used.size else 0)].copy()     # numpy's length: max(minlength, max + 1)
            return np.bincount(args[0].raw(), *args[1:], **kwargs)
        if func is np.argsort:
            return np.argsort(args[0].raw(), *args[1:], **kwargs)
        if func is np.where:
            return cls(func(args[0], args[1].raw(), args[2].raw()), self.encoding)
        if func is np.zeros_like:
            return cls(func(args[0].raw(), *args[1:], **kwargs), self.encoding)
        if func is np.append:
            return cls(func(args[0].raw(), args[1].raw(), *args[2:], **kwargs), self.encoding)
        if func is np.lexsort:
            if not all(issubclass(t, (EncodedArray, np.ndarray)) for t in types):
                return NotImplemented
            return func([a.raw() if isinstance(a, EncodedArray) else np.asarray(a) for a in args[0]], *args[1:], **kwargs)
        if func is np.insert:
            return cls(func(args[0].raw(), args[1], args[2].raw(), *args[3:], **kwargs), self.encoding)
        if func is np.lib.stride_tricks.sliding_window_view:
            return cls(func(args[0].raw(), *args[1:], **kwargs), self.encoding)
        return NotImplemented


class EncodedRaggedArray(RaggedArray):
    """RaggedArray of codes + encoding (encoded_array.py:161-232)."""

    def __init__(self, data, shape, *args, **kwargs):
        assert isinstance(data, EncodedArray), data
        super().__init__(data._harray(), shape, *args, **kwargs)
        self._encoding = data.encoding

    @classmethod
    def _from_parts(cls, data, starts, lens, offsets, n_rows, total, encoding):
        obj = cls.__new__(cls)
        obj._init(data, starts, lens, offsets, n_rows, total)
        obj._encoding = encoding
        return obj

    def _like(self, data, starts, lens, offsets, n_rows, total):
        return EncodedRaggedArray._from_parts(data, starts, lens, offsets, n_rows, total, self._encoding)

    @property
    def encoding(self):
        return self._encoding

    def _wrap_row(self, row):
        return EncodedArray(row, self._encoding)

    def raw(self):
        self._compact()
        return RaggedArray._from_parts(self._as_plain(self._data), None, self._lens, self._offsets, self._n_rows,
                                       self._total)

    @staticmethod
    def _as_plain(flat):
        return flat._unpacked() if isinstance(flat, _PackedDna) else flat

    def ravel(self):
        self._compact()
        return EncodedArray(self._data, self._encoding)

    def _compact(self):
        if self._starts is not None and isinstance(self._data, _PackedDna):
            self._data = self._data._unpacked()
        super()._compact()

    def tolist(self):
        return [row.to_string() for row in self]

    def copy(self):
        return EncodedRaggedArray(EncodedArray(self.ravel().raw().copy(), self._encoding), self.lengths)

    def __repr__(self):
        if len(self) == 0:
            return ""
        n = 5 if self.size > 1000 else len(self)
        rows = [str(self[i]) for i in range(min(n, len(self)))]
        info = ", %s" % self.encoding if not self.encoding.is_base_encoding() else ""
        quotes = "'" if self.encoding.is_one_to_one_encoding() else ""
        indent = " " * len("encoded_ragged_array([")
        lines = ["%s%s%s%s," % (indent, quotes, r, quotes) for r in rows]
        lines[0] = lines[0].replace(indent, "encoded_ragged_array([", 1)
        if self.size > 1000:
            lines.insert(-1, "...")
        lines[-1] = lines[-1][:-1] + "]" + info + ")"
        return "\n".join(lines)

    def __eq__(self, other):
        """elementwise == against a str/EncodedArray scalar (e.g. ``sequence == "G"``) -> ragged bool array"""
        if isinstance(other, (str, EncodedArray)) and not isinstance(other, EncodedRaggedArray):
            code = as_encoded_array(other, self._encoding).raw() if isinstance(other, str) else other.raw()
            code = int(np.asarray(code).reshape(-1)[0])
            self._compact()
            store = self._as_plain(self._data)
            if store.dtype == np.uint8 and 0 <= code <= 255:     # one compare kernel; the flags stay in HBM (README.rst:38-42)
                from .device import as_bool
                flat = as_bool(get_ops().vec_compare(store, "==", code))
            else:
                flat = HArray(host=store.host() == code)
            return RaggedArray._from_parts(flat, None, self._lens, self._offsets, self._n_rows, self._total)
        if isinstance(other, EncodedRaggedArray):
            return (self.encoding == other.encoding and np.array_equal(self.lengths, other.lengths)
                    and np.array_equal(self.ravel().raw(), other.ravel().raw()))
        return NotImplemented

    __hash__ = None


# ------------------------------------------------------------------------------------------ functions
def as_encoded_array(s, target_encoding=None):
    """encoded_array.py:547-613"""
    if isinstance(s, (EncodedArray, EncodedRaggedArray)):
        if target_encoding is None or s.encoding == target_encoding:
            return s
        if not s.encoding.is_base_encoding():
            raise EncodingException("Trying to encode already encoded array with encoding %s to encoding %s. "
                                    "This is not supported. Use the change_encoding function."
                                    % (s.encoding, target_encoding))
        return target_encoding.encode(s)
    if target_encoding is None:
        target_encoding = BaseEncoding
    if target_encoding.is_numeric():
        if isinstance(s, (np.ndarray, RaggedArray)):
            return s
        if isinstance(s, list) and (len(s) == 0 or isinstance(s[0], (list, np.ndarray, int, np.integer))):
            return RaggedArray(s)
    elif isinstance(s, list) and len(s) > 0 and isinstance(s[0], EncodedArray):
        encoding = s[0].encoding
        assert all(a.encoding == encoding for a in s)
        data = np.concatenate([np.atleast_1d(a.raw()) for a in s])
        return EncodedRaggedArray(EncodedArray(data, encoding), [len(a) for a in s])
    if isinstance(s, np.ndarray) and (s.dtype == object or np.issubdtype(s.dtype, np.character)):
        s = s.tolist()
    return target_encoding.encode(s)


def from_encoded_array(encoded_array):
    """the whole content as text: a str for an EncodedArray, a list of str for the rows of an EncodedRaggedArray
    (encoded_array.py:621-652; ``str()`` of an array shows only its head)"""
    if isinstance(encoded_array, EncodedRaggedArray):
        return [from_encoded_array(row) for row in encoded_array]
    return encoded_array.to_string()


class EncodingException(Exception):
    pass


def change_encoding(encoded_array, new_encoding):
    """encoded_array.py:655-695: decode to ASCII then encode with the new encoding"""
    assert isinstance(encoded_array, (EncodedArray, EncodedRaggedArray)), \
        "Can only change encoding of EncodedArray or EncodedRaggedArray"
    if encoded_array.encoding == new_encoding:
        return encoded_array
    decoded = encoded_array if encoded_array.encoding.is_base_encoding() \
        else encoded_array.encoding.decode(encoded_array)
    if new_encoding.is_base_encoding():
        return decoded
    return new_encoding.encode(decoded)
