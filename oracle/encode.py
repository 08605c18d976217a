"""ASCII -> DNA code LUT and ragged gather (test infrastructure — see oracle/__init__.py).

Follows:
  bionumpy/encodings/alphabet_encoding.py:19-51  AlphabetEncoding._initialize/_encode/_decode
  bionumpy/encoded_array.py:655-695              change_encoding (ravel -> decode -> encode)
  bionumpy/io/file_buffers.py:315-338            TextBufferExtractor.get_field_by_number (ragged view)
  bionumpy/encodings/__init__.py:11-26           DigitEncodingFactory / QualityEncoding (byte - 33)
"""
import numpy as np

from .ragged import flat_indices
from .text import EncodingError


def dna_lut(alphabet="ACGT"):
    """256-entry LUT, case-insensitive, 255 = invalid (alphabet_encoding.py:19-32)."""
    upper = np.array([ord(c) for c in alphabet.upper()], dtype=np.uint8)
    lut = np.full(256, 255, dtype=np.uint8)
    lut[upper] = np.arange(len(alphabet))
    lut[upper + (ord("a") - ord("A"))] = np.arange(len(alphabet))
    return lut


_DNA_LUT = dna_lut()
_DNA_ALPHABET = np.array([ord(c) for c in "ACGT"], dtype=np.uint8)


def gather_rows(data, starts, lengths):
    """RaggedView -> contiguous flat array (EncodedRaggedArray.ravel(); copy A6 of SURVEY §8a)."""
    data = np.asarray(data)
    return data[flat_indices(starts, lengths)]


def encode_dna(byte_array, lut=_DNA_LUT, alphabet_size=4):
    """AlphabetEncoding._encode (alphabet_encoding.py:34-46): LUT gather, EncodingError(offset)."""
    byte_array = np.asarray(byte_array, dtype=np.uint8)
    ret = lut[byte_array]
    if np.any(ret >= alphabet_size):
        offset = int(np.flatnonzero(ret.ravel() == 255)[0])
        raise EncodingError("Invalid character(s) when encoding to AlphabetEncoding", offset)
    return ret


def decode_dna(codes):
    """AlphabetEncoding._decode (alphabet_encoding.py:48-51)."""
    return _DNA_ALPHABET[np.asarray(codes)]


def quality_scores(byte_array):
    """QualityEncoding = DigitEncodingFactory('!')._encode (encodings/__init__.py:15-16,26)."""
    return np.asarray(byte_array, dtype=np.uint8) - np.uint8(33)
