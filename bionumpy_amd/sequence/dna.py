"""Reverse complement on the MI355X path (bionumpy/sequence/dna.py:36-65) and canonical k-mers.

``get_reverse_complement(sequence) == complement(sequence)[..., ::-1]``: rows reversed, bases complemented, same
encoding and row layout as the input.  2-bit DNA stays packed in HBM (``bnpk_reverse_complement_packed``: one
reversal of the 2-bit groups of a word + one NOT per 32 bases); ASCII sequences go through the reference's 128-entry
complement table (``bnpk_reverse_complement_bytes``); any other alphabet over A, C, G, T, N (ACGTn, ACTG, ACTGn: one byte
per letter) is complemented by way of its letters with the same kernel between two table look-ups.
"""
from ..encoded_array import (EncodedArray, EncodedRaggedArray, BaseEncoding, AlphabetEncoding, as_encoded_array,
                             packed_words, _PackedDna)
from ..ops import get_ops
from ..streams import streamable

_complements = {"A": "T", "G": "C", "C": "G", "T": "A", "N": "N"}


def _check_encoding(encoding):
    """the encodings the reference's complement lookup accepts (dna.py:13-19), restricted to what the kernels cover"""
    if encoding == BaseEncoding:
        return "ascii"
    if isinstance(encoding, AlphabetEncoding):
        alphabet = "".join(encoding.get_alphabet())
        if alphabet.upper() == "ACGT":
            return "dna"
        for c in alphabet.upper():                           # (dna.py:23: _complements[c] for c in alphabet)
            if c not in _complements:
                raise KeyError(c)
        return "alphabet"                                    # ACGTn, ACTG, ACTGn ...: one byte per letter
    raise ValueError("Invalid encoding for dna-complement: %s" % (encoding,))


def _alphabet_tables(encoding):
    """(code -> upper-case letter, letter -> code) as 256-entry tables: an alphabet other than ACGT is complemented by way of
    its letters — the reference's lookup is as_encoded_array(complemented alphabet, encoding) (dna.py:22-26) — with the byte
    kernel in the middle: codes -> letters, rows reversed and letters complemented, letters -> codes"""
    import numpy as np
    letters = [c.upper() for c in encoding.get_alphabet()]
    to_letter = np.full(256, 255, dtype=np.uint8)
    to_letter[:len(letters)] = [ord(c) for c in letters]
    to_code = np.full(256, 255, dtype=np.uint8)
    for code, c in enumerate(letters):
        to_code[ord(c)] = code
    return to_letter, to_code


def _as_rows(sequence):
    single = isinstance(sequence, EncodedArray)
    ragged = EncodedRaggedArray(sequence.ravel(), [sequence.size]) if single else sequence
    ragged._compact()
    return ragged, single


@streamable()
def get_reverse_complement(sequence):
    """Reverse complement of one or more DNA sequences (sequence/dna.py:47-65)."""
    sequence = as_encoded_array(sequence)
    kind = _check_encoding(sequence.encoding)
    ops = get_ops()
    if kind == "ascii" and isinstance(sequence, EncodedRaggedArray) and not sequence.is_compact() and \
            hasattr(ops, "reverse_complement_rows"):
        # a column of a text chunk (chunk.sequence): reversed and complemented from where it lies, not gathered first
        n_rows, total, offsets = len(sequence), sequence.total(), sequence.offsets()
        out = ops.reverse_complement_rows(sequence._flat_data(), sequence._starts, offsets, n_rows, total)
        return EncodedRaggedArray._from_parts(out, None, sequence._lens, offsets, n_rows, total, sequence.encoding)
    ragged, single = _as_rows(sequence)
    n_rows, total, offsets = len(ragged), ragged.total(), ragged.offsets()
    if kind == "dna":
        out = _PackedDna(ops.reverse_complement_packed(packed_words(ragged._data), offsets, n_rows, total), total)
    elif kind == "alphabet":
        to_letter, to_code = _alphabet_tables(sequence.encoding)
        flat = ragged._flat_data()
        if total:
            letters = ops.lut_bytes(flat, to_letter, str(sequence.encoding))
            letters = ops.reverse_complement_bytes(letters, offsets, n_rows, total)
            out = ops.lut_bytes(letters, to_code, str(sequence.encoding))   # (a complement outside the alphabet: EncodingError, as there)
        else:
            out = flat
    else:
        out = ops.reverse_complement_bytes(ragged._flat_data(), offsets, n_rows, total)
    if single:
        return EncodedArray(out, sequence.encoding)
    return EncodedRaggedArray._from_parts(out, None, ragged._lens, offsets, n_rows, total, sequence.encoding)
