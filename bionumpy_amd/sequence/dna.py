"""Reverse complement on the MI355X path (bionumpy/sequence/dna.py:36-65) and canonical k-mers.

``get_reverse_complement(sequence) == complement(sequence)[..., ::-1]``: rows reversed, bases complemented, same
encoding and row layout as the input.  2-bit DNA stays packed in HBM (``bnpk_reverse_complement_packed``: one
reversal of the 2-bit groups of a word + one NOT per 32 bases); ASCII sequences go through the reference's 128-entry
complement table (``bnpk_reverse_complement_bytes``).
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
        if alphabet.upper() != "ACGT":
            raise NotImplementedError("reverse complement on the MI355X path: BaseEncoding or the ACGT alphabet")
        return "dna"
    raise ValueError("Invalid encoding for dna-complement: %s" % (encoding,))


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
    ragged, single = _as_rows(sequence)
    ops = get_ops()
    n_rows, total, offsets = len(ragged), ragged.total(), ragged.offsets()
    if kind == "dna":
        out = _PackedDna(ops.reverse_complement_packed(packed_words(ragged._data), offsets, n_rows, total), total)
    else:
        out = ops.reverse_complement_bytes(ragged._flat_data(), offsets, n_rows, total)
    if single:
        return EncodedArray(out, sequence.encoding)
    return EncodedRaggedArray._from_parts(out, None, ragged._lens, offsets, n_rows, total, sequence.encoding)
