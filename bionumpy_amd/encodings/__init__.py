"""Encodings on the sequence path (bionumpy/encodings/__init__.py:1-27)."""
from ..encoded_array import (Encoding, BaseEncoding, ASCIIEncoding, NumericEncoding, OneToOneEncoding,
                             AlphabetEncoding, ACGTEncoding, DNAEncoding, QualityEncoding, DigitEncodingFactory,
                             ACTGEncoding, ACTGnEncoding, ACGTnEncoding, DigitEncoding, ACUGEncoding, RNAENcoding,
                             AminoAcidEncoding, BamEncoding)
from .exceptions import EncodingError
from .kmer_encodings import KmerEncoding

__all__ = ["BaseEncoding", "Encoding", "AlphabetEncoding", "DNAEncoding", "ACGTEncoding", "QualityEncoding",
           "KmerEncoding", "EncodingError", "NumericEncoding", "OneToOneEncoding", "ASCIIEncoding",
           "DigitEncodingFactory", "ACTGEncoding", "ACTGnEncoding", "ACGTnEncoding", "DigitEncoding", "ACUGEncoding",
           "RNAENcoding", "AminoAcidEncoding", "BamEncoding"]
