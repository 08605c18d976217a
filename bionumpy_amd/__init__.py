"""bionumpy_amd — MI355X-native backend for BioNumPy's sequence hot path.

FASTQ/FASTA chunk decode -> EncodedRaggedArray -> 2-bit DNA -> k-mer / minimizer hash -> count / index,
behind BioNumPy's own API names (``open``, chunk ``.sequence``, ``EncodedArray``, ``change_encoding``,
``get_kmers``, ``count_encoded`` ...).  All numeric work runs in hand-written HIP kernels for gfx950
reached through the C-ABI in include/bnpk.h; there is no CPU fallback.
"""
from . import _native  # noqa: F401  (fails loudly if libbnpk.so is missing)
from .encoded_array import (EncodedArray, EncodedRaggedArray, as_encoded_array, change_encoding, BaseEncoding,
                            DNAEncoding, ACGTEncoding, AlphabetEncoding, QualityEncoding, ACTGEncoding, ACTGnEncoding,
                            ACGTnEncoding, DigitEncoding, ACUGEncoding, RNAENcoding, AminoAcidEncoding, BamEncoding,
                            OneToOneEncoding)
from .ragged import RaggedArray, RaggedShape
from .exceptions import FormatException, EncodingError
from . import encodings, io, sequence, streams
from .encodings import KmerEncoding
from .io import bnp_open, count_entries, FastQBuffer, TwoLineFastaBuffer, MultiLineFastaBuffer
from .sequence.debruin import DeBruijnGraph, ColoredDeBruijnGraph
from .sequence import (match_string, get_motif_scores, get_reverse_complement, get_kmers, count_kmers, get_minimizers, count_encoded, EncodedCounts, SparseKmerCounts,
                       KmerIndex, KmerLookup, KmerEncoder, Minimizers, PositionWeightMatrix, PWM)
from .streams import streamable, bincount, histogram, mean, quantile
from .memory_mapping import MemMapEncodedRaggedArray
from .datatypes import SequenceEntry, SequenceEntryWithQuality, replace

open = bnp_open

__version__ = "0.1.0"
