"""Sequence ops on the hot path (bionumpy/sequence/__init__.py)."""
from .kmers import get_kmers, count_kmers, KmerEncoder
from .minimizers import get_minimizers, Minimizers
from .dna import get_reverse_complement
from .string_matcher import match_string
from . import string_matcher
from . import position_weight_matrix
from .position_weight_matrix import PWM, PositionWeightMatrix, get_motif_scores
from .count_encoded import count_encoded, EncodedCounts, SparseKmerCounts
from . import indexing
from .indexing import KmerIndex, KmerLookup
from . import debruin
from .debruin import DeBruijnGraph, ColoredDeBruijnGraph

__all__ = ["get_kmers", "count_kmers", "KmerEncoder", "Minimizers", "PositionWeightMatrix", "get_minimizers", "get_reverse_complement", "match_string", "string_matcher", "PWM", "get_motif_scores", "position_weight_matrix", "count_encoded", "EncodedCounts", "SparseKmerCounts",
           "KmerIndex", "KmerLookup", "indexing", "debruin", "DeBruijnGraph", "ColoredDeBruijnGraph"]
