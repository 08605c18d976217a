from .kmer_indexing import KmerIndex, KmerLookup

__all__ = ["KmerIndex", "KmerLookup"]
