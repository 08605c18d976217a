"""CPU oracle for the FASTQ -> 2-bit -> k-mer -> count hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

It is a numpy restatement of the reference algorithm (bionumpy v1.0.14 at
/root/reference plus the un-vendored dependency ``npstructures>=0.2.15``, whose
``RaggedArray`` / ``BitArray`` semantics are restated from their published
behaviour).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; ``bionumpy_amd`` never does
and has no CPU fallback.

Parity status: PINNED.  The reference package cannot be imported in the build
container (``import npstructures`` fails, no network), so the oracle is pinned
against the reference's own golden vectors instead (tests/test_oracle_goldens.py):

* docs_source/topics/kmers.rst:73-78     raw int64 31-mers of example_data/big.fq.gz
* docs_source/topics/kmers.rst:11-27     3-mer / minimizer string goldens, empty rows
* tests/test_kmer.py:85-102              label order (first base = LSB), 3-mer counts
* tests/test_minimizers.py:44-80         numeric + string minimizers
* tests/test_kmer_index.py:12-28         KmerIndex / KmerLookup
* tests/test_debruijn.py:16-35           DeBruijnGraph.forward / backward, ColoredDeBruijnGraph lookups
* tests/buffers.py:17-40,104-112         FASTQ / FASTA text fixtures
* tests/test_io_exceptions.py:11-33      FormatException.line_number
* README.rst:38-42                       G count 53686 of big.fq.gz
* bionumpy/io/files.py:117-162           511 + 489 rows at read_chunk(300000)

Every function cites the reference file:line it follows.
"""
from .ragged import row_starts, row_reduce, col_sums
from .text import (FormatException, IncompleteEntryException, EncodingError,
                   scan_one_line_buffer, FASTQ, TWO_LINE_FASTA,
                   scan_multiline_fasta, multiline_from_data, ChunkReader, open_text, join_fields)
from .encode import (dna_lut, gather_rows, encode_dna, decode_dna,
                     quality_scores)
from .kmers import (pack_2bit, sliding_window_2bit, kmer_hashes_flat,
                    get_kmers, get_kmers_generic, get_minimizers,
                    kmer_to_string, kmer_labels, count_dense, count_weighted, count_sparse,
                    merge_sparse, build_kmer_index, kmer_index_pairs, debruijn_neighbours, colored_debruijn, kmer_from_string,
                    reverse_complement, reverse_complement_hash, canonical_kmers,
                    match_string, pwm_scores)
