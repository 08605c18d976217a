/*
 * bnpk.h — C-ABI of the MI355X (gfx950) sequence hot path: FASTQ/FASTA chunk decode ->
 * 2-bit DNA -> k-mer / minimizer hash -> count / index.
 *
 * The reference (bionumpy v1.0.14, pure Python on numpy + npstructures) has no FFI for this
 * path; its only backend seam is the `bnp.set_backend(cupy)` monkey-patch
 * (bionumpy/__init__.py:47-94, bionumpy/cupy_compatible/parser.py:10-17), which puts the
 * host/device boundary at the raw uint8 chunk.  This header puts the boundary at the same
 * place and declares one entry point per numpy expression the path is built from; each
 * declaration cites the reference expression (file:line) it replaces.  The Python host side
 * (bionumpy_amd/) binds these with ctypes; see INTEGRATION.md for the stub a bionumpy
 * maintainer would add.
 *
 * Conventions
 *   - every function returns 0 (BNPK_OK) or a negative bnpk_status; nothing throws across the ABI
 *   - `d_` pointers are device (HBM) pointers, `h_` pointers are host pointers; the caller owns
 *     every buffer (the Python side allocates HBM through torch, pinned host memory through
 *     bnpk_host_alloc); the library keeps only a small grow-only scratch arena inside the ctx
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls are
 *     asynchronous on that stream unless documented otherwise.  Calls of one ctx may use different
 *     streams: the scratch arena is handed from stream to stream with an event (a call on another
 *     stream than the previous call's waits, on the device, for that call's work), everything the
 *     caller owns is the caller's to order
 *   - byte buffers handed to the scanners must be 16-byte aligned (any torch allocation is)
 *   - one ctx per GPU per host thread
 */
#ifndef BNPK_H
#define BNPK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bnpk_ctx bnpk_ctx;

typedef enum {
  BNPK_OK = 0,
  BNPK_ERR_ARG = -1,        /* bad argument (null pointer, k out of range, ...) */
  BNPK_ERR_ALIGN = -2,      /* device byte buffer not 16-byte aligned */
  BNPK_ERR_HIP = -3,        /* a HIP runtime call failed; see bnpk_last_hip_error */
  BNPK_ERR_NOMEM = -4,      /* scratch arena / workspace too small or allocation failed */
  BNPK_ERR_NODEVICE = -5,   /* no gfx950 device visible */
  BNPK_ERR_RANGE = -6       /* size exceeds what the entry point supports */
} bnpk_status;

#define BNPK_NONE INT64_MAX   /* "no error found" value of the d_err outputs below */

/* ---- lifecycle / diagnostics --------------------------------------------------------- */
int         bnpk_version(void);
const char* bnpk_strerror(int status);
int         bnpk_device_count(void);
int         bnpk_ctx_create(int device, bnpk_ctx** out);
void        bnpk_ctx_destroy(bnpk_ctx* ctx);
const char* bnpk_last_hip_error(bnpk_ctx* ctx);
/* name (<=63 chars), compute units, total HBM bytes of the ctx's device */
int         bnpk_device_info(bnpk_ctx* ctx, char* name64, int* compute_units, int64_t* hbm_bytes);

/* ---- multi-GPU: the merge of per-GPU histograms, RCCL over xGMI ------------------------------------
 * One process per GPU.  The reference has no multi-device path; these entry points replace EncodedCounts.__add__ across
 * chunks (bionumpy/sequence/count_encoded.py:38-55) for chunks counted on different GPUs.  `comm` is an ncclComm_t —
 * the caller's own, or one from bnpk_comm_init: rank 0 takes an id with bnpk_comm_unique_id and hands its
 * BNPK_COMM_ID_BYTES bytes to the other ranks (any side channel), then every rank calls bnpk_comm_init.  RCCL is loaded
 * on first use; BNPK_ERR_NODEVICE if librccl is not there, BNPK_ERR_HIP + bnpk_last_comm_error() if RCCL reports an error.
 *   bnpk_allreduce_hist         dense histograms (k <= 13): ncclAllReduce(int64, sum) over the bins, in place
 *   bnpk_exchange_counts        every rank tells every rank n_per_peer int64 numbers (how much it will send: per rank, or
 *                               per fine bucket of that rank's key range): h_send_counts[p * n_per_peer + i] goes to rank
 *                               p, h_recv_counts[q * n_per_peer + i] came from rank q.  Synchronous.
 *   bnpk_exchange_by_key_range  sparse histograms: d_send holds the items for rank 0, 1, ... back to back
 *                               (h_send_counts[p] int64 words each — raw hashes grouped by key range, bnpk_kmers_partition's
 *                               output, or the keys / the counts of a local histogram cut at the range boundaries), d_recv
 *                               receives the slices of rank 0, 1, ... back to back (h_recv_counts[q] words, from
 *                               bnpk_exchange_counts).  One grouped ncclSend/ncclRecv step: the N - 1 transfers of a GPU
 *                               run concurrently, one per xGMI link; the rank's own slice is a device copy.  Asynchronous
 *                               on `stream`.
 *   bnpk_exchange_slices        the same step with the slice for rank p at d_send + h_send_offsets[p] (words): what a rank
 *                               sends in ONE of several steps — the keys of the j-th part of every peer's range — lies
 *                               scattered in its partitioned array.  A caller runs step j + 1 on a stream of its own while
 *                               it counts what step j delivered (bionumpy_amd/parallel.py: plan "keys", groups > 1). */
#define BNPK_COMM_ID_BYTES 128
int bnpk_comm_available(void);          /* 1 if RCCL loads in this process (dlopen + dlsym; creates nothing), else 0 */
int bnpk_comm_unique_id(uint8_t* id128);
int bnpk_comm_init(bnpk_ctx* ctx, const uint8_t* id128, int n_ranks, int rank, void** comm_out);
int bnpk_comm_destroy(void* comm);
int bnpk_comm_shape(void* comm, int* n_ranks, int* rank);
const char* bnpk_last_comm_error(void);
int bnpk_allreduce_hist(bnpk_ctx* ctx, void* comm, int64_t* d_hist, int64_t bins, void* stream);
int bnpk_exchange_counts(bnpk_ctx* ctx, void* comm, const int64_t* h_send_counts, int n_per_peer, int64_t* h_recv_counts,
                         void* stream);
int bnpk_exchange_by_key_range(bnpk_ctx* ctx, void* comm, const int64_t* d_send, const int64_t* h_send_counts, int64_t* d_recv,
                               const int64_t* h_recv_counts, void* stream);
int bnpk_exchange_slices(bnpk_ctx* ctx, void* comm, const int64_t* d_send, const int64_t* h_send_offsets, const int64_t* h_send_counts,
                         int64_t* d_recv, const int64_t* h_recv_counts, void* stream);

/* per-kernel hipEvent timers (used by bench.py for the live roofline numbers) */
int bnpk_prof_enable(bnpk_ctx* ctx, int on);
int bnpk_prof_reset(bnpk_ctx* ctx);
int bnpk_prof_count(bnpk_ctx* ctx);                 /* resolves pending events (synchronises) */
int bnpk_prof_get(bnpk_ctx* ctx, int i, char* name64, double* total_ms, int64_t* launches);

/* the rate the device streams at (measurement aid of bench.py): `reps` copies of `bytes` bytes from d_src to
 * d_dst with 16-byte non-temporal accesses, timed with hipEvents; *h_gb_per_s = (bytes read + bytes written) / time.
 * Synchronous. */
int bnpk_copy_peak(bnpk_ctx* ctx, const void* d_src, void* d_dst, int64_t bytes, int reps, double* h_gb_per_s, void* stream);
/* the same copy in four forms, h_gb_per_s4 = {16-byte non-temporal accesses in a grid-stride loop (bnpk_copy_peak), plain
 * accesses in the same loop, one 16-byte element per thread and no loop (the float4 copy MI355X_MICROARCH.md quotes at
 * 6.29 TB/s), four independent elements per thread and iteration}: bench.py states the fastest as what this chip copies at */
int bnpk_copy_rates(bnpk_ctx* ctx, const void* d_src, void* d_dst, int64_t bytes, int reps, double* h_gb_per_s4, void* stream);

/* ---- tuning knobs (tests and experiments; the defaults are what the product path uses) --------
 * "finish_mode": which finishing kernels bnpk_finish_sorted launches — 0 = chosen per call (default): a sample of the
 *                buckets is probed; duplicate-heavy keys take the cascade of duplicate-aware kernels (one wavefront per
 *                bucket, one workgroup per bucket, general kernel — each taking the buckets the one before could not
 *                hold), other keys the fast kernel and, if it gives up, the cascade without its first stage;
 *                1 = the general kernel only; 2 = the fast kernel + redo list (general kernel if it gives up);
 *                3 = the workgroup-per-bucket duplicate-aware kernel (+ general kernel for what it hands back);
 *                4 = the whole cascade; 5 = the fast kernel's ranking with multiplicities and exact output positions
 *                (finish_multi.hip; what mode 0 takes when the fast kernel refuses keys that are nearly all distinct),
 *                + general kernel for buckets with a bin of more than 64 keys; 6 = one workgroup per bucket with a bitonic sort
 *                in LDS (finish_small.hip; what mode 0 takes for histograms of up to 2^25 keys: no dependence on what the keys
 *                look like).  Same results in every mode.
 * "fastq_encoder": the tile kernels of bnpk_fastq_census / bnpk_fastq_encode — 1 = the fast kernels, with the general
 *                ones for the tiles they hand back (default), 0 = the general kernels only.  Same results either way.
 * "index_pairs": bnpk_index_build as ONE partition of (k-mer, row) words where the rows fit a first level's buckets (<= 1024
 *                rows; 1, default) or always as the distinct values of rank(k-mer) * n_rows + row (0).  Same pairs.
 * "sparse_claim": may bnpk_count_sparse take the claiming level (1, default) or only plain levels (0)?  Same results.
 * "l1_ring":     the scatter of bnpk_kmers_partition (the first radix level, fused with k-mer generation) — 0 = the
 *                write-combining scatter that lays every round out anew (default; what the other levels use), 1 = one fixed
 *                128-byte line per bucket in LDS, ranks by atomics, nothing laid out or carried (round 6: bit-identical
 *                buckets, 21.7 ms against 17.1 per 6e9 k-mers — kept for the measurements in NOTES.md).  Same buckets.
 * Unknown names and values out of range return BNPK_ERR_ARG. */
int bnpk_set_option(bnpk_ctx* ctx, const char* name, int64_t value);

/* ---- host staging: pinned buffers and async copies ----------------------------------- */
/* replaces np.frombuffer(file.read(n)) + cp.asanyarray(chunk)
 * (bionumpy/io/parser.py:203-206, bionumpy/cupy_compatible/parser.py:11-17) */
int bnpk_host_alloc(size_t bytes, void** h_out);
int bnpk_host_free(void* h_ptr);
int bnpk_copy_h2d_async(void* d_dst, const void* h_src, size_t bytes, void* stream);
int bnpk_copy_d2h_async(void* h_dst, const void* d_src, size_t bytes, void* stream);
/* file.readinto(buffer) of a plain file for batches of hundreds of megabytes (bionumpy/io/parser.py:203-206): `bytes` bytes
 * from `file_offset` of the open file `fd` into the page-locked h_dst by n_threads threads (pread of disjoint slices, in
 * pieces of piece_bytes); with d_dst != NULL every piece also goes on to d_dst + its offset with hipMemcpyAsync on
 * `stream` as soon as it is read, so the upload runs while the rest is still being read.  *h_read = bytes read (less than
 * `bytes` only at the end of the file).  Synchronous for the reads, asynchronous for the copies. */
int bnpk_pread_parallel(bnpk_ctx* ctx, int fd, int64_t file_offset, void* h_dst, int64_t bytes, int n_threads, int64_t piece_bytes,
                        void* d_dst, void* stream, int64_t* h_read);
/* np.count_nonzero(file[file_offset:file_offset + bytes] == byte) straight from the open file `fd` (n_threads preads):
 * the lines in front of a reader that begins in the middle of a file — what keeps FormatException.line_number and
 * n_lines_read counted from the start of the file (bionumpy/io/parser.py:141-143) when the file is sharded over ranks. */
int bnpk_count_byte_file(int fd, int64_t file_offset, int64_t bytes, uint8_t byte, int n_threads, int64_t* h_count);
int bnpk_stream_sync(void* stream);
/* the first n (<= 4096) words of a device array, on the host: one hipMemcpyAsync into a page-locked mailbox of the ctx behind
 * everything enqueued on `stream`, one hipStreamSynchronize — how the host scalars of the chunk loop (totals, error cells,
 * cut tables) come back; a third of the cost of a pageable copy.  One host thread per ctx. */
int bnpk_fetch_i64(bnpk_ctx* ctx, const int64_t* d_src, int64_t n, int64_t* h_dst, void* stream);

/* ---- A2: newline scan ------------------------------------------------------------------
 * replaces `np.flatnonzero(chunk == NEWLINE)` (bionumpy/io/one_line_buffer.py:63,
 * bionumpy/io/multiline_buffer.py:93) as a two-pass tile census + ordered compaction. */
int64_t bnpk_scan_tiles(int64_t n_bytes);           /* number of tiles the census uses */
/* pass 1: d_tile_offsets[0..tiles] = exclusive scan of per-tile match counts (total at [tiles]) */
int bnpk_byte_census(bnpk_ctx* ctx, const uint8_t* d_buf, int64_t n, uint8_t value,
                     int64_t* d_tile_offsets, void* stream);
/* pass 2: d_pos[j] = position of the j-th byte equal to `value` (ascending), j < limit */
int bnpk_byte_positions(bnpk_ctx* ctx, const uint8_t* d_buf, int64_t n, uint8_t value,
                        const int64_t* d_tile_offsets, int64_t limit, int64_t* d_pos, void* stream);

/* A2 + A3 in one pass: bnpk_byte_positions of '\n' over the first n_lines (a multiple of lines_per_entry) newlines of the
 * chunk AND bnpk_validate_entries' checks — the thread that finds the newline in front of a line looks at the line's first
 * byte.  d_err3 as bnpk_validate_entries writes it, except [2]: bit 0 = one of the first lines_per_entry header lines ends
 * in '\r', bit 1 = the first line is empty (the reference then leaves carriage returns alone: one_line_buffer.py:176-182),
 * i.e. has_cr = (d_err3[2] & 3) == 1. */
int bnpk_line_positions(bnpk_ctx* ctx, const uint8_t* d_buf, int64_t n, const int64_t* d_tile_offsets, int64_t n_lines,
                        int lines_per_entry, uint8_t header, int check_plus, int64_t* d_pos, int64_t* d_err3, void* stream);

/* ---- A3: record validation ------------------------------------------------------------
 * replaces OneLineBuffer._validate + FastQBuffer._validate
 * (bionumpy/io/one_line_buffer.py:156-173, bionumpy/io/fastq_buffer.py:39-45).
 * n_lines must be a multiple of lines_per_entry.  d_err[0] = first entry whose header line does
 * not start with `header` (BNPK_NONE if none; reference line_number = entry*lines_per_entry),
 * d_err[1] = first entry whose third line does not start with '+' (only if check_plus;
 * line_number = 2 + entry*lines_per_entry), d_err[2] = 1 if one of the first lines_per_entry
 * header lines ends in '\r' (_modify_for_carriage_return, one_line_buffer.py:176-182). */
int bnpk_validate_entries(bnpk_ctx* ctx, const uint8_t* d_buf, const int64_t* d_newlines,
                          int64_t n_lines, int lines_per_entry, uint8_t header, int check_plus,
                          int64_t* d_err3, void* stream);

/* ---- A1: the chunks of a reader with small windows, cut out of one big batch ---------------------------
 * replaces the per-chunk loop of NumpyFileReader.read_chunk (bionumpy/io/parser.py:96-171) for a batch that is in HBM
 * already: a chunk is the complete entries inside a window of `window` (= min_chunk_size) bytes that starts at the first
 * unconsumed byte — min_chunk_size more than the bytes held if those are min_chunk_size or more (:117-120) —, a window
 * without a complete entry grows by `window` (:128-131), the file continues behind the last complete entry.
 * d_newlines: the n_lines newline positions of the batch's complete entries (bnpk_byte_positions; n_lines a multiple of
 * lines_per_entry), avail: bytes of the file the batch holds (>= last newline + 1), finished: the file ends with the
 * batch (the last window may be short), first_held: what the reference's reader would hold in front of the batch's first
 * window beyond the batch's first byte ... i.e. (window end of the chunk before) - (its end); max_chunk: 0 or the size
 * beyond which a window without a complete entry is an error.
 * d_out (bnpk_window_cuts_words(max_cuts) words, max_cuts <= 256): [0] chunks cut, [1] 1 = a window grew past max_chunk,
 * [2] window end of the last chunk cut, [3] entries consumed; then per chunk {entries up to and including it, its end
 * byte, 1 if one of its first lines_per_entry header lines ends in '\r' (one_line_buffer.py:176-182), its window end}.
 * A window that reaches past `avail` of an unfinished file is not cut: it belongs to the next batch. */
int64_t bnpk_window_cuts_words(int max_cuts);
int bnpk_window_cuts(bnpk_ctx* ctx, const uint8_t* d_buf, const int64_t* d_newlines, int64_t n_lines, int lines_per_entry,
                     int64_t window, int64_t avail, int finished, int64_t first_held, int64_t max_chunk, int max_cuts,
                     int64_t* d_out, void* stream);
/* d_out[i] = d_newlines[i] - (first byte of the chunk that holds line i), for the lines of the chunks of d_cuts
 * (bnpk_window_cuts' output): every chunk's newline table relative to its own first byte, as a scan of the chunk alone
 * would have given it (np.flatnonzero(chunk == "\n"), one_line_buffer.py:63) */
int bnpk_rebase_lines(bnpk_ctx* ctx, const int64_t* d_newlines, int64_t n_lines, const int64_t* d_cuts, int lines_per_entry,
                      int64_t* d_out, void* stream);

/* ---- A4/A5: field table ----------------------------------------------------------------
 * replaces OneLineBuffer._get_buffer_extractor + TextBufferExtractor.get_field_by_number
 * (bionumpy/io/one_line_buffer.py:140-152, bionumpy/io/file_buffers.py:315-338): start and
 * length of line `field` of every entry (+line_offset on the start, optional '\r' strip). */
int bnpk_field_table(bnpk_ctx* ctx, const uint8_t* d_buf, const int64_t* d_newlines,
                     int64_t n_entries, int lines_per_entry, int field, int line_offset,
                     int strip_cr, int64_t* d_starts, int64_t* d_lens, void* stream);

/* ---- A2-A7 fused for the k-mer pipeline ------------------------------------------------------------------
 * One-line-per-field formats (FASTQ: lines_per_entry 4, two-line FASTA: 2; seq_line = index of the sequence
 * line inside an entry).  Replaces, without materialising the newline table, the field tables or the row
 * offsets: OneLineBuffer.from_raw_buffer + _validate (bionumpy/io/one_line_buffer.py:45-71,156-173,
 * bionumpy/io/fastq_buffer.py:39-45), get_field_by_number(1) (bionumpy/io/file_buffers.py:315-338),
 * EncodedRaggedArray.ravel() + AlphabetEncoding._encode (bionumpy/encodings/alphabet_encoding.py:19-46) and
 * BitArray.pack (bionumpy/sequence/kmers.py:121).
 *   bnpk_fastq_census   reads the text once; synchronous.  h_totals = {newlines in the buffer, lines that take part
 *                       (a multiple of lines_per_entry), sequence bases, 1 if CRs are stripped}.  d_tile_table
 *                       (bnpk_fastq_table_words(n) int64) carries the per-tile state to the encoder.
 *   bnpk_fastq_encode   reads the text once more: d_packed (n_bases/32 + 2 words, the layout of
 *                       bnpk_gather_encode_dna), d_row_ends (n_bases/64 + 2 words: bit i set on the last base of
 *                       every read), d_err3 = {first entry whose header line does not start with `header`, first
 *                       entry whose third line does not start with '+' (check_plus), smallest flat offset of a
 *                       byte that is not A/C/G/T in either case}; BNPK_NONE where there is no error.
 *   bnpk_kmer_starts_from_ends   the ragged trim `ragged[..., :-(k-1)]` (bionumpy/sequence/kmers.py:100) as a
 *                       mask: bit i set iff a k-mer starts at base i (same mask as bnpk_kmer_start_mask);
 *                       *d_count = number of k-mers. */
int64_t bnpk_fastq_tiles(int64_t n_bytes);
int64_t bnpk_fastq_table_words(int64_t n_bytes);
int bnpk_fastq_census(bnpk_ctx* ctx, const uint8_t* d_buf, int64_t n, int lines_per_entry, int seq_line,
                      int64_t* d_tile_table, int64_t* h_totals, void* stream);
int bnpk_fastq_encode(bnpk_ctx* ctx, const uint8_t* d_buf, int64_t n, int lines_per_entry, int seq_line, uint8_t header,
                      int check_plus, const int64_t* d_tile_table, int64_t n_lines_used, int64_t n_bases,
                      uint64_t* d_packed, uint64_t* d_row_ends, int64_t* d_err3, void* stream);
int bnpk_kmer_starts_from_ends(bnpk_ctx* ctx, const uint64_t* d_row_ends, int64_t n_bases, int k, uint64_t* d_starts,
                               int64_t* d_count, void* stream);

/* ---- ragged offsets ----------------------------------------------------------------------
 * replaces npstructures RaggedShape (starts = cumsum(lengths) - lengths):
 * d_offsets[0..n] = exclusive scan of d_lens (d_offsets[n] = total).  If window > 1 the scanned
 * value is max(0, len - (window-1)): the row lengths after `ragged[..., :-(window-1)]`
 * (bionumpy/sequence/kmers.py:100, bionumpy/sequence/rollable.py:66). */
int bnpk_row_offsets(bnpk_ctx* ctx, const int64_t* d_lens, int64_t n, int window,
                     int64_t* d_offsets, void* stream);

/* rows [first_row, first_row + n_rows) of a compact 2-bit packed ragged array (d_packed: n_bases_in bases in the layout of
 * bnpk_gather_encode_dna, d_offsets: its row offsets) as a compact packed array of their own: the bases [first_base,
 * first_base + n_bases) shifted down to bit 0 (n_bases / 32 + 2 words, zero behind the last base) and the rows' n_rows + 1
 * offsets minus first_base.  What `change_encoding(chunk.sequence, DNAEncoding)` amounts to for a chunk that is a run of rows
 * of a batch encoded as a whole (encoded_array.py:655-695 per chunk; bionumpy_amd/io/parser.py cuts the reference's 5 MB
 * chunks out of 128 MB batches). */
int bnpk_packed_rows_slice(bnpk_ctx* ctx, const uint64_t* d_packed, int64_t n_bases_in, const int64_t* d_offsets, int64_t first_row,
                           int64_t n_rows, int64_t first_base, int64_t n_bases, uint64_t* d_out_packed, int64_t* d_out_offsets,
                           void* stream);
/* ---- A6 + A7: ragged gather + ASCII -> DNA code ------------------------------------------
 * replaces EncodedRaggedArray.ravel() of the RaggedView (bionumpy/encoded_array.py:688-690) fused
 * with AlphabetEncoding._encode for 'ACGT' (bionumpy/encodings/alphabet_encoding.py:19-46):
 * A/a C/c G/g T/t -> 0 1 2 3, anything else is an error; *d_err_offset = smallest flat offset of
 * an invalid byte (EncodingError.offset) or BNPK_NONE (caller initialises it to BNPK_NONE via
 * bnpk_fill_i64).  Row r is d_buf[d_starts[r] .. +len) and lands at flat offset d_offsets[r].
 * d_codes (1 byte/base, the reference layout) and d_packed (2 bits/base, 32 bases per uint64,
 * base i at bits 2*(i%32) of word i/32 == npstructures BitArray.pack(.., bit_stride=2),
 * bionumpy/sequence/kmers.py:121) are both optional; d_packed needs total/32 + 2 words.
 * buf_size: bytes of d_buf (the kernel reads up to 31 bytes either side of a row, never outside [0, buf_size)).
 * d_row_ends (optional, total/64 + 2 words): bit i set on the last base of every non-empty row — what bnpk_row_end_mask
 * would compute from d_offsets afterwards with one atomic per row; here the lanes know it as they go. */
int bnpk_gather_encode_dna(bnpk_ctx* ctx, const uint8_t* d_buf, int64_t buf_size, const int64_t* d_starts,
                           const int64_t* d_offsets, int64_t n_rows, int64_t total,
                           uint8_t* d_codes, uint64_t* d_packed, uint64_t* d_row_ends, int64_t* d_err_offset,
                           void* stream);
/* A7 for any alphabet: d_out[i] = h_lut256[d_in[i]] — `self._lookup[byte_array]` of AlphabetEncoding._encode
 * (bionumpy/encodings/alphabet_encoding.py:19-46; the table maps both cases of every letter to its code and
 * everything else to 255).  *d_err_offset = smallest offset of a byte that maps to 255 (EncodingError.offset), or
 * BNPK_NONE (initialised by the caller).  In place (d_out == d_in) is allowed; both 16-byte aligned. */
int bnpk_lut_bytes(bnpk_ctx* ctx, const uint8_t* d_in, int64_t n, const uint8_t* h_lut256, uint8_t* d_out,
                   int64_t* d_err_offset, void* stream);
/* plain ragged gather with an optional constant subtracted (33 -> QualityEncoding,
 * bionumpy/encodings/__init__.py:15-16,26; 0 -> names / raw text) */
int bnpk_gather_rows(bnpk_ctx* ctx, const uint8_t* d_buf, const int64_t* d_starts,
                     const int64_t* d_offsets, int64_t n_rows, int64_t total, int subtract,
                     uint8_t* d_out, void* stream);
/* d_out[i] = d_buf[d_pos[i] + delta]: the strided byte gathers `chunk[new_lines + 1]` of
 * MultiLineFastaBuffer (bionumpy/io/multiline_buffer.py:37-38,93-94) */
int bnpk_take_bytes(bnpk_ctx* ctx, const uint8_t* d_buf, const int64_t* d_pos, int64_t m, int64_t delta,
                    uint8_t* d_out, void* stream);
/* ASCII -> code over a flat array (no gather), same error rule */
int bnpk_encode_dna_flat(bnpk_ctx* ctx, const uint8_t* d_ascii, int64_t n, uint8_t* d_codes,
                         uint64_t* d_packed, int64_t* d_err_offset, void* stream);
/* codes (1 B/base) <-> packed 2-bit words; codes -> ASCII ('ACGT'[code], _decode :48-51) */
int bnpk_pack_codes(bnpk_ctx* ctx, const uint8_t* d_codes, int64_t n, uint64_t* d_packed, void* stream);
int bnpk_unpack_codes(bnpk_ctx* ctx, const uint64_t* d_packed, int64_t n, int to_ascii,
                      uint8_t* d_out, void* stream);

/* ---- A8: 2-bit k-mer hashes ------------------------------------------------------------------
 * replaces _get_dna_kmers + the ragged trim (bionumpy/sequence/kmers.py:90-126): for row r and
 * i < max(0, L_r-k+1): d_hashes[d_out_offsets[r] + i] = sum_{j<k} code[d_in_offsets[r]+i+j] << 2j
 * (first base = least significant, tests/test_kmer.py:85-94).  0 < k < 32.
 * d_out_offsets = bnpk_row_offsets(lens, window=k). */
int bnpk_kmers(bnpk_ctx* ctx, const uint64_t* d_packed, const int64_t* d_in_offsets,
               const int64_t* d_out_offsets, int64_t n_rows, int64_t n_out, int k,
               int64_t* d_hashes, void* stream);

/* ---- per-row reductions of ragged uint8 data --------------------------------------
 * replaces np.sum / np.mean / np.min / np.max(ragged, axis=-1) on the quality scores of a chunk
 * (scripts/small_example.py:36-46; npstructures RaggedArray reductions): row r = d_data[d_offsets[r] .. d_offsets[r+1]).
 * d_sums (int64), d_mins, d_maxs (uint8) get one value per row; any of them may be NULL.  An empty row gives
 * sum 0, min 255, max 0 (the caller raises, as numpy does for a reduction without identity). */
int bnpk_row_reduce_u8(bnpk_ctx* ctx, const uint8_t* d_data, const int64_t* d_offsets, int64_t n_rows,
                       int64_t* d_sums, uint8_t* d_mins, uint8_t* d_maxs, void* stream);
/* the same over rows that were never gathered: row r = d_data[d_starts[r] .. + d_offsets[r+1] - d_offsets[r]) (d_starts NULL:
 * back to back, as above; data_size: bytes of d_data, needed with d_starts), every byte minus `subtract` (0..255, uint8
 * wrap-around) first — np.mean(chunk.quality, axis=1) straight from the text of the chunk: the quality column is a view of it
 * with 33 subtracted (bionumpy/encodings/__init__.py:15-16,26; io/file_buffers.py:426-440), and a reduction does not need
 * the 7.5 GB copy per 50 M reads that materialising it costs. */
int bnpk_row_reduce_u8_view(bnpk_ctx* ctx, const uint8_t* d_data, int64_t data_size, const int64_t* d_starts,
                            const int64_t* d_offsets, int64_t n_rows, int subtract, int64_t* d_sums, uint8_t* d_mins,
                            uint8_t* d_maxs, void* stream);

/* the same for 8-byte elements — int64 (k-mer hashes: Minimizers.__call__ is kmer_hashes.raw().min(axis=-1),
 * bionumpy/sequence/minimizers.py:15-17) or float64 (is_f64: motif scores): sums / mins / maxs of the element type, any of
 * them NULL.  Integer results are exact (wrapping sums, as numpy); float64 sums are added in another order than numpy's
 * pairwise summation (equal up to rounding), min / max are exact and propagate NaN as numpy does.  An empty row gives 0
 * in all three (the caller raises for min / max, as numpy does). */
int bnpk_row_reduce_wide(bnpk_ctx* ctx, const void* d_data, int is_f64, const int64_t* d_offsets, int64_t n_rows, void* d_sums,
                         void* d_mins, void* d_maxs, void* stream);

/* Element-wise helpers of the read filters (scripts/small_example.py:36-46: `np.mean(chunk.quality, axis=1) > 30`,
 * `mask[::3] = False`, `mask1 & mask2`, `chunk[mask]`) on per-row values that stay in HBM (bionumpy_amd/device_vector.py):
 *   bnpk_vec_ratio_rows  d_out[i] = (double)d_sums[i] / (double)(d_offsets[i+1] - d_offsets[i])   — np.mean(ragged, axis=-1)
 *   bnpk_vec_compare     d_out[i] = d_x[i] OP scalar as 0/1; dtype 0 = float64 (scalar_f64), 1 = int64, 2 = uint8
 *                        (scalar_i64); op 0 <, 1 <=, 2 >, 3 >=, 4 ==, 5 != (IEEE: comparisons with nan are false, != true)
 *   bnpk_mask_logic      d_out = d_a AND / OR / XOR d_b (op 0 / 1 / 2) or NOT d_a (op 3, d_b unused); masks are 0/1 bytes
 *   bnpk_mask_fill       d_mask[start + i * step] = value for i < count   — mask[start:stop:step] = value
 *   bnpk_take_i64        d_out[i] = d_arr[d_idx[i]], i < m   — arr[idx]; with the row list of `hist != 0` it turns a dense
 *                        k-mer histogram into the sorted (key, count) form (count_kmers for 9 <= k <= 13)
 * (the set bits of a mask are counted and listed by bnpk_byte_census / bnpk_byte_positions with value 1) */
int bnpk_vec_ratio_rows(bnpk_ctx* ctx, const int64_t* d_sums, const int64_t* d_offsets, int64_t n, double* d_out, void* stream);
int bnpk_vec_compare(bnpk_ctx* ctx, const void* d_x, int64_t n, int dtype, int op, double scalar_f64, int64_t scalar_i64,
                     uint8_t* d_out, void* stream);
int bnpk_mask_logic(bnpk_ctx* ctx, const uint8_t* d_a, const uint8_t* d_b, int64_t n, int op, uint8_t* d_out, void* stream);
int bnpk_mask_fill(bnpk_ctx* ctx, uint8_t* d_mask, int64_t n, int64_t start, int64_t step, int64_t count, int value, void* stream);
int bnpk_take_i64(bnpk_ctx* ctx, const int64_t* d_arr, const int64_t* d_idx, int64_t m, int64_t* d_out, void* stream);

/* The index arithmetic of the write-back path:
 *   bnpk_entry_table     whole entries d_rows[i] of a one-line-per-field buffer: d_starts[i] = the byte behind the newline
 *                        in front of the entry (0 for entry 0), d_lens[i] = through the newline of its last line —
 *                        TextThroughputExtractor's entry_starts / entry_ends (bionumpy/io/file_buffers.py:426-440)
 *   bnpk_join_line_lens  d_lens[r] = bytes of entry r once its lines are written out: per line the field's row length (one
 *                        byte for a line without a field) + prefix[i] header bytes + the newline — the `lengths`
 *                        OneLineBuffer.join_fields sums up (bionumpy/io/one_line_buffer.py:119-134); input of bnpk_join_lines */
int bnpk_entry_table(bnpk_ctx* ctx, const int64_t* d_newlines, int lines_per_entry, const int64_t* d_rows, int64_t m,
                     int64_t* d_starts, int64_t* d_lens, void* stream);
int bnpk_join_line_lens(bnpk_ctx* ctx, int64_t n_rows, int n_lines, const int64_t* const* d_field_offsets, const int* prefix,
                        int64_t* d_lens, void* stream);

/* Per-column sums of ragged uint8 data: np.sum / np.mean(ragged, axis=0) (scripts/small_example.py:20-22,49-52).
 * d_sums[c] = sum over the rows with more than c elements of their element c, d_counts[c] = number of such rows,
 * c < n_cols (= the longest row). */
int bnpk_col_sums_u8(bnpk_ctx* ctx, const uint8_t* d_data, const int64_t* d_offsets, int64_t n_rows, int64_t total,
                     int64_t n_cols, int64_t* d_sums, int64_t* d_counts, void* stream);

/* ---- A13: multi-line FASTA (bionumpy/io/multiline_buffer.py:33-106) ---------------------------------------------
 * bnpk_multiline_cut: `new_entries = np.flatnonzero(chunk[new_lines + 1] == ">")` reduced to what from_raw_buffer
 * (multiline_buffer.py:89-101) needs of it: the index (into d_newlines) of the LAST newline that is followed by the
 * marker (-1 if none: "No complete entry found") and how many there are.  d_newlines: the ordered positions of the
 * newlines of chunk[:-1] (bnpk_byte_positions).
 * bnpk_multiline_table: get_data (multiline_buffer.py:46-62) for the cut chunk d_buf[0, size) and its n_newlines
 * newlines: line i is [newline[i-1] + 1, newline[i]) (the last line ends at size - 1), minus a trailing '\r' if
 * strip_cr (_modify_ends_for_carriage_returns :103-106, the caller probes the first ten lines); a line is a header if
 * it is line 0 or starts with the marker.  Record r: header text d_header_starts/lens[r] (without the marker), its
 * sequence = the concatenation of its sequence lines, d_record_lens[r] bytes; the sequence lines of all records in
 * order: d_seq_line_starts/lens.  All five outputs sized n_newlines + 2; h_totals3 = {records, sequence lines,
 * sequence bytes}. */
int bnpk_multiline_cut(bnpk_ctx* ctx, const uint8_t* d_buf, const int64_t* d_newlines, int64_t n_newlines, uint8_t marker,
                       int64_t* h_last_entry_newline, int64_t* h_n_entry_newlines, void* stream);
int bnpk_multiline_table(bnpk_ctx* ctx, const uint8_t* d_buf, int64_t size, const int64_t* d_newlines, int64_t n_newlines,
                         uint8_t marker, int strip_cr, int64_t* d_header_starts, int64_t* d_header_lens,
                         int64_t* d_record_lens, int64_t* d_seq_line_starts, int64_t* d_seq_line_lens, int64_t* h_totals3,
                         void* stream);
/* MultiLineFastaBuffer.from_data (multiline_buffer.py:68-86): record r = marker, its name, '\n', then its sequence
 * (ASCII, rows of d_seq at d_seq_offsets) in lines of `width` letters (n_characters_per_line = 80), every line ended
 * by '\n'.  d_out_offsets (n_records + 1) receives the byte offset of every record; *h_total the size of the text.
 * Call with d_out == NULL to get the sizes, then with a buffer of *h_total bytes. */
int bnpk_multiline_wrap(bnpk_ctx* ctx, const uint8_t* d_names, const int64_t* d_name_offsets, const uint8_t* d_seq,
                        const int64_t* d_seq_offsets, int64_t n_records, int width, uint8_t marker, int64_t* d_out_offsets,
                        uint8_t* d_out, int64_t out_capacity, int64_t* h_total, void* stream);

/* ---- join_fields: the text of records from their fields ----------------------------
 * replaces OneLineBuffer.join_fields / from_data (bionumpy/io/one_line_buffer.py:99-134, io/fastq_buffer.py:46-61):
 * entry r is its n_lines <= 4 lines; line i = prefix[i] (0 or 1) bytes `header`, row r of field i
 * (d_field_data[i] / d_field_offsets[i], every byte + add[i]: quality scores are written as score + 33), '\n'.
 * d_field_data[i] == NULL: the line is the constant byte fill[i] ('+').  d_entry_offsets (n_rows+1) = exclusive scan
 * of the entry lengths, total = its last entry.  The pointer arrays live on the host.
 * d_field_starts (may be NULL; entries may be NULL): field i was never gathered — its row r lies at
 * d_field_data[i] + d_field_starts[i][r] (a column of the chunk's own text: the name and quality lines of a chunk whose
 * sequence was replaced), its length is still d_field_offsets[i][r+1] - d_field_offsets[i][r]; field_sizes[i] = bytes of
 * d_field_data[i] then. */
int bnpk_join_lines(bnpk_ctx* ctx, int64_t n_rows, int n_lines, const uint8_t* const* d_field_data,
                    const int64_t* const* d_field_offsets, const int64_t* const* d_field_starts, const int64_t* field_sizes,
                    const int* add, const int* prefix, const uint8_t* fill, uint8_t header,
                    const int64_t* d_entry_offsets, int64_t total, uint8_t* d_out, void* stream);

/* ---- reverse complement -----------------------------------------------------------
 * replaces get_reverse_complement = complement(sequence)[..., ::-1] (bionumpy/sequence/dna.py:36-65): every row
 * reversed, every base complemented.  d_offsets (n_rows+1) are the row offsets of the flat input; the output has
 * the same layout.
 *   _packed: 2-bit DNA codes in the bnpk_gather_encode_dna layout (total/32 + 2 words); complement of a code is
 *            3 - code (dna.py:22-27 applied to the alphabet "ACGT").
 *   _bytes:  ASCII (BaseEncoding); the reference's 128-entry table (dna.py:10,29-33): A<->T, C<->G, N->N, every
 *            other byte -> 0. */
int bnpk_reverse_complement_packed(bnpk_ctx* ctx, const uint64_t* d_packed, const int64_t* d_offsets,
                                   int64_t n_rows, int64_t total, uint64_t* d_out, void* stream);
int bnpk_reverse_complement_bytes(bnpk_ctx* ctx, const uint8_t* d_bytes, const int64_t* d_offsets,
                                  int64_t n_rows, int64_t total, uint8_t* d_out, void* stream);
/* _bytes over rows that were never gathered: row r is d_bytes[d_in_starts[r] .. + d_offsets[r+1] - d_offsets[r]) of a buffer
 * of in_size bytes (the sequence column of a text chunk: get_reverse_complement(chunk.sequence)); d_in_starts NULL = _bytes.
 * The output is compact (d_offsets). */
int bnpk_reverse_complement_rows(bnpk_ctx* ctx, const uint8_t* d_bytes, int64_t in_size, const int64_t* d_in_starts,
                                 const int64_t* d_offsets, int64_t n_rows, int64_t total, uint8_t* d_out, void* stream);

/* In place h[i] = min(h[i], rc(h[i])), rc(h) = the hash (layout of bnpk_kmers, first base least significant) of the
 * reverse complement of the k-mer with hash h: strand-independent ("canonical") k-mers.  Not in the reference;
 * defined by the oracle (oracle/kmers.py: canonical_kmers) on top of get_reverse_complement. */
int bnpk_canonical_kmers(bnpk_ctx* ctx, int64_t* d_hashes, int64_t n, int k, void* stream);

/* One bit per base of the flat packed stream, set where a k-mer starts: bits [offsets[r], offsets[r+1]-(k-1)) of
 * every row with at least k bases (d_mask needs total/64 + 2 words) — the ragged trim `ragged[..., :-(k-1)]`
 * (bionumpy/sequence/kmers.py:100) as a mask, so that the fused generator below needs no row lookup. */
int bnpk_kmer_start_mask(bnpk_ctx* ctx, const int64_t* d_offsets, int64_t n_rows, int64_t total, int k,
                         uint64_t* d_mask, void* stream);
/* the read-end mask of rows given by their offsets (bit i set on the last element of every non-empty row; total / 64 + 2
 * words, zeroed here): with bnpk_kmer_starts_from_ends the faster way to the mask bnpk_kmer_start_mask writes. */
int bnpk_row_end_mask(bnpk_ctx* ctx, const int64_t* d_offsets, int64_t n_rows, int64_t total, uint64_t* d_ends, void* stream);

/* Row-lookup-free form of bnpk_kmers (kmers_per_window = 1) and bnpk_minimizers (kmers_per_window =
 * window_size - k + 1 <= 26): d_start_mask marks the flat positions at which a window starts
 * (bnpk_kmer_start_mask with k = the window length in bases); the output is the same ragged-flat array.
 * BNPK_ERR_RANGE if the window holds more k-mers than the fast path covers (use bnpk_minimizers). */
int bnpk_windows_flat(bnpk_ctx* ctx, const uint64_t* d_packed, const uint64_t* d_start_mask, int64_t n_bases, int k,
                      int kmers_per_window, int64_t n_out, int64_t* d_out, void* stream);

/* match_string (bionumpy/sequence/string_matcher.py:16-55): for every window of m symbols, in the
 * ragged-flat order of the windows (rows trimmed by m-1 like the k-mers), 1 if the window equals the pattern, else 0.
 * d_start_mask = bnpk_kmer_start_mask(offsets, k = m).  _packed: 2-bit symbols, pattern given as its hash (layout of
 * bnpk_kmers), m <= 31.  _bytes: any byte sequence, pattern = m <= 64 host bytes. */
int bnpk_match_windows_packed(bnpk_ctx* ctx, const uint64_t* d_packed, const uint64_t* d_start_mask, int64_t n_bases,
                              int m, uint64_t pattern_hash, int64_t n_out, uint8_t* d_out, void* stream);
int bnpk_match_windows_bytes(bnpk_ctx* ctx, const uint8_t* d_bytes, const uint64_t* d_start_mask, int64_t n_bytes,
                             int m, const uint8_t* h_pattern, int64_t n_out, uint8_t* d_out, void* stream);

/* The same reduced per row without the flags: d_counts[r] = windows of row r (bases d_offsets[r] .. d_offsets[r + 1] of the
 * packed words) that equal the pattern — what `match_string(reads, p).sum(axis=-1)` / `.any(axis=-1)` of the reference's
 * callers comes to (string_matcher.py:16-55 returns the flags, ragged reductions follow).  Rows shorter than m count 0. */
int bnpk_match_rows_packed(bnpk_ctx* ctx, const uint64_t* d_packed, int64_t n_bases, const int64_t* d_offsets, int64_t n_rows,
                           int m, uint64_t pattern_hash, int64_t* d_counts, void* stream);

/* Position weight matrix scores (bionumpy/sequence/position_weight_matrix.py:86-104,177-196): for every
 * window of `width` <= 64 bases, in the ragged-flat order of the windows, the double-precision sum
 * ((0 + M[0][c_0]) + M[1][c_1]) + ... — the accumulation order of the reference's calculate_scores, so finite scores
 * are bit-identical.  h_matrix: width x 4 doubles, [position][code]; d_start_mask = bnpk_kmer_start_mask(k = width). */
int bnpk_pwm_scores(bnpk_ctx* ctx, const uint64_t* d_packed, const uint64_t* d_start_mask, int64_t n_bases, int width,
                    const double* h_matrix, int64_t n_out, double* d_out, void* stream);

/* Same hashes as bnpk_kmers, but never materialised in row order: written exactly once, already partitioned (not
 * stably) by the `bits`-bit digit at bit `shift` (bits <= bnpk_radix_max_bits()) — level 1 of the MSD radix
 * partition of the sparse histogram fused into the generation (bionumpy/sequence/kmers.py:121-126 + the
 * np.unique of SURVEY §3.5).  The multiset of values equals bnpk_kmers'; d_child_offsets (2^bits + 1 entries,
 * optional) receives the bucket boundaries, its last entry the number of k-mers.  d_out needs one entry per
 * k-mer (bnpk_row_offsets(lens, window=k) total).  canonical != 0: every hash h is replaced by min(h, rc(h)) as
 * in bnpk_canonical_kmers. */
int bnpk_kmers_partition(bnpk_ctx* ctx, const uint64_t* d_packed, const uint64_t* d_kmer_starts, int64_t n_bases, int k,
                         int canonical, int shift, int bits, int64_t* d_out, int64_t* d_child_offsets, void* stream);

/* A8 for alphabets that are not 4 letters wide: KmerEncoder.__call__ over every window of k codes
 * (bionumpy/sequence/kmers.py:17-27,87; sequence/rollable.py:46-66): hash = sum_j code[p + j] * alphabet_size^j in
 * wrapping int64 arithmetic (numpy's uint8 windows .dot(int64 weights)), trimmed per row like bnpk_kmers.
 * d_codes: 1 byte per letter, rows at d_in_offsets. */
int bnpk_kmers_generic(bnpk_ctx* ctx, const uint8_t* d_codes, const int64_t* d_in_offsets, const int64_t* d_out_offsets,
                       int64_t n_rows, int64_t n_out, int k, int alphabet_size, int64_t* d_hashes, void* stream);
/* ---- A11: minimizers -----------------------------------------------------------------------------
 * replaces get_minimizers / Minimizers.__call__ (bionumpy/sequence/minimizers.py:8-54): for every
 * window of `window_size` bases the minimum raw hash of its window_size-k+1 k-mers.
 * d_out_offsets = bnpk_row_offsets(lens, window=window_size). */
int bnpk_minimizers(bnpk_ctx* ctx, const uint64_t* d_packed, const int64_t* d_in_offsets,
                    const int64_t* d_out_offsets, int64_t n_rows, int64_t n_out, int k,
                    int window_size, int64_t* d_out, void* stream);
/* the same for alphabets that are not 4 letters wide (get_minimizers accepts any AlphabetEncoding,
 * bionumpy/sequence/minimizers.py:48-52): the smallest — as signed int64, which is how numpy's .min compares the wrapped
 * hashes of wide alphabets — of the window_size-k+1 hashes bnpk_kmers_generic gives for the window's k-mers.
 * d_out_offsets = bnpk_row_offsets(lens, window = window_size). */
int bnpk_minimizers_generic(bnpk_ctx* ctx, const uint8_t* d_codes, const int64_t* d_in_offsets, const int64_t* d_out_offsets,
                            int64_t n_rows, int64_t n_out, int k, int window_size, int alphabet_size, int64_t* d_out,
                            void* stream);

/* count_encoded / np.bincount over the uint8 CODES of an encoded array (bionumpy/sequence/count_encoded.py:166-182;
 * encoded_array.py:463-466: bincount) without widening them to int64: d_hist[b] += #{i < n: d_values[i] == b} for b < n_bins
 * (<= 256; other bytes are not counted).  bnpk_count_packed2: the same for DNA packed 2 bits per base (the BitArray layout of
 * bnpk_gather_encode_dna), four bins.  bnpk_count_bytes_rows: one histogram per row of ragged codes (axis=-1), n_bins <= 8,
 * d_hist[r * n_bins + b] written (not added to). */
int bnpk_count_bytes(bnpk_ctx* ctx, const uint8_t* d_values, int64_t n, int n_bins, int64_t* d_hist, void* stream);
int bnpk_count_packed2(bnpk_ctx* ctx, const uint64_t* d_packed, int64_t n_bases, int64_t* d_hist4, void* stream);
int bnpk_count_bytes_rows(bnpk_ctx* ctx, const uint8_t* d_values, const int64_t* d_offsets, int64_t n_rows, int64_t total,
                          int n_bins, int64_t* d_hist, void* stream);
/* ---- A9: counting ---------------------------------------------------------------------------------
 * dense: replaces np.bincount(values, minlength=4^k) of count_encoded
 * (bionumpy/sequence/count_encoded.py:166-177); d_hist (n_bins int64) is ACCUMULATED into, so the
 * sum over chunks (EncodedCounts.__add__, count_encoded.py:41-55) is the same buffer. */
int bnpk_count_dense(bnpk_ctx* ctx, const int64_t* d_values, int64_t n, int64_t n_bins,
                     int64_t* d_hist, void* stream);
/* per-row dense counts (count_encoded(axis=-1) on a ragged array, count_encoded.py:178-182):
 * d_hist is [n_rows, n_bins] int64, zero-initialised by the caller */
int bnpk_count_dense_rows(bnpk_ctx* ctx, const int64_t* d_values, const int64_t* d_offsets,
                          int64_t n_rows, int64_t n, int64_t n_bins, int64_t* d_hist, void* stream);

/* weighted counts: np.bincount(values, weights=w, minlength=n_bins) as count_encoded(values, weights) calls it
 * (bionumpy/sequence/count_encoded.py:166-187) — 1-D weights over flat values (n_rows = 1), 2-D weights over flat values
 * ("for row in weights": value_stride = 0, weight_stride = n), or 1-D weights over the rows of a matrix of values ("for
 * row in values": value_stride = n, weight_stride = 0):
 *     d_hist[r * n_bins + d_values[r * value_stride + i]] += d_weights[r * weight_stride + i],  r < n_rows, i < n.
 * weights_f64 = 0: int64 weights and an int64 histogram (exact: what numpy's double accumulation of integer weights gives
 * below 2^53), 1: float64 weights and a float64 histogram — accumulated with atomics, i.e. in another order than numpy's
 * left-to-right loop.  BIT-EQUAL to numpy: integer and bool weights (weights_f64 = 0), and float64 weights whose partial sums
 * per bin are all exactly representable (integers below 2^53, dyadic fractions of bounded size — every additions is exact,
 * so their order does not matter).  Any other float64 weights: equal up to the rounding of the additions, |difference| <=
 * (items in the bin) x eps x sum |w| (the tests check exactly representable weights with tolerance 0 and random ones with
 * rtol 1e-12).  d_hist is accumulated
 * into (zero it first).  *h_out_of_range = 1 if a value was not in [0, n_bins) (it was skipped; numpy would have grown
 * the histogram or raised).  Synchronous. */
int bnpk_count_weighted(bnpk_ctx* ctx, const int64_t* d_values, const void* d_weights, int weights_f64, int64_t n,
                        int64_t n_rows, int64_t value_stride, int64_t weight_stride, int64_t n_bins, void* d_hist,
                        int* h_out_of_range, void* stream);

/* sparse (k > 8 has no reference implementation; defined as np.unique(h, return_counts=True),
 * SURVEY.md §3.5): step 1 sorts the keys (d_alt is an n-element ping-pong buffer; only bits
 * [begin_bit, end_bit) take part, stably — begin_bit > 0 is the key-range partition of the
 * multi-GPU exchange; *h_in_alt = 1 if the sorted keys ended up in d_alt, known on the host
 * without a synchronisation), step 2 counts distinct keys (synchronous: returns the count
 * through h_n_unique), step 3 writes sorted unique keys + run lengths. */
int bnpk_sort_keys(bnpk_ctx* ctx, int64_t* d_keys, int64_t* d_alt, int64_t n, int begin_bit,
                   int end_bit, int* h_in_alt, void* stream);
int bnpk_sort_pairs(bnpk_ctx* ctx, int64_t* d_keys, int64_t* d_keys_alt, int64_t* d_vals,
                    int64_t* d_vals_alt, int64_t n, int key_bits, int* h_in_alt, void* stream);
/* Hand-written path of the same operation (the one the pipeline uses; the rocPRIM sort + run kernels are the
 * fallback for inputs with heavy-hitter buckets):
 *   bnpk_radix_partition  one MSD level: the keys of every segment [d_seg_offsets[p], d_seg_offsets[p+1]) are
 *                         partitioned (not stably) by the `bits`-bit digit at bit `shift` with an LDS
 *                         write-combining scatter (whole 128-byte lines only); d_child_offsets gets the
 *                         n_seg * 2^bits + 1 boundaries of the child buckets, which are the segments of the next
 *                         level.  d_seg_offsets may be NULL when n_seg == 1.  d_out must not alias d_keys.
 *                         Keys must be < 2^63; n < 2^35.
 *   bnpk_finish_sorted    every bucket [d_bucket_offsets[b], d_bucket_offsets[b+1]) (keys equal above bit
 *                         `low_bits`, at most bnpk_finish_capacity() of them, any order) is sorted in LDS,
 *                         run-length-counted and written as sorted distinct keys + multiplicities.  Synchronous:
 *                         returns the number of distinct keys.  Buckets over the capacity (heavy-hitter keys) must
 *                         have been counted by the caller beforehand: d_big_table holds n_big {bucket, number of
 *                         distinct keys, offset into d_big_keys / d_big_counts} int64 triples sorted by bucket;
 *                         their pairs are copied into place.  *h_overflow != 0: the outputs must be discarded
 *                         (bnpk_sort_keys + run kernels) — bit 0: a bucket exceeded the capacity without being
 *                         listed (the caller's mistake); bit 1: a kernel that waits for other workgroups' results
 *                         gave up waiting (a run-time condition — e.g. the CUs were shared with another stream's
 *                         kernels; d_part is intact in that case and can be sorted instead).
 *                         d_state needs bnpk_finish_state_words(n_buckets) int64.  d_part is WORKSPACE: its contents
 *                         are undefined afterwards (the duplicate-aware kernel writes every bucket's distinct keys
 *                         back over the bucket's own keys before they are moved into place). */
int64_t bnpk_radix_max_bits(void);
int64_t bnpk_finish_capacity(void);
/* bnpk_finish_sorted over buckets of fixed stride (bnpk_radix_partition_claimed + bnpk_claimed_finalize): bucket b's keys lie
 * at d_part + b * part_stride, d_bucket_offsets are the offsets of the equivalent dense layout; part_stride 0 = bnpk_finish_sorted */
int bnpk_finish_sorted_strided(bnpk_ctx* ctx, int64_t* d_part, int64_t n, int64_t part_stride, const int64_t* d_bucket_offsets,
                               int64_t n_buckets, int low_bits, int64_t* d_keys_out, int64_t* d_counts_out, int64_t* d_state,
                               const int64_t* d_big_table, int n_big, const int64_t* d_big_keys, const int64_t* d_big_counts,
                               int64_t* h_n_unique, int* h_overflow, void* stream);
int bnpk_radix_partition(bnpk_ctx* ctx, const int64_t* d_keys, int64_t n, const int64_t* d_seg_offsets, int64_t n_seg,
                         int shift, int bits, int64_t* d_out, int64_t* d_child_offsets, void* stream);
/* The same level WITHOUT its histogram pass (round 5: the second level of the 31-mer path — 48 GB that were read only to
 * count digits).  Every child bucket c = segment * 2^bits + digit owns bnpk_claimed_stride() slots of d_buckets: whole 128-byte
 * lines at the front, [0, bnpk_claimed_cap_lo()), claimed by the workgroups as they flush them (atomicAdd on d_fill[2c]); the
 * < 16 leftover keys of every slab behind them (d_fill[2c + 1]).  A bucket's keys are the two dense runs
 * [0, min(d_fill[2c], cap_lo)) and [cap_lo, cap_lo + min(d_fill[2c+1], stride - cap_lo)) of its slots; keys without a place
 * (a bucket over the finishing kernels' capacity, a segment of very many slabs) go to d_bag — *d_bag_fill of them, an
 * unordered multiset the caller counts on its own and merges in; *d_bag_fill > bag_cap: keys were dropped, use
 * bnpk_radix_partition.  bnpk_claimed_finalize moves every bucket's leftover keys right behind its lines (one dense run per
 * bucket from then on, at slot c * stride) and turns d_fill into the n_buckets + 1 offsets of the equivalent dense layout
 * (sizes, output positions); bnpk_finish_sorted_strided reads the buckets where they lie. */
int64_t bnpk_claimed_stride(void);
int64_t bnpk_claimed_cap_lo(void);
int bnpk_radix_partition_claimed(bnpk_ctx* ctx, const int64_t* d_keys, int64_t n, const int64_t* d_seg_offsets, int64_t n_seg,
                                 int shift, int bits, int64_t* d_buckets, uint32_t* d_fill, int64_t* d_bag, int64_t bag_cap,
                                 int64_t* d_bag_fill, void* stream);
int bnpk_claimed_finalize(bnpk_ctx* ctx, int64_t* d_buckets, const uint32_t* d_fill, int64_t n_buckets, int64_t* d_bucket_offsets,
                          void* stream);
/* one more MSD level over MANY SMALL segments (each at most bnpk_radix_small_capacity() keys, bits <= 4): one
 * workgroup takes a whole segment, ranks with wave ballots and writes it back as one contiguous run.  Same
 * outputs as bnpk_radix_partition; BNPK_ERR_RANGE if a segment was larger (outputs then invalid). */
int64_t bnpk_radix_small_capacity(void);
int bnpk_radix_partition_small(bnpk_ctx* ctx, const int64_t* d_keys, int64_t n, const int64_t* d_seg_offsets,
                               int64_t n_seg, int shift, int bits, int64_t* d_out, int64_t* d_child_offsets,
                               void* stream);
/* what the caller of bnpk_finish_sorted decides on, in one pass over the bucket offsets and ONE download: d_out[0] = keys of
 * the largest bucket, d_out[1] = buckets of more than `cap` keys, then {bucket, its first key's index, its keys} of up to
 * max_list (<= 4096) of those in no particular order (all of them if d_out[1] <= max_list).  d_out: 2 + 3 * max_list words.
 * (replaces sizes.max() / (sizes > cap).nonzero() on the offsets: two reductions and two downloads) */
int bnpk_bucket_census(bnpk_ctx* ctx, const int64_t* d_bucket_offsets, int64_t n_buckets, int64_t cap, int max_list, int64_t* d_out,
                       void* stream);
int64_t bnpk_finish_state_words(int64_t n_buckets);
int bnpk_finish_sorted(bnpk_ctx* ctx, int64_t* d_part, int64_t n, const int64_t* d_bucket_offsets,
                       int64_t n_buckets, int low_bits, int64_t* d_keys_out, int64_t* d_counts_out, int64_t* d_state,
                       const int64_t* d_big_table, int n_big, const int64_t* d_big_keys, const int64_t* d_big_counts,
                       int64_t* h_n_unique, int* h_overflow, void* stream);
/* ---- the sparse histogram as ONE call (round 6) ----------------------------------------------------------
 * np.unique(keys, return_counts=True) for keys < 2^key_bits: d_keys_out / d_counts_out (n entries each, neither the input)
 * receive the sorted distinct keys and their counts, *h_n_unique how many.  Replaces the k > 13 branch the reference does not
 * have (bionumpy/sequence/count_encoded.py:150-188 counts k <= 8 densely; SURVEY §3.5 defines the rest as np.unique) and the
 * planner that lived in Python through round 5: levels planned for buckets of ~6500 keys, the last one as the claiming level
 * where the workspace has room for its slots, one census of the real bucket sizes, up to two more levels or up to 1024 heavy
 * buckets counted on their own, the finishing kernels; the library sort only for what none of that takes.
 *   d_keys          CONSUMED (workspace afterwards)
 *   skip_bits       leading bits of the key_bits that all keys share (the key range a rank owns after the exchange), else 0
 *   n_plan          keys to plan the levels for if the keys are not spread evenly (canonical k-mers: 2 n), 0 = n
 *   d_part_offsets  2^part_bits + 1 offsets if d_keys is already grouped by its top part_bits bits (bnpk_kmers_partition), else
 *                   NULL / 0
 *   d_work          work_bytes bytes of device memory.  bnpk_count_sparse_workspace(n, key_bits, skip_bits, n_plan, part_bits, mode) says how many:
 *                   mode 2 = enough for ANY input (about 6 n words: heavy-hitter buckets are counted in a batch of their own, the
 *                   library sort takes what nothing else does); mode 1 = the claiming level's slots (n_buckets * 7680 keys,
 *                   ~1.4 n words: what well-spread keys take, the fastest path); mode 0 = plain levels only (~n words).  The
 *                   call uses what it is given: the claiming level iff its slots fit, BNPK_ERR_NOMEM if the input needs a
 *                   path the workspace has no room for (d_keys is lost then)
 *   h_info5         optional: {path (1 claiming level, 2 plain levels, 3 library sort), levels run, host round trips, keys in
 *                   the bag, heavy buckets counted in a batch of their own}
 * Synchronous (the number of distinct keys is an answer). */
int64_t bnpk_count_sparse_workspace(int64_t n, int key_bits, int skip_bits, int64_t n_plan, int part_bits, int mode);
int bnpk_count_sparse(bnpk_ctx* ctx, int64_t* d_keys, int64_t n, int key_bits, int skip_bits, int64_t n_plan,
                      const int64_t* d_part_offsets, int part_bits, void* d_work, int64_t work_bytes, int64_t* d_keys_out,
                      int64_t* d_counts_out, int64_t* h_n_unique, int64_t* h_info5, void* stream);
/* KmerIndex.create_index (bionumpy/sequence/indexing/kmer_indexing.py:24-47) as ONE call: the sorted distinct (k-mer, row)
 * pairs of n (k-mer, row) pairs, k-mers < 2^key_bits (key_bits <= 62), rows in [0, n_rows).  No key-value sort: the distinct
 * k-mers (bnpk_count_sparse), every k-mer's rank among them (a binary search narrowed by a prefix table to ~8 keys), the
 * distinct values of rank * n_rows + row (bnpk_count_sparse again: they order like the pairs), split back.  The inputs are
 * left alone.  d_keys_out / d_rows_out: n entries each; d_counts_out (optional, n entries): how often every pair occurred;
 * *h_n_pairs: the number of distinct pairs.  BNPK_ERR_RANGE if distinct k-mers * n_rows does not fit 62 bits. */
int64_t bnpk_index_build_workspace(int64_t n, int key_bits, int64_t n_rows);
int bnpk_index_build(bnpk_ctx* ctx, const int64_t* d_kmers, const int64_t* d_rows, int64_t n, int key_bits, int64_t n_rows,
                     void* d_work, int64_t work_bytes, int64_t* d_keys_out, int64_t* d_rows_out, int64_t* d_counts_out,
                     int64_t* h_n_pairs, void* stream);

/* A10 for sparse histograms: the sum of two (sorted distinct keys, counts) lists — EncodedCounts.__add__
 * (bionumpy/sequence/count_encoded.py:38-48) as `streamable(sum)` folds it over the chunks of a file — by a merge along
 * the merge path: keys of either list are copied, counts of keys both lists hold are added.  d_out_* need na + nb
 * entries; *h_n_out = number of distinct keys of the sum (synchronous). */
int bnpk_merge_add(bnpk_ctx* ctx, const int64_t* d_a_keys, const int64_t* d_a_counts, int64_t na, const int64_t* d_b_keys,
                   const int64_t* d_b_counts, int64_t nb, int64_t* d_out_keys, int64_t* d_out_counts, int64_t* h_n_out,
                   void* stream);
/* A12, KmerIndex.create_index (bionumpy/sequence/indexing/kmer_indexing.py:24-47) without a key-value sort: with
 * rank[i] = index of k-mer i among the sorted distinct k-mers (bnpk_search_sorted), id[i] = rank[i] * n_rows + row[i]
 * orders like (kmer, row); the distinct ids (the sparse counting path again) are split back into
 * d_keys_out[j] = d_sorted_keys[id / n_rows], d_rows_out[j] = id % n_rows. */
int bnpk_pair_compose(bnpk_ctx* ctx, const int64_t* d_rank, const int64_t* d_rows, int64_t n, int64_t n_rows, int64_t* d_ids,
                      void* stream);
int bnpk_pair_split(bnpk_ctx* ctx, const int64_t* d_ids, int64_t n, int64_t n_rows, const int64_t* d_sorted_keys,
                    int64_t* d_keys_out, int64_t* d_rows_out, void* stream);
/* d_tile_offsets needs bnpk_run_tiles(n)+1 entries */
int64_t bnpk_run_tiles(int64_t n);
int bnpk_run_census(bnpk_ctx* ctx, const int64_t* d_sorted, const int64_t* d_second, int64_t n,
                    int64_t* d_tile_offsets, int64_t* h_n_runs, void* stream);
/* d_run_starts[j] = index of the first element of run j (n_runs+1 entries, last = n);
 * d_keys_out[j] = d_sorted[start]; optional d_second_out[j] = d_second[start] */
int bnpk_run_heads(bnpk_ctx* ctx, const int64_t* d_sorted, const int64_t* d_second, int64_t n,
                   const int64_t* d_tile_offsets, int64_t n_runs, int64_t* d_keys_out,
                   int64_t* d_second_out, int64_t* d_run_starts, void* stream);
/* d_counts[j] = sum of weights over run j (d_weight_prefix = exclusive scan of the weights,
 * n+1 entries) or, with d_weight_prefix NULL, the run length */
int bnpk_run_sums(bnpk_ctx* ctx, const int64_t* d_run_starts, int64_t n_runs,
                  const int64_t* d_weight_prefix, int64_t* d_counts, void* stream);
int bnpk_exclusive_scan_i64(bnpk_ctx* ctx, const int64_t* d_in, int64_t n, int64_t* d_out, void* stream);

/* ---- A12: k-mer index helpers ------------------------------------------------------------------------
 * row id of every flat element of a ragged array (np.repeat(arange(n_rows), lens)); with the sorted
 * (kmer,row) pairs, run census on pairs and lower/upper bound this rebuilds
 * KmerIndex.create_index / get_indices (bionumpy/sequence/indexing/kmer_indexing.py:24-55). */
int bnpk_row_ids(bnpk_ctx* ctx, const int64_t* d_offsets, int64_t n_rows, int64_t n,
                 int64_t* d_rows, void* stream);
/* d_out[i] = first index j with d_sorted[j] >= q (upper=0) or > q (upper=1) */
int bnpk_search_sorted(bnpk_ctx* ctx, const int64_t* d_sorted, int64_t n, const int64_t* d_queries,
                       int64_t m, int upper, int64_t* d_out, void* stream);

/* ---- utilities ------------------------------------------------------------------------------------------ */
int bnpk_fill_i64(bnpk_ctx* ctx, int64_t* d_ptr, int64_t n, int64_t value, void* stream);
/* synthetic FASTQ (bench/test input; record = "@%010d\n" + L bases + "\n+\n" + L*'I' + "\n").
 * mode 0: bases i.i.d. uniform (the reference's own generator, benchmarks/rules/simulation.smk:3-11);
 * mode 1: reads sampled from a fixed pseudo-random genome of genome_len bases.
 * Bytes are a pure function of (seed, read id, position) — see bionumpy_amd/synth.py. */
int64_t bnpk_synth_record_bytes(int read_len);
int bnpk_synth_fastq(bnpk_ctx* ctx, uint8_t* d_out, int64_t first_read, int64_t n_reads, int read_len,
                     uint64_t seed, int mode, int64_t genome_len, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BNPK_H */
