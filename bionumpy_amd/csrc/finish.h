// Internal: what the finishing kernels of the sparse k-mer histogram share (finish.hip, finish_dup.hip).
// pstride (every launcher): 0 = the buckets lie back to back at bucket_off[b]; otherwise bucket b's keys lie at
// part + b * pstride (radix.hip: buckets of fixed stride, claimed line by line) and bucket_off only says how many they are /
// where the counts of the loose convention and the output go.
#pragma once
#include "common.h"

// d_state header words.  [0] flags (1 = over-capacity bucket without a pre-counted entry, 2 = the general kernel's
// look-back gave up, 4 = too many buckets with duplicates for the fast kernel), [1] ticket counter of the general
// kernel, [2] distinct keys, [3] redo list length, [4] announcements of the fast kernel, [5] buckets only the general
// kernel holds, [6] length of the list of buckets the wavefront kernel left to the workgroup kernel; the fast kernel's ticket counter has a 128-byte line of its own ([96, 112)); the per-bucket arrays
// start at word FS_FAST.
constexpr int FS_FLAGS = 0, FS_TICKET = 1, FS_UNIQUE = 2, FS_REDO = 3, FS_NLOG = 4, FS_MISFIT = 5, FS_TODO = 6, FS_LOG = 8;
// (the duplicate-aware path does not use the fast kernel's log: its words hold the probe's counters)
constexpr int FS_PROBE_BAD = 8, FS_PROBE_DISTINCT = 9, FS_PROBE_KEYS = 10, FS_PROBE_GIVEUP = 11;   // [11] keys the sampled buckets that did not fit had shown when that became clear
constexpr int FS_FTICKET = 96, FS_FAST = 112;

// most keys a bucket may hold (the general kernel and the duplicate-aware kernel; the fast kernel takes 7680)
constexpr int FINISH_CAP = 8192;

// Duplicate-aware finishing (finish_dup.hip).  `part` is read AND overwritten: every bucket's sorted distinct keys are
// first written back over the bucket's own keys (a bucket never has more distinct keys than keys), their counts to the
// same positions of `loose_counts`, and D[b] = its number of distinct keys to Dv[b]; buckets the kernel gives up on
// (more distinct keys than its table holds, long probe sequences) are appended to redo_ids / header[FS_REDO] with
// Dv[b] = 0 for the general kernel to finish the same way.  todo_ids (may be NULL): only the buckets of this list, whose
// length is header[FS_TODO] (the wavefront kernel's hand-backs).
int bnpk_finish_dup_launch(bnpk_ctx* ctx, uint64_t* part, const int64_t* bucket_off, int64_t n_buckets, int low_bits,
                           unsigned long long* header, int64_t* Dv, unsigned* redo_ids, int64_t* loose_counts,
                           const int64_t* big_table, int n_big, const uint64_t* big_keys, const int64_t* big_counts,
                           const unsigned* todo_ids, int64_t pstride, hipStream_t s);
// dst[T[b] + i] = src[bucket_off[b] + i] for i < T[b + 1] - T[b]: the loose per-bucket runs moved to their final place
// (src and dst must be different buffers); header[FS_UNIQUE] = T[n_buckets].
// (pstride != 0: bucket b's run in src starts at b * pstride instead of bucket_off[b])
int bnpk_finish_compact_launch(bnpk_ctx* ctx, const int64_t* src, int64_t* dst, const int64_t* bucket_off, const int64_t* T,
                               int64_t n_buckets, unsigned long long* header, int64_t pstride, hipStream_t s);

// One wavefront per bucket, a 704-slot table each (finish_wave.hip): see there.
int bnpk_finish_wave_launch(bnpk_ctx* ctx, bool probe, int64_t probe_buckets, uint64_t* part, int64_t n, const int64_t* bucket_off,
                            int64_t n_buckets, int low_bits, unsigned long long* header, int64_t* Dv, unsigned* todo_ids,
                            int64_t* loose_counts, const int64_t* big_table, int n_big, const uint64_t* big_keys,
                            const int64_t* big_counts, int64_t pstride, hipStream_t s);

// The fast kernel's ranking with multiplicities, run-length emission from the sorted stage and exact output positions
// (finish_multi.hip): nearly-distinct keys with a repeat in most buckets.  status: n_buckets zeroed 32-bit words; the
// header's first FS_FAST words zeroed.  Buckets it leaves to finish_sorted_kernel<REDO> (a bin of more than 64 keys) are
// listed in redo_ids / redo_bases (header[FS_REDO]); header[FS_UNIQUE] = distinct keys; flag 2 = a wait gave up (the
// output is incomplete: the caller takes another kernel).  Buckets over 7680 keys must be pre-counted (big_table).
// bytes of the parking ring for `grid` workgroups: taken from the ctx's scratch arena by the launcher (a caller that holds a
// pointer into the arena asks for at least this much beforehand, so that the arena does not move under it)
int64_t bnpk_finish_multi_park_bytes(int grid);
int bnpk_finish_multi_launch(bnpk_ctx* ctx, const uint64_t* part, const int64_t* bucket_off, int64_t n_buckets, int low_bits,
                             unsigned long long* header, unsigned* status, uint64_t* keys_out, int64_t* counts_out,
                             const int64_t* big_table, int n_big, const uint64_t* big_keys, const int64_t* big_counts,
                             unsigned* redo_ids, int64_t* redo_bases, int64_t pstride, hipStream_t s);

// One workgroup per bucket, a bitonic sort in LDS (finish_small.hip): small histograms (up to 2^25 keys), any key distribution;
// the duplicate-aware kernels' convention (distinct keys back over the bucket, counts to loose_counts, D to Dv); buckets over
// FINISH_CAP keys must be pre-counted (big_table) or set flag 1.
int bnpk_finish_bitonic_launch(bnpk_ctx* ctx, uint64_t* part, const int64_t* bucket_off, int64_t n_buckets, unsigned long long* header,
                               int64_t* Dv, int64_t* loose_counts, const int64_t* big_table, int n_big, const uint64_t* big_keys,
                               const int64_t* big_counts, int64_t pstride, hipStream_t s);
int bnpk_finish_bitonic_probe_launch(bnpk_ctx* ctx, const uint64_t* part, const int64_t* bucket_off, int64_t n_buckets,
                                     int64_t probe_buckets, unsigned long long* header, int64_t pstride, hipStream_t s);
