// The planners of the sparse histogram and of the k-mer index as single C-ABI calls (round 6; SURVEY §8b lists
// bnpk_count_sparse and bnpk_index_build as the entry points a caller binds).
//
//   bnpk_count_sparse   np.unique(keys, return_counts=True) (A9 for k > 13, SURVEY §3.5: the reference has no implementation):
//                       plans the MSD levels for the buckets the finishing kernels take, runs them (the last one without its
//                       histogram pass where the workspace allows: radix.hip's claiming level + the bag), looks at the real
//                       bucket sizes (one census, ONE download), adds up to two levels or pre-counts a few heavy buckets,
//                       finishes; whatever cannot be taken that way is sorted (the heavy-hitter fall-back).  Rounds 1-5 had this
//                       in Python (ops.count_sparse: 200 lines around a dozen entry points); a C or Cython caller had to
//                       rewrite it.  The workspace comes from the caller (bnpk_count_sparse_workspace), the function
//                       synchronises where an answer decides the next launch: the bag's fill (claiming level), the census
//                       (plain level), the number of distinct keys (always) — two or three round trips per call.
//   bnpk_index_build    KmerIndex.create_index (bionumpy/sequence/indexing/kmer_indexing.py:24-47): the sorted distinct
//                       (k-mer, row) pairs WITHOUT a key-value sort: distinct k-mers (one sparse count), every k-mer's rank
//                       among them (a prefix table of up to 2^22 entries narrows the binary search to a cache line or two), the distinct
//                       values of rank * n_rows + row (a second sparse count), split back into (k-mer, row).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "common.h"

// radix.hip: the first level of the index's partition ((k-mer, row) pairs -> words grouped by the k-mer's top bits)
int bnpk_pairs_partition_launch(bnpk_ctx* ctx, const int64_t* d_keys, const int64_t* d_rows, int64_t n, int key_bits, int bits,
                                int row_bits, int64_t* d_words, int64_t* d_child_off, unsigned* d_tags, int64_t* d_list,
                                int64_t* d_list_n, hipStream_t s);

#include <cstdio>
#include <cstdlib>

namespace {

// BNPK_SPARSE_DEBUG=1: which request the workspace could not serve, on stderr
int nomem_at(int line, size_t used, size_t size) {
  static const bool on = getenv("BNPK_SPARSE_DEBUG") != nullptr;
  if (on) fprintf(stderr, "bnpk sparse.hip:%d: workspace exhausted (%zu of %zu bytes used)\n", line, used, size);
  return BNPK_ERR_NOMEM;
}
#define SP_NOMEM(arena) nomem_at(__LINE__, (arena).used, (arena).size)

// average bucket the plan aims for.  The fast finishing kernels and the claiming level's buckets take 7680 keys (7552 + the slabs'
// leftovers); random keys spread with sigma = sqrt(7000) = 84, so 7000 leaves 6.6 sigma, and keys that repeat overflow into the
// bag / the batch of over-full buckets as they do at any target.  (6000 until round 4, 6500 until round 6: 57 M reads x 150 bp are
// 6.84e9 31-mers = 6523 per bucket at 20 bits, and the 21st bit cost 128 ms against 88: 2^21 half-empty buckets, no claiming level.)
constexpr int64_t FINISH_TARGET = 7000;
constexpr int MAX_PRECOUNTED = 1024;      // buckets over the finishing capacity that are counted in a batch of their own (round 5: 256, by the library sort)
constexpr int64_t CLAIM_MIN_KEYS = 1ll << 20;

struct arena_t {
  char* base = nullptr;
  size_t size = 0, used = 0;
  void* take(size_t bytes) {
    const size_t at = (used + 255) & ~(size_t)255;
    if (at + bytes > size) return nullptr;
    used = at + bytes;
    return base + at;
  }
  int64_t* words(int64_t n) { return reinterpret_cast<int64_t*>(take((size_t)std::max<int64_t>(n, 1) * 8)); }
  size_t left() const { return size - std::min(size, (used + 255) & ~(size_t)255); }
};

// digit widths of the MSD levels still to run so that the buckets average <= FINISH_TARGET keys, the top `done` bits resolved
int radix_plan(int64_t n, int key_bits, int done, int levels[8]) {
  int need = 0;
  while (need < key_bits && (n >> need) > FINISH_TARGET) ++need;
  const int rest = std::max(0, need - done);
  if (rest == 0) return 0;
  int max_bits = 10;                                   // 10-bit digits flush whole 128-byte lines ...
  if ((rest + 10) / 11 < (rest + 9) / 10) max_bits = 11;   // ... but an 11-bit digit is cheaper than one more level
  const int n_levels = (rest + max_bits - 1) / max_bits;
  const int base = rest / n_levels, extra = rest % n_levels;
  for (int i = 0; i < n_levels; ++i) levels[i] = base + (i < extra ? 1 : 0);
  return n_levels;
}

size_t state_bytes(int64_t n_buckets) { return (size_t)bnpk_finish_state_words(n_buckets) * 8; }

enum {
  SP_TRY_PLAIN = 1,                                     // (internal) the claiming level could not be used: take the plain one
  SP_NEEDS_KEYS = 2                                     // (internal) segment-blind words would have to be sorted as a whole: see `blind`
};

struct sparse_info {
  int path = 0;          // 1 = claiming level + strided finish, 2 = plain levels + finish, 3 = sorted (fall-back), 0 = empty input
  int levels = 0;        // partition levels run (the claiming one included)
  int syncs = 0;         // host round trips
  int64_t n_bag = 0;     // keys the claiming level put into the bag
  int n_precounted = 0;  // heavy buckets counted one by one
};

// d_part_offsets / part_bits / n_seg_in: the keys come grouped in n_seg_in segments (0: 2^part_bits of them) inside each of
// which the top part_bits bits (below the skipped ones) are the same.
// blind: the keys do not say which segment they lie in (the index's words: the k-mer's top bits are the segment and not part of
// the word) — equal keys of two segments are two results, so nothing may order or merge keys ACROSS segments: the claiming
// level's bag (merged by key) is only accepted empty, the batch of over-full buckets is relabelled (precount_buckets), and where
// only a sort of everything is left the call gives up with SP_NEEDS_KEYS (the caller builds the index from whole keys).
int count_sparse_impl(bnpk_ctx* ctx, int64_t* d_keys, int64_t n, int key_bits, int skip_bits, int64_t n_plan,
                      const int64_t* d_part_offsets, int part_bits, arena_t& arena, int64_t* d_keys_out, int64_t* d_counts_out,
                      int64_t* h_n_unique, sparse_info& info, hipStream_t s, int depth, int64_t n_seg_in = 0, bool blind = false);

// words[j]'s bits from `shift` up <- labels[i] (NULL: i itself), i = the segment j lies in (seg_first[i] <= j, nb segments)
__global__ __launch_bounds__(256) void relabel_kernel(uint64_t* __restrict__ words, int64_t n, const int64_t* __restrict__ seg_first,
                                                      int nb, int shift, const int64_t* __restrict__ labels) {
  const uint64_t low_mask = (1ull << shift) - 1ull;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
    int lo = 0, hi = nb;                                 // first segment that starts behind j
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (seg_first[mid] <= j) lo = mid + 1; else hi = mid;
    }
    const int i = lo - 1;
    const uint64_t label = labels ? (uint64_t)labels[i] : (uint64_t)i;
    words[j] = (words[j] & low_mask) | (label << shift);
  }
}

// (sorted distinct keys, counts) by the library sort + run kernels: what heavy-hitter inputs take.  `work` is consumed.
// keys_out NULL: the distinct keys go to the ping-pong buffer the sort left free, *keys_where says which, *sorted_where where
// the sorted keys lie (dead after the call: the caller may reuse it); `alt` NULL: taken from the arena.
int count_by_sorting(bnpk_ctx* ctx, int64_t* work, int64_t n, int key_bits, arena_t& arena, int64_t* keys_out, int64_t* counts_out,
                     int64_t* h_n_unique, sparse_info& info, hipStream_t s, int64_t* alt = nullptr, int64_t** keys_where = nullptr,
                     int64_t** sorted_where = nullptr) {
  if (!alt) alt = arena.words(n);
  const int64_t tiles = bnpk_run_tiles(n);
  int64_t* tile_off = arena.words(tiles + 1);
  if (!alt || !tile_off) return SP_NOMEM(arena);
  int in_alt = 0;
  BNPK_CHECK(bnpk_sort_keys(ctx, work, alt, n, 0, std::min(key_bits, 64), &in_alt, s));
  int64_t* sorted = in_alt ? alt : work;
  int64_t n_runs = 0;
  BNPK_CHECK(bnpk_run_census(ctx, sorted, nullptr, n, tile_off, &n_runs, s));
  ++info.syncs;
  int64_t* starts = arena.words(n_runs + 1);
  if (!starts) return SP_NOMEM(arena);
  if (!keys_out) keys_out = in_alt ? work : alt;
  BNPK_CHECK(bnpk_run_heads(ctx, sorted, nullptr, n, tile_off, n_runs, keys_out, nullptr, starts, s));
  BNPK_CHECK(bnpk_run_sums(ctx, starts, n_runs, nullptr, counts_out, s));
  if (keys_where) *keys_where = keys_out;
  if (sorted_where) *sorted_where = sorted;
  *h_n_unique = n_runs;
  info.path = 3;
  return BNPK_OK;
}

// (table, keys, counts) for bnpk_finish_sorted: the listed buckets {bucket, index of its first key, its keys} (ascending by
// bucket) counted in ONE batch — gathered into one array, sorted and run-length-counted once (buckets differ in their top bits,
// so the batch sorts bucket by bucket), the distinct keys cut back into buckets by a binary search of every bucket's first key
// among the running key totals.
// depth 0 (round 6): the batch is counted by the planner itself — its segments are the listed buckets, `done` bits of every
// key are known inside each, one more level sized from the largest splits what mere skew made too large (the k-mers of a genome
// whose top digits are unevenly filled: sacCer3) — and the library sort is left to what no level can split (one key a million
// times), inside that call.
int precount_buckets(bnpk_ctx* ctx, const int64_t* keys, std::vector<int64_t>& listed, int key_bits, int skip_bits, int done,
                     arena_t& arena, int64_t** table_out, int64_t** big_keys, int64_t** big_counts, sparse_info& info, hipStream_t s,
                     int depth, bool blind) {
  const int nb = (int)(listed.size() / 3);
  // Segment-blind words (the index): two listed buckets of different first-level segments may hold EQUAL words, and the cut below
  // wants a batch that sorts bucket by bucket.  The top `done` bits of a word are the same all over its bucket, so for the time
  // of the count they carry the bucket's number in the batch instead, and the distinct words get their own bits back.
  const int label_shift = key_bits - skip_bits - done;
  if (blind && (depth != 0 || nb > (1ll << std::min(done, 20)))) return SP_NEEDS_KEYS;
  std::vector<int64_t> order(nb);
  for (int i = 0; i < nb; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return listed[3 * a] < listed[3 * b]; });
  std::vector<int64_t> ids(nb), lo8(nb), byte_off(nb + 1, 0), prefix(nb);
  int64_t total = 0;
  for (int i = 0; i < nb; ++i) {
    const int64_t* row = &listed[3 * order[i]];
    ids[i] = row[0];
    lo8[i] = row[1] * 8;
    prefix[i] = total;
    total += row[2];
    byte_off[i + 1] = total * 8;
  }
  // the batch, its distinct keys and their counts; the running totals go over the batch once it is counted
  int64_t largest = 0;
  std::vector<int64_t> word_off(nb + 1);
  for (int i = 0; i <= nb; ++i) word_off[i] = byte_off[i] / 8;
  for (int i = 0; i < nb; ++i) largest = std::max(largest, word_off[i + 1] - word_off[i]);
  const bool recurse = depth == 0;
  int64_t* batch = arena.words(total + 1);
  int64_t* batch_alt = recurse ? nullptr : arena.words(total + 1);
  int64_t* k = recurse ? arena.words(total) : nullptr;
  int64_t* c = arena.words(total);
  int64_t* d_lo = arena.words(nb);
  int64_t* d_off = arena.words(nb + 1);
  int64_t* d_woff = arena.words(nb + 1);
  int64_t* d_prefix = arena.words(nb);
  int64_t* d_starts = arena.words(nb);
  int64_t* d_labels = blind ? arena.words(nb) : nullptr;
  int64_t* table = arena.words(3 * (int64_t)nb);
  if (blind && !d_labels) return SP_NOMEM(arena);
  if (!batch || (!recurse && !batch_alt) || (recurse && !k) || !c || !d_lo || !d_off || !d_woff || !d_prefix || !d_starts || !table)
    return SP_NOMEM(arena);
  BNPK_HIP(ctx, hipMemcpyAsync(d_lo, lo8.data(), (size_t)nb * 8, hipMemcpyHostToDevice, s));
  BNPK_HIP(ctx, hipMemcpyAsync(d_off, byte_off.data(), (size_t)(nb + 1) * 8, hipMemcpyHostToDevice, s));
  BNPK_HIP(ctx, hipMemcpyAsync(d_woff, word_off.data(), (size_t)(nb + 1) * 8, hipMemcpyHostToDevice, s));
  BNPK_HIP(ctx, hipMemcpyAsync(d_prefix, prefix.data(), (size_t)nb * 8, hipMemcpyHostToDevice, s));
  if (blind) {
    std::vector<int64_t> own_bits(nb);
    for (int i = 0; i < nb; ++i) own_bits[i] = ids[i] & ((1ll << done) - 1);
    BNPK_HIP(ctx, hipMemcpyAsync(d_labels, own_bits.data(), (size_t)nb * 8, hipMemcpyHostToDevice, s));
  }
  BNPK_HIP(ctx, hipStreamSynchronize(s));                // (the host vectors go out of scope; pageable copies are staged anyway)
  BNPK_CHECK(bnpk_gather_rows(ctx, reinterpret_cast<const uint8_t*>(keys), d_lo, d_off, nb, total * 8, 0,
                              reinterpret_cast<uint8_t*>(batch), s));
  const unsigned relabel_grid = (unsigned)std::min<int64_t>(ceil_div(total, 256), (int64_t)ctx->compute_units * 16);
  if (blind) {
    hipLaunchKernelGGL(relabel_kernel, dim3(relabel_grid), dim3(256), 0, s, reinterpret_cast<uint64_t*>(batch), total, (const int64_t*)d_woff, nb,
                       label_shift, (const int64_t*)nullptr);
    BNPK_HIP(ctx, hipGetLastError());
  }
  int64_t d = 0;
  int64_t* cum = nullptr;
  if (recurse) {
    // (planned as if every one of the 2^done possible segments were as large as the largest listed one)
    const int64_t n_plan = done < 40 ? std::min<int64_t>(largest << done, 1ll << 61) : 1ll << 61;
    sparse_info inner;
    BNPK_CHECK(count_sparse_impl(ctx, batch, total, key_bits, skip_bits, n_plan, d_woff, done, arena, k, c, &d, inner, s, depth + 1, nb));
    info.syncs += inner.syncs;
    info.levels += inner.levels;
    cum = batch;                                         // (consumed by the call: free now)
  } else {
    BNPK_CHECK(count_by_sorting(ctx, batch, total, key_bits, arena, nullptr, c, &d, info, s, batch_alt, &k, &cum));
  }
  BNPK_CHECK(bnpk_exclusive_scan_i64(ctx, c, d, cum, s));
  BNPK_CHECK(bnpk_search_sorted(ctx, cum, d + 1, d_prefix, nb, 0, d_starts, s));
  if (blind && d > 0) {
    hipLaunchKernelGGL(relabel_kernel, dim3(std::max(1u, std::min(relabel_grid, (unsigned)ceil_div(d, 256)))), dim3(256), 0, s,
                       reinterpret_cast<uint64_t*>(k), d, (const int64_t*)d_starts, nb, label_shift, (const int64_t*)d_labels);
    BNPK_HIP(ctx, hipGetLastError());
  }
  std::vector<int64_t> starts(nb);
  BNPK_CHECK(bnpk_fetch_i64(ctx, d_starts, nb, starts.data(), s));
  ++info.syncs;
  std::vector<int64_t> host_table(3 * (size_t)nb);
  for (int i = 0; i < nb; ++i) {
    const int64_t end = i + 1 < nb ? starts[i + 1] : d;
    host_table[3 * i] = ids[i];
    host_table[3 * i + 1] = end - starts[i];
    host_table[3 * i + 2] = starts[i];
  }
  BNPK_HIP(ctx, hipMemcpyAsync(table, host_table.data(), host_table.size() * 8, hipMemcpyHostToDevice, s));
  BNPK_HIP(ctx, hipStreamSynchronize(s));
  *table_out = table;
  *big_keys = k;
  *big_counts = c;
  info.n_precounted = nb;
  return BNPK_OK;
}

// bytes the claiming level + strided finish need for n keys in n_b buckets (besides the outputs)
size_t claimed_bytes(int64_t n, int64_t n_b) {
  const int64_t stride = bnpk_claimed_stride();
  const int64_t bag_cap = std::max<int64_t>(n / 8, 1 << 16);
  return (size_t)n_b * stride * 8 + (size_t)n_b * 8 + (size_t)bag_cap * 8 + 8 + (size_t)(n_b + 1) * 8 + state_bytes(n_b) + 16 * 256;
}

// the last level + finishing stage through buckets of fixed stride (bnpk_radix_partition_claimed, bnpk_claimed_finalize,
// bnpk_finish_sorted_strided); the keys that found no place in their bucket (the bag) are counted on their own and merged in.
// SP_TRY_PLAIN if the bag overflowed / a wait between workgroups gave up / the merge has no room: `cur` is intact then.
int count_claimed(bnpk_ctx* ctx, int64_t* cur, int64_t n, const int64_t* offsets, int64_t n_seg, int shift, int bits, int key_bits,
                  arena_t& arena, int64_t* keys_out, int64_t* counts_out, int64_t* h_n_unique, sparse_info& info, hipStream_t s,
                  int depth, bool blind) {
  const int64_t n_b = n_seg << bits, stride = bnpk_claimed_stride();
  const int64_t bag_cap = std::max<int64_t>(n / 8, 1 << 16);
  const size_t mark = arena.used;
  int64_t* buckets = arena.words(n_b * stride);
  uint32_t* fill = reinterpret_cast<uint32_t*>(arena.take((size_t)n_b * 8));
  int64_t* bag = arena.words(bag_cap);
  int64_t* bag_fill = arena.words(1);
  int64_t* b_off = arena.words(n_b + 1);
  int64_t* state = reinterpret_cast<int64_t*>(arena.take(state_bytes(n_b)));
  if (!buckets || !fill || !bag || !bag_fill || !b_off || !state) {
    arena.used = mark;
    return SP_TRY_PLAIN;
  }
  BNPK_CHECK(bnpk_radix_partition_claimed(ctx, cur, n, offsets, n_seg, shift, bits, buckets, fill, bag, bag_cap, bag_fill, s));
  BNPK_CHECK(bnpk_claimed_finalize(ctx, buckets, fill, n_b, b_off, s));
  int64_t n_bag = 0;
  BNPK_CHECK(bnpk_fetch_i64(ctx, bag_fill, 1, &n_bag, s));
  ++info.syncs;
  info.n_bag = n_bag;
  if (n_bag > bag_cap || (blind && n_bag > 0)) {         // keys were dropped (or could not be merged back by value): the plain level
    arena.used = mark;
    return SP_TRY_PLAIN;
  }
  const int64_t n_in = n - n_bag;
  int64_t d = 0;
  int overflow = 0;
  if (n_in > 0) {
    BNPK_CHECK(bnpk_finish_sorted_strided(ctx, buckets, n_in, stride, b_off, n_b, shift, keys_out, counts_out, state, nullptr, 0,
                                          nullptr, nullptr, &d, &overflow, s));
    ++info.syncs;
    if (overflow) {                                      // (a wait between workgroups gave up: the caller's plain path sorts)
      arena.used = mark;
      return SP_TRY_PLAIN;
    }
  }
  ++info.levels;
  info.path = 1;
  if (n_bag > 0) {
    // the bag's keys: counted the same way (a smaller problem), then one merge along the merge path.  The buckets are free:
    // the bag's result and the merged lists live there if they fit (keys_out may be the caller's input: not scratch).
    arena_t sub;
    sub.base = reinterpret_cast<char*>(buckets);
    sub.size = (size_t)n_b * stride * 8;
    int64_t* bk = sub.words(n_bag);
    int64_t* bc = sub.words(n_bag);
    int64_t* mk = arena.words(d + n_bag);                // (the merged lists: behind the level's slots if there is room, else in them)
    int64_t* mc = mk ? arena.words(d + n_bag) : nullptr;
    if (!mc) {
      mk = sub.words(d + n_bag);
      mc = sub.words(d + n_bag);
    }
    if (!bk || !bc || !mk || !mc) {                      // (no room for the merge: `cur` is intact, the plain level counts it)
      arena.used = mark;
      --info.levels;
      return SP_TRY_PLAIN;
    }
    int64_t d_bag = 0;
    sparse_info inner;
    const int rb = count_sparse_impl(ctx, bag, n_bag, key_bits, 0, n_bag, nullptr, 0, sub, bk, bc, &d_bag, inner, s, depth + 1);
    info.syncs += inner.syncs;
    if (rb == BNPK_ERR_NOMEM) {
      arena.used = mark;
      --info.levels;
      return SP_TRY_PLAIN;
    }
    BNPK_CHECK(rb);
    int64_t m = 0;
    BNPK_CHECK(bnpk_merge_add(ctx, d ? keys_out : nullptr, d ? counts_out : nullptr, d, bk, bc, d_bag, mk, mc, &m, s));
    ++info.syncs;
    BNPK_HIP(ctx, hipMemcpyAsync(keys_out, mk, (size_t)m * 8, hipMemcpyDeviceToDevice, s));
    BNPK_HIP(ctx, hipMemcpyAsync(counts_out, mc, (size_t)m * 8, hipMemcpyDeviceToDevice, s));
    d = m;
  }
  *h_n_unique = d;
  return BNPK_OK;
}

int count_sparse_impl(bnpk_ctx* ctx, int64_t* d_keys, int64_t n, int key_bits, int skip_bits, int64_t n_plan,
                      const int64_t* d_part_offsets, int part_bits, arena_t& arena, int64_t* d_keys_out, int64_t* d_counts_out,
                      int64_t* h_n_unique, sparse_info& info, hipStream_t s, int depth, int64_t n_seg_in, bool blind) {
  if (n == 0) {
    *h_n_unique = 0;
    return BNPK_OK;
  }
  if (depth > 3) return BNPK_ERR_RANGE;
  int64_t* cur = d_keys;
  int64_t* spare = nullptr;                              // a free n-word buffer: what the previous level read
  if (key_bits > 63)                                     // (no room for the phantom bit: the library sort)
    return count_by_sorting(ctx, cur, n, key_bits, arena, d_keys_out, d_counts_out, h_n_unique, info, s);
  const int kb = key_bits - skip_bits;
  const int64_t* offsets = d_part_offsets;
  int done = part_bits;
  int64_t n_seg = n_seg_in > 0 ? n_seg_in : 1ll << done;
  // A small input that comes in segments (the index's words behind their first level, a chunk's k-mers behind the fused level):
  // the segments' sizes are known, so the levels are planned for the LARGEST of them instead of the average — a genome's k-mers
  // fill their top digits unevenly (sacCer3: the fullest of 1024 buckets holds 6x the average), and planning for the average
  // meant a claiming attempt whose bag overflows, a level too narrow, and a batch of hundreds of over-full buckets.  One census
  // (one answer) up front; with the bitonic finishing kernel small buckets are cheap.
  if (depth == 0 && offsets && n_seg > 1 && n <= (1ll << 25)) {
    int64_t* census = arena.words(2 + 3 * 4);
    if (!census) return SP_NOMEM(arena);
    BNPK_CHECK(bnpk_bucket_census(ctx, offsets, n_seg, bnpk_finish_capacity(), 4, census, s));
    int64_t got[2] = {0, 0};
    BNPK_CHECK(bnpk_fetch_i64(ctx, census, 2, got, s));
    ++info.syncs;
    const int64_t per_segment = done < 40 ? std::min<int64_t>(got[0] << done, 1ll << 61) : 1ll << 61;    // (as if all 2^done were that large)
    const int64_t as_if = n_seg_in > 0 ? got[0] : per_segment;
    if (as_if > (n_plan > 0 ? n_plan : n)) n_plan = as_if;
  }
  int plan[8];
  const int n_levels = radix_plan(n_plan > 0 ? n_plan : n, kb, done, plan);
  // (the claiming level is for the caller's keys.  What a recursion counts — the bag of a claiming level, a batch of over-full
  // buckets — is what did NOT spread evenly: on deep coverage of a small genome the bag's own claiming attempt overflowed in turn,
  // 20 ms for nothing before the plain level ran — round 6)
  const bool may_claim = ctx->sparse_claim != 0 && depth == 0;
  for (int level = 0; level < n_levels; ++level) {
    const int bits = plan[level];
    const int shift = kb - done - bits;
    if (level == n_levels - 1 && may_claim && bits >= 1 && bits <= 10 && n >= CLAIM_MIN_KEYS &&
        (n_seg << bits) * bnpk_claimed_stride() <= 3 * n && claimed_bytes(n, n_seg << bits) <= arena.left()) {
      const int r = count_claimed(ctx, cur, n, offsets, n_seg, shift, bits, key_bits, arena, d_keys_out, d_counts_out, h_n_unique, info, s, depth, blind);
      if (r != SP_TRY_PLAIN) return r;
    }
    int64_t* out = spare ? spare : arena.words(n);
    int64_t* child = arena.words((n_seg << bits) + 1);
    if (!out || !child) return SP_NOMEM(arena);
    BNPK_CHECK(bnpk_radix_partition(ctx, cur, n, offsets, n_seg, shift, bits, out, child, s));
    spare = cur;
    cur = out;
    offsets = child;
    done += bits;
    n_seg <<= bits;
    ++info.levels;
  }
  if (!offsets) {
    int64_t* two = arena.words(2);
    if (!two) return SP_NOMEM(arena);
    const int64_t host_two[2] = {0, n};
    BNPK_HIP(ctx, hipMemcpyAsync(two, host_two, 16, hipMemcpyHostToDevice, s));
    BNPK_HIP(ctx, hipStreamSynchronize(s));
    offsets = two;
  }
  // The plan assumes well-spread keys; the real bucket sizes decide.  MANY buckets over the finishing kernels' capacity
  // (skewed / duplicate-heavy keys): up to two extra levels sized from the largest bucket.  A FEW (heavy-hitter k-mers:
  // extra levels cannot split equal keys): those are counted one by one and handed to the finishing kernel ready-made.
  const int64_t cap = bnpk_finish_capacity();
  bool fits = false;
  int64_t *table = nullptr, *big_keys = nullptr, *big_counts = nullptr;
  int n_big = 0;
  for (int attempt = 0; attempt < 3; ++attempt) {
    int64_t* census = arena.words(2 + 3 * MAX_PRECOUNTED);
    if (!census) return SP_NOMEM(arena);
    BNPK_CHECK(bnpk_bucket_census(ctx, offsets, n_seg, cap, MAX_PRECOUNTED, census, s));
    std::vector<int64_t> got(2 + 3 * MAX_PRECOUNTED);
    BNPK_CHECK(bnpk_fetch_i64(ctx, census, (int64_t)got.size(), got.data(), s));
    ++info.syncs;
    const int64_t largest = got[0], n_over = got[1];
    if (largest <= cap) {
      fits = true;
      break;
    }
    int bits = std::max(1, (int)std::ceil(std::log2((double)largest / (0.7 * (double)cap))));
    bits = std::min(std::min(11, kb - done), bits);
    if (n_over <= MAX_PRECOUNTED && depth == 0) {         // (a batch that is itself being pre-counted: levels, then the sort)
      std::vector<int64_t> listed(got.begin() + 2, got.begin() + 2 + 3 * n_over);
      BNPK_CHECK(precount_buckets(ctx, cur, listed, key_bits, skip_bits, done, arena, &table, &big_keys, &big_counts, info, s, depth, blind));
      n_big = (int)n_over;
      fits = true;
      break;
    }
    // (never more buckets than half the keys: levels cannot split EQUAL keys, and two 8-bit levels over a batch whose keys differ
    // in their last bits only — one k-mer in a thousand rows — asked for 2^25 buckets' offsets and state for 1.4 M keys: found
    // by tests/test_fuzz.py as a workspace that was too small)
    while (bits > 0 && (n_seg << bits) > std::max<int64_t>(n / 2, 1 << 12)) --bits;
    if (attempt == 2 || bits <= 0) break;                // too many heavy buckets: the full sort below
    int64_t* out = spare ? spare : arena.words(n);
    int64_t* child = arena.words((n_seg << bits) + 1);
    if (!out || !child) break;                           // (no room for another level: the sort)
    if (bits <= 4 && n_seg > 4096 && largest <= bnpk_radix_small_capacity())
      BNPK_CHECK(bnpk_radix_partition_small(ctx, cur, n, offsets, n_seg, kb - done - bits, bits, out, child, s));
    else
      BNPK_CHECK(bnpk_radix_partition(ctx, cur, n, offsets, n_seg, kb - done - bits, bits, out, child, s));
    spare = cur;
    cur = out;
    offsets = child;
    done += bits;
    n_seg <<= bits;
    ++info.levels;
  }
  if (fits) {
    // (bnpk_finish_sorted uses the partitioned keys as workspace; d_keys_out is never the input: checked by the entry point)
    int64_t* state = reinterpret_cast<int64_t*>(arena.take(state_bytes(n_seg)));
    if (!state) return SP_NOMEM(arena);
    int64_t d = 0;
    int overflow = 0;
    BNPK_CHECK(bnpk_finish_sorted(ctx, cur, n, offsets, n_seg, kb - done, d_keys_out, d_counts_out, state, table, n_big, big_keys,
                                  big_counts, &d, &overflow, s));
    ++info.syncs;
    if (overflow & 1) return BNPK_ERR_RANGE;             // every bucket was checked against the capacity above
    if (overflow == 0) {
      *h_n_unique = d;
      info.path = 2;
      return BNPK_OK;
    }
    // a wait between workgroups gave up (a run-time condition): the partitioned keys are intact, the sort gives the answer
  }
  if (blind) return SP_NEEDS_KEYS;
  return count_by_sorting(ctx, cur, n, key_bits, arena, d_keys_out, d_counts_out, h_n_unique, info, s, spare);
}

// ---- k-mer index ---------------------------------------------------------------------------------------------------------
constexpr int PT_MAX_BITS = 22;                           // prefix table: first distinct key at or above every prefix of up to 22 bits

// as many prefix bits as leave ~8 distinct keys per prefix: the search behind the table stays inside one or two cache lines
int prefix_bits(int64_t d, int key_bits) {
  int b = 8;
  while (b < PT_MAX_BITS && (d >> b) > 8) ++b;
  return std::min(b, key_bits);
}

// table[p] = lower_bound(keys, p << (key_bits - pt_bits)) for p in [0, 2^pt_bits]
__global__ void prefix_table_kernel(const int64_t* __restrict__ keys, int64_t d, int key_bits, int pt_bits, int64_t* __restrict__ table) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p > (1ll << pt_bits)) return;
  if (p == (1ll << pt_bits)) {
    table[p] = d;
    return;
  }
  const int sh = key_bits - pt_bits;
  const int64_t q = p << sh;
  int64_t lo = 0, hi = d;
  while (lo < hi) {
    const int64_t mid = lo + ((hi - lo) >> 1);
    if (keys[mid] < q) lo = mid + 1; else hi = mid;
  }
  table[p] = lo;
}

// ids[i] = rank(kmer_i) * n_rows + row_i, the rank by a binary search inside the k-mer's prefix range (a few cache lines)
__global__ __launch_bounds__(256) void rank_compose_kernel(const int64_t* __restrict__ kmers, const int64_t* __restrict__ rows,
                                                           int64_t n, const int64_t* __restrict__ keys,
                                                           const int64_t* __restrict__ table, int key_bits, int pt_bits,
                                                           int64_t n_rows, int64_t* __restrict__ ids) {
  const int sh = key_bits - pt_bits;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int64_t q = kmers[i];
    const int64_t p = q >> sh;
    int64_t lo = table[p], hi = table[p + 1];
    while (lo < hi) {
      const int64_t mid = lo + ((hi - lo) >> 1);
      if (keys[mid] < q) lo = mid + 1; else hi = mid;
    }
    ids[i] = lo * n_rows + rows[i];
  }
}

// ---- the index as ONE partition of (k-mer, row) pairs (round 6) ------------------------------------------------------------
// flips[j] = 1 where the tag bit of word j differs from word j - 1's: a new first-level bucket starts there
__global__ __launch_bounds__(256) void pair_flips_kernel(const uint64_t* __restrict__ words, int64_t d, int64_t* __restrict__ flips) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < d; j += stride)
    flips[j] = j > 0 ? (int64_t)((words[j] ^ words[j - 1]) & 1ull) : 0;
}

// word j lies in the (before[j] + flips[j])-th non-empty first-level bucket: k-mer = bucket : word's k-mer bits, row = its row bits
__global__ __launch_bounds__(256) void pair_decode_kernel(const uint64_t* __restrict__ words, const int64_t* __restrict__ before,
                                                          int64_t d, const int64_t* __restrict__ list, int low_bits, int row_bits,
                                                          int64_t* __restrict__ keys_out, int64_t* __restrict__ rows_out) {
  const uint64_t row_mask = (1ull << row_bits) - 1ull;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < d; j += stride) {
    const uint64_t w = words[j] >> 1;
    const bool flip = j > 0 && ((words[j] ^ words[j - 1]) & 1ull);
    const int64_t bucket = list[before[j] + (flip ? 1 : 0)];
    keys_out[j] = (int64_t)(((uint64_t)bucket << low_bits) | (w >> row_bits));
    rows_out[j] = (int64_t)(w & row_mask);
  }
}

}  // namespace

extern "C" {

int64_t bnpk_count_sparse_workspace(int64_t n, int key_bits, int skip_bits, int64_t n_plan, int part_bits, int mode) {
  if (n <= 0) return 256;
  int plan[8];
  const int kb = std::max(1, std::min(key_bits, 62) - skip_bits);
  const int n_levels = radix_plan(n_plan > 0 ? n_plan : n, kb, part_bits, plan);
  int total_bits = part_bits;
  for (int i = 0; i < n_levels; ++i) total_bits += plan[i];
  const int64_t n_b = 1ll << total_bits;
  // mode 0: the levels' ping-pong buffer (none if no level runs), every level's offsets, the finishing state, the census
  // ... and room to count a few buckets that came out a little over the finishing capacity in a batch of their own (repeated
  // k-mers make the bucket sizes of a genome's reads vary: at 60x coverage a handful of 2^20 buckets end up over 8192 keys)
  const size_t heavy_room = std::min<size_t>((size_t)256 << 20, (size_t)n * 8 * 5);
  size_t bytes = (n_levels > 0 ? (size_t)n * 8 : 0) + (size_t)n_b * 8 * 2 + state_bytes(n_b) + (size_t)(2 + 3 * MAX_PRECOUNTED) * 8 * 4 +
                 (1 << 20) + heavy_room;
  if (mode >= 1 && n_levels > 0 && plan[n_levels - 1] <= 10 && n >= CLAIM_MIN_KEYS &&
      n_b * bnpk_claimed_stride() <= 3 * n)
    // mode 1: or the claiming level's slots — and, for inputs small enough that a slab's leftovers overflow the buckets' tails
    // into the bag as a matter of course (one level over a few million keys), room to merge the bag's counts in
    bytes = std::max(bytes, claimed_bytes(n, n_b) + (size_t)n_b * 8 + (1 << 20) + heavy_room +
                                (n <= (1ll << 26) ? (size_t)(n + n / 8 + (1 << 16)) * 16 : 0));
  if (mode >= 2)
    // mode 2: any input — the ping-pong buffer whether planned or not, two more levels' offsets (2^8 times the buckets at most),
    // five arrays of n words for the heavy buckets' batch (the batch, its distinct keys and counts, the inner call's ping-pong
    // buffer, the runs' starts of the library sort)
    bytes = std::max(bytes, (size_t)n * 8 * 6 + (size_t)std::min<int64_t>(n_b << 8, 2 * n + 2) * 8 * 2 + state_bytes(std::min<int64_t>(n_b << 8, n + 1)) +
                                ((size_t)bnpk_run_tiles(n) + 2) * 8 * 2 + (1 << 20));
  return (int64_t)bytes;
}

int bnpk_count_sparse(bnpk_ctx* ctx, int64_t* d_keys, int64_t n, int key_bits, int skip_bits, int64_t n_plan,
                      const int64_t* d_part_offsets, int part_bits, void* d_work, int64_t work_bytes, int64_t* d_keys_out,
                      int64_t* d_counts_out, int64_t* h_n_unique, int64_t* h_info5, void* stream) {
  if (!ctx || n < 0 || key_bits < 1 || key_bits > 64 || skip_bits < 0 || skip_bits >= key_bits || part_bits < 0 || part_bits > 20 ||
      !h_n_unique || work_bytes < 0)
    return BNPK_ERR_ARG;
  if (part_bits > 0 && !d_part_offsets) return BNPK_ERR_ARG;
  if (n >= (1ll << 35)) return BNPK_ERR_RANGE;
  if (n > 0 && (!d_keys || !d_keys_out || !d_counts_out || !d_work)) return BNPK_ERR_ARG;
  if (n > 0 && (d_counts_out == d_keys || d_keys_out == d_keys || d_keys_out == d_counts_out)) return BNPK_ERR_ARG;
  arena_t arena;
  arena.base = reinterpret_cast<char*>(d_work);
  arena.size = (size_t)work_bytes;
  sparse_info info;
  *h_n_unique = 0;
  const int st = count_sparse_impl(ctx, d_keys, n, key_bits, skip_bits, n_plan, part_bits ? d_part_offsets : nullptr, part_bits, arena,
                                   d_keys_out, d_counts_out, h_n_unique, info, (hipStream_t)stream, 0);
  if (h_info5) {
    h_info5[0] = info.path;
    h_info5[1] = info.levels;
    h_info5[2] = info.syncs;
    h_info5[3] = info.n_bag;
    h_info5[4] = info.n_precounted;
  }
  return st;
}

int64_t bnpk_index_build_workspace(int64_t n, int key_bits, int64_t n_rows) {
  if (n <= 0) return 256;
  // the k-mers' copy, its distinct keys + counts, the ids, the prefix table; the two counts share one workspace
  return (int64_t)((size_t)n * 8 * 4 + (((size_t)1 << PT_MAX_BITS) + 2) * 8 + (1 << 16)) +
         std::max(bnpk_count_sparse_workspace(n, key_bits, 0, 0, 0, 2), bnpk_count_sparse_workspace(n, 62, 0, 0, 0, 2));
}

int bnpk_index_build(bnpk_ctx* ctx, const int64_t* d_kmers, const int64_t* d_rows, int64_t n, int key_bits, int64_t n_rows,
                     void* d_work, int64_t work_bytes, int64_t* d_keys_out, int64_t* d_rows_out, int64_t* d_counts_out,
                     int64_t* h_n_pairs, void* stream) {
  if (!ctx || n < 0 || key_bits < 1 || key_bits > 62 || n_rows < 1 || !h_n_pairs || work_bytes < 0) return BNPK_ERR_ARG;
  *h_n_pairs = 0;
  if (n == 0) return BNPK_OK;
  if (!d_kmers || !d_rows || !d_work || !d_keys_out || !d_rows_out) return BNPK_ERR_ARG;
  if (n >= (1ll << 35)) return BNPK_ERR_RANGE;
  hipStream_t s = (hipStream_t)stream;
  arena_t arena;
  arena.base = reinterpret_cast<char*>(d_work);
  arena.size = (size_t)work_bytes;
  // ---- the direct way (round 6): ONE partition that carries the row along.  The pair's sort key (2k + row bits) does not fit a
  // word, but behind a first level over the k-mer's top t bits it does: word = (k-mer's other bits : row : tag) <= 63 bits
  // (radix.hip: pair_source, written by the fixed-line scatter, the only one that ranks by a digit the word does not hold); the
  // words are counted like any keys — the planner, with the first level's buckets as its segments — and come out sorted and
  // de-duplicated bucket after bucket; the tag bit flips at every bucket boundary of that compacted list, a scan of the flips
  // says which bucket a word lies in, and the k-mer's top bits are put back.  No rank look-ups, no second count.
  int row_bits = 1;
  while (row_bits < 31 && (n_rows - 1) >> row_bits) ++row_bits;
  {
    int need = 0;
    while (need < key_bits && (n >> need) > FINISH_TARGET) ++need;
    const int t = std::min(10, std::max(std::max(1, row_bits), std::min(need, 10)));
    if (ctx->index_pairs && row_bits <= t && key_bits > t && key_bits - t + row_bits + 1 <= 63) {
      int64_t* words = arena.words(n);
      int64_t* wout = arena.words(n);
      int64_t* counts = arena.words(n);
      int64_t* child = arena.words((1ll << t) + 1);
      int64_t* list = arena.words((1ll << t) + 1);
      unsigned* tags = reinterpret_cast<unsigned*>(arena.words(64));
      if (!words || !wout || !counts || !child || !list || !tags) return SP_NOMEM(arena);
      BNPK_CHECK(bnpk_pairs_partition_launch(ctx, d_kmers, d_rows, n, key_bits, t, row_bits, words, child, tags, list, list + (1ll << t), s));
      const int word_bits = key_bits - t + row_bits + 1;
      const size_t mark_pairs = arena.used;
      int64_t m = 0;
      sparse_info info;
      // (segments = the first level's buckets, none of the word's bits resolved inside them; planned per segment)
      const int st = count_sparse_impl(ctx, words, n, word_bits, 0, std::max<int64_t>(n >> t, 1), child, 0, arena, wout, counts, &m, info, s, 0,
                                       1ll << t, true);
      if (st != SP_NEEDS_KEYS && st != BNPK_ERR_NOMEM) {     // (no room for this construction: the other one has its own budget)
        BNPK_CHECK(st);
        {
          bnpk_timer tm(ctx, "index_decode", s);
          int64_t* flips = words;                              // (consumed by the count: free)
          const unsigned grid = grid_for(std::min<int64_t>(ceil_div(m, 256), (int64_t)ctx->compute_units * 16));
          hipLaunchKernelGGL(pair_flips_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<const uint64_t*>(wout), m, flips);
          BNPK_HIP(ctx, hipGetLastError());
        }
        arena.used = mark_pairs;                               // (the count's own tables are done with)
        int64_t* before = arena.words(m + 1);
        if (!before) return SP_NOMEM(arena);
        BNPK_CHECK(bnpk_exclusive_scan_i64(ctx, words, m, before, s));
        {
          bnpk_timer tm(ctx, "index_decode", s);
          const unsigned grid = grid_for(std::min<int64_t>(ceil_div(m, 256), (int64_t)ctx->compute_units * 16));
          hipLaunchKernelGGL(pair_decode_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<const uint64_t*>(wout), (const int64_t*)before, m,
                             (const int64_t*)list, key_bits - t, row_bits, d_keys_out, d_rows_out);
          BNPK_HIP(ctx, hipGetLastError());
        }
        if (d_counts_out) BNPK_HIP(ctx, hipMemcpyAsync(d_counts_out, counts, (size_t)m * 8, hipMemcpyDeviceToDevice, s));
        *h_n_pairs = m;
        return BNPK_OK;
      }
      // (words that only a sort of everything could count — thousands of over-full buckets of equal words, a wait that gave up:
      // they do not say which bucket they came from, so the index is built from whole keys below; the inputs are untouched)
      arena.used = 0;
    }
  }
  // ---- more rows than a first level has buckets (or the option off): the distinct values of rank(k-mer) * n_rows + row ----------
  int64_t* work_keys = arena.words(n);
  int64_t* distinct = arena.words(n);
  int64_t* counts = arena.words(n);
  int64_t* ids = arena.words(n);
  int64_t* table = arena.words((1ll << PT_MAX_BITS) + 1);
  if (!work_keys || !distinct || !counts || !ids || !table) return SP_NOMEM(arena);
  const size_t mark = arena.used;
  // 1. the distinct k-mers (the input is the caller's: counted on a copy)
  BNPK_HIP(ctx, hipMemcpyAsync(work_keys, d_kmers, (size_t)n * 8, hipMemcpyDeviceToDevice, s));
  int64_t d = 0;
  sparse_info info;
  BNPK_CHECK(count_sparse_impl(ctx, work_keys, n, key_bits, 0, n, nullptr, 0, arena, distinct, counts, &d, info, s, 0));
  arena.used = mark;
  // 2. every k-mer's rank among them, composed with its row: id = rank * n_rows + row orders like (k-mer, row)
  int id_bits = 1;
  while (id_bits < 63 && ((d * n_rows - 1) >> id_bits) != 0) ++id_bits;
  if (d > 0 && (d > ((1ll << 62) / n_rows))) return BNPK_ERR_RANGE;
  {
    bnpk_timer t(ctx, "index_rank_compose", s);
    const int pt_bits = prefix_bits(d, key_bits);
    hipLaunchKernelGGL(prefix_table_kernel, dim3((unsigned)(((1 << pt_bits) + 1 + 255) / 256)), dim3(256), 0, s,
                       (const int64_t*)distinct, d, key_bits, pt_bits, table);
    hipLaunchKernelGGL(rank_compose_kernel, dim3(grid_for(std::min<int64_t>(ceil_div(n, 256), (int64_t)ctx->compute_units * 32))),
                       dim3(256), 0, s, d_kmers, d_rows, n, (const int64_t*)distinct, (const int64_t*)table, key_bits, pt_bits, n_rows, ids);
    BNPK_HIP(ctx, hipGetLastError());
  }
  // 3. the distinct ids (ids is consumed; its distinct values go to work_keys, their multiplicities to counts)
  int64_t m = 0;
  sparse_info info2;
  BNPK_CHECK(count_sparse_impl(ctx, ids, n, id_bits, 0, n, nullptr, 0, arena, work_keys, counts, &m, info2, s, 0));
  // 4. back to (k-mer, row)
  BNPK_CHECK(bnpk_pair_split(ctx, work_keys, m, n_rows, distinct, d_keys_out, d_rows_out, s));
  if (d_counts_out) BNPK_HIP(ctx, hipMemcpyAsync(d_counts_out, counts, (size_t)m * 8, hipMemcpyDeviceToDevice, s));
  *h_n_pairs = m;
  return BNPK_OK;
}

}  // extern "C"
