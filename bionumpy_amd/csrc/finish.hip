// Histogram finishing kernel (A9, sparse path): input = k-mer keys radix-sorted on their TOP `part_bits`
// bits only (buckets of equal top bits are contiguous, order inside a bucket arbitrary).  One pass
//   * ranks the distinct keys inside every bucket (buckets are tiny when part_bits ~ log2(n)),
//   * run-length-counts them,
//   * and compacts (key, count) into the globally sorted output with a decoupled look-back over tiles,
// replacing the low radix passes AND the separate run-census / run-heads / run-sums passes:
// per key it reads 8 B and writes 16 B (all-distinct worst case) — the algorithmic floor of an RLE.
//
// A bucket larger than FB_CAP keys (heavy-hitter k-mers, low part_bits) sets the overflow flag; the caller
// then falls back to the full sort + run kernels of count.hip, so results never depend on this fast path.
#include <stdlib.h>

#include <algorithm>

#include "common.h"
#include "scan.h"

namespace {

constexpr int FB_T = 2048;                     // keys owned by one workgroup
constexpr int FB_CAP = 1024;                   // largest bucket the fast path handles
constexpr int FB_W = FB_T + FB_CAP;            // LDS window (24 KiB of keys)
constexpr int FB_WORDS = FB_W / 64;            // 48 bit-mask words
constexpr int FB_ITERS = FB_W / BNPK_BLOCK;    // 12 slots per lane

// state[] layout (unsigned long long): [0] ticket counter, [1] overflow flag, [2] n_unique, [3 + t] tile t
constexpr int ST_TICKET = 0, ST_OVERFLOW = 1, ST_UNIQUE = 2, ST_TILES = 3;
constexpr unsigned long long FLAG_AGG = 1ull << 62, FLAG_INC = 2ull << 62, VALUE_MASK = (1ull << 62) - 1;

__device__ __forceinline__ int prev_set(const uint64_t* __restrict__ m, int i) {
  int w = i >> 6;
  uint64_t v = m[w] & (~0ull >> (63 - (i & 63)));
  while (v == 0) v = m[--w];
  return (w << 6) + 63 - __clzll((long long)v);
}

// next set bit strictly after i, or -1
__device__ __forceinline__ int next_set(const uint64_t* __restrict__ m, int i) {
  int w = i >> 6;
  uint64_t v = ((i & 63) == 63) ? 0ull : (m[w] & (~0ull << ((i & 63) + 1)));
  while (v == 0) {
    if (++w >= FB_WORDS) return -1;
    v = m[w];
  }
  return (w << 6) + __ffsll((long long)v) - 1;
}

__global__ __launch_bounds__(BNPK_BLOCK) void finish_buckets_kernel(const uint64_t* __restrict__ A, int64_t n,
                                                                    int shift, int64_t n_tiles,
                                                                    uint64_t* __restrict__ keys_out,
                                                                    int64_t* __restrict__ counts_out,
                                                                    unsigned long long* __restrict__ state,
                                                                    int ablate) {
  __shared__ uint64_t key[FB_W];
  __shared__ uint64_t smask[FB_WORDS];          // bucket-start bits
  __shared__ uint64_t fmask[FB_WORDS];          // first-occurrence bits inside the owned range
  __shared__ int fprefix[FB_WORDS + 1];
  __shared__ long long sh_tile, sh_base;
  __shared__ int sh_s, sh_e;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // tickets make "tile id" follow the real dispatch order, so the look-back below can never wait on a
  // workgroup that has not started (no assumption about blockIdx scheduling).
  if (tid == 0) sh_tile = (long long)atomicAdd(&state[ST_TICKET], 1ull);
  __syncthreads();
  const int64_t t = sh_tile;
  const int64_t g0 = t * FB_T;
  const int wlen = (int)min((int64_t)FB_W, n - g0);
  const uint64_t prev = (g0 > 0) ? A[g0 - 1] : ~0ull;
#pragma unroll
  for (int j = 0; j < FB_ITERS; ++j) {
    int i = tid + BNPK_BLOCK * j;
    key[i] = (i < wlen) ? A[g0 + i] : ~0ull;    // sentinel: a bucket id no real key (< 2^62) can have
  }
  __syncthreads();
  // bucket-start bits; slot word index (4j + wave) is wave-uniform, so one ballot gives one mask word
#pragma unroll
  for (int j = 0; j < FB_ITERS; ++j) {
    int i = tid + BNPK_BLOCK * j;
    uint64_t k = key[i];
    uint64_t kp = (i > 0) ? key[i - 1] : prev;
    uint64_t m = __ballot((k >> shift) != (kp >> shift));
    if (lane == 0) smask[4 * j + wave] = m;
  }
  __syncthreads();
  if (tid == 0) {
    // owned range [s, e): the buckets that START inside the tile proper
    const int tile_end = min(FB_T, wlen);
    int s = -1, e = -1;
    for (int w = 0; w < FB_WORDS && (s < 0 || e < 0); ++w) {
      uint64_t v = smask[w];
      if (s < 0 && v) s = (w << 6) + __ffsll((long long)v) - 1;
      if (e < 0) {
        int lo = tile_end - (w << 6);
        uint64_t vv = lo <= 0 ? v : (lo >= 64 ? 0ull : (v & (~0ull << lo)));
        if (vv) e = (w << 6) + __ffsll((long long)vv) - 1;
      }
    }
    if (s < 0 || s >= tile_end) { s = 0; e = 0; }          // whole tile inside a bucket owned by an earlier tile
    else if (e < 0) { e = wlen; atomicOr(&state[ST_OVERFLOW], 1ull); }   // ran off the window
    sh_s = s;
    sh_e = e;
  }
  __syncthreads();
  const int s = sh_s, e = sh_e;
  // pass 1: bucket bounds + first-occurrence flag of every owned slot
  unsigned bounds[FB_ITERS];
  unsigned first_bits = 0;
  bool too_big = false;
#pragma unroll
  for (int j = 0; j < FB_ITERS; ++j) {
    int i = tid + BNPK_BLOCK * j;
    bool active = (i >= s) && (i < e);
    bool first = false;
    bounds[j] = 0;
    if (active) {
      int bs = prev_set(smask, i);
      int be = next_set(smask, i);
      if (be < 0 || be > e) be = e;
      if (be - bs > FB_CAP) too_big = true;
      bounds[j] = (unsigned)bs | ((unsigned)be << 16);
      uint64_t x = key[i];
      first = true;
      for (int q = bs; q < i; ++q)
        if (key[q] == x) { first = false; break; }
    }
    uint64_t fm = __ballot(first);
    if (lane == 0) fmask[4 * j + wave] = fm;
    if (first) first_bits |= 1u << j;
  }
  if (too_big) atomicOr(&state[ST_OVERFLOW], 1ull);
  __syncthreads();
  if (wave == 0) {                               // exclusive prefix of the popcounts of the 48 mask words
    int c = (lane < FB_WORDS) ? __popcll(fmask[lane]) : 0;
    int inc = wave_inclusive_scan(c);
    if (lane < FB_WORDS) fprefix[lane] = inc - c;
    if (lane == FB_WORDS - 1) fprefix[FB_WORDS] = inc;
  }
  __syncthreads();
  const long long aggregate = fprefix[FB_WORDS];
  if (wave == 0) {
    // decoupled look-back: each tile publishes ONE 64-bit word {flag, value}; no payload ordering needed.
    // The whole wavefront inspects 64 predecessors per round (lane 0 = nearest).
    unsigned long long* mine = &state[ST_TILES + t];
    long long base = 0;
    if (ablate & 1) base = g0;                    // timing ablation only (BNPK_ABLATE): skip the look-back
    else if (t > 0) {
      if (lane == 0)
        __hip_atomic_store(mine, FLAG_AGG | (unsigned long long)aggregate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int64_t hi = t - 1;
      while (true) {
        int64_t p = hi - lane;
        unsigned long long v = (p >= 0) ? __hip_atomic_load(&state[ST_TILES + p], __ATOMIC_RELAXED,
                                                            __HIP_MEMORY_SCOPE_AGENT)
                                        : FLAG_INC;                      // before tile 0: inclusive prefix 0
        unsigned long long flag = v & ~VALUE_MASK;
        uint64_t inc_mask = __ballot(flag == FLAG_INC);
        uint64_t invalid_mask = __ballot(flag == 0);
        int first_inc = inc_mask ? (__ffsll((long long)inc_mask) - 1) : 64;
        uint64_t need = (first_inc >= 63) ? ~0ull : ((2ull << first_inc) - 1ull);   // lanes 0..first_inc
        if (invalid_mask & need) { __builtin_amdgcn_s_sleep(2); continue; }          // not published yet
        long long contrib = (lane <= first_inc) ? (long long)(v & VALUE_MASK) : 0;
        contrib = wave_reduce_sum(contrib);
        base += __shfl(contrib, 0, 64);
        if (first_inc < 64) break;
        hi -= 64;
      }
    }
    if (lane == 0) {
      __hip_atomic_store(mine, FLAG_INC | (unsigned long long)(base + aggregate), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      if (t == n_tiles - 1) state[ST_UNIQUE] = (unsigned long long)(base + aggregate);
      sh_base = base;
    }
  }
  __syncthreads();
  const int64_t out0 = sh_base;
  // pass 2: every first occurrence writes (key, multiplicity) at its sorted position
#pragma unroll
  for (int j = 0; j < FB_ITERS; ++j) {
    if (!(first_bits & (1u << j))) continue;
    int i = tid + BNPK_BLOCK * j;
    int bs = (int)(bounds[j] & 0xffffu), be = (int)(bounds[j] >> 16);
    uint64_t x = key[i];
    int cnt = 0, rank = 0;
    for (int q = bs; q < be; ++q) {
      uint64_t y = key[q];
      cnt += (y == x);
      rank += (y < x) && ((fmask[q >> 6] >> (q & 63)) & 1ull);
    }
    int local = fprefix[bs >> 6] + __popcll(fmask[bs >> 6] & ((1ull << (bs & 63)) - 1ull)) + rank;
    if (ablate & 2) continue;                    // timing ablation only: skip the stores
    keys_out[out0 + local] = x;
    counts_out[out0 + local] = cnt;
  }
}


// ---------------------------------------------------------------------------------------------------------
// Tier 1: buckets of at most ~32 keys (part_bits ~ log2 n on well-spread keys).  Wave-synchronous: every
// wavefront owns a chunk of 1024 keys and walks it in 64-key windows that advance by 32 (a bucket starting in
// the first half of a window must end inside the window); keys live in registers, neighbours are reached
// with lane shuffles, there is no LDS window, no workgroup barrier and no inter-workgroup waiting:
// pass COUNT writes the number of distinct keys per chunk, a device scan turns that into output offsets,
// pass WRITE recomputes the windows and stores (key, multiplicity) at their final sorted positions.
// (A single-pass decoupled look-back was measured first: with ~8k wavefront-chunks in flight every chunk
// walks ~100 rounds of predecessors — 190 ms instead of 10.)
constexpr int FS_STEP = 32;
constexpr int FS_ITERS = 32;
constexpr int FS_CHUNK = FS_STEP * FS_ITERS;        // keys owned by one wavefront

template <bool WRITE>
__global__ __launch_bounds__(BNPK_BLOCK) void finish_small_kernel(const uint64_t* __restrict__ A, int64_t n,
                                                                  int shift, int64_t n_chunks,
                                                                  int64_t* __restrict__ chunk_counts,
                                                                  const int64_t* __restrict__ chunk_offsets,
                                                                  uint64_t* __restrict__ keys_out,
                                                                  int64_t* __restrict__ counts_out,
                                                                  unsigned long long* __restrict__ flags) {
  const int lane = threadIdx.x & 63;
  const int64_t chunk = (int64_t)blockIdx.x * (BNPK_BLOCK / 64) + (threadIdx.x >> 6);
  if (chunk >= n_chunks) return;
  // an oversized bucket anywhere voids the whole tier: later wavefronts leave at once
  if (!WRITE && __hip_atomic_load(&flags[ST_OVERFLOW], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
  const int64_t g0 = chunk * FS_CHUNK;
  const uint64_t le_mask = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);     // bits 0..lane
  uint64_t before = (g0 > 0) ? A[g0 - 1] : ~0ull;                               // key just before the window
  int64_t out = WRITE ? chunk_offsets[chunk] : 0;
  int run = 0;
  bool overflow = false;
  // software pipeline: the next window's keys are in flight while this one is processed
  int64_t idx = g0 + lane;
  uint64_t key = (idx < n) ? A[idx] : ~0ull;                                    // sentinel bucket after the data
  for (int it = 0; it < FS_ITERS; ++it) {
    const int64_t nidx = idx + FS_STEP;
    const uint64_t next_key = (it + 1 < FS_ITERS && nidx < n) ? A[nidx] : ~0ull;
    uint64_t kp = __shfl_up(key, 1, 64);
    if (lane == 0) kp = before;
    const uint64_t smask = __ballot((key >> shift) != (kp >> shift));
    const uint64_t below = smask & le_mask;
    const int bs = below ? 63 - __clzll((long long)below) : 64;
    const bool owned = (bs < FS_STEP) && (idx < n);
    const uint64_t above = smask & ~le_mask;
    const int be = above ? __ffsll((long long)above) - 1 : 64;
    if (owned && be == 64) overflow = true;                                     // bucket runs off the window
    if (!WRITE && __any(overflow)) {
      if (lane == 0) atomicOr(&flags[ST_OVERFLOW], 1ull);
      return;
    }
    bool first = owned;                                                         // first occurrence in its bucket?
    for (int d = 1; __any(owned && (lane - d >= bs)); ++d) {
      uint64_t y = __shfl(key, (lane - d) & 63, 64);
      if (owned && (lane - d >= bs) && y == key) first = false;
    }
    const uint64_t fmask = __ballot(first);
    if (WRITE) {
      int cnt = 0, rank = 0;
      for (int d = 0; __any(first && (bs + d < be)); ++d) {
        int q = bs + d;
        uint64_t y = __shfl(key, q & 63, 64);
        if (first && q < be) {
          cnt += (y == key);
          rank += (y < key) && ((fmask >> q) & 1ull);
        }
      }
      if (first) {
        int64_t at = out + run + __popcll(fmask & ((1ull << (bs & 63)) - 1ull)) + rank;
        keys_out[at] = key;
        counts_out[at] = cnt;
      }
    }
    run += __popcll(fmask);
    before = __shfl(key, FS_STEP - 1, 64);                                      // key just before the next window
    key = next_key;
    idx = nidx;
  }
  if (!WRITE) {
    if (lane == 0) chunk_counts[chunk] = run;
    if (__any(overflow) && lane == 0) atomicOr(&flags[ST_OVERFLOW], 1ull);
  }
}

}  // namespace

extern "C" {

int64_t bnpk_finish_state_words(int64_t n) {
  // tier 1: 4 flag words + (chunks + 1) offsets; tier 2: 3 words + one per tile
  const int64_t m = n <= 0 ? 0 : n;
  return 8 + std::max<int64_t>(ceil_div(m, FS_CHUNK) + 1, ceil_div(m, FB_T) + 1);
}

int bnpk_finish_buckets(bnpk_ctx* ctx, const int64_t* d_part_sorted, int64_t n, int key_bits, int part_bits,
                        int64_t* d_keys_out, int64_t* d_counts_out, int64_t* d_state, int64_t* h_n_unique,
                        int* h_overflow, void* stream) {
  if (!ctx || n < 0 || key_bits < 1 || key_bits > 62 || part_bits < 1 || part_bits > key_bits || !h_n_unique ||
      !h_overflow || !d_state)
    return BNPK_ERR_ARG;
  *h_n_unique = 0;
  *h_overflow = 0;
  if (n == 0) return BNPK_OK;
  if (!d_part_sorted || !d_keys_out || !d_counts_out) return BNPK_ERR_ARG;
  if (d_keys_out == d_part_sorted) return BNPK_ERR_ARG;           // tiles re-read their neighbours' keys
  int64_t n_tiles = ceil_div(n, FB_T);
  if (n_tiles > BNPK_MAX_BLOCKS) return BNPK_ERR_RANGE;
  hipStream_t s = (hipStream_t)stream;
  const char* ab = getenv("BNPK_ABLATE");        // kernel-timing experiments only; results are invalid when set
  const int ablate = ab ? atoi(ab) : 0;
  BNPK_HIP(ctx, hipMemsetAsync(d_state, 0, (size_t)bnpk_finish_state_words(n) * sizeof(int64_t), s));
  {
    bnpk_timer t(ctx, "finish_buckets", s);
    hipLaunchKernelGGL(finish_buckets_kernel, dim3((unsigned)n_tiles), dim3(BNPK_BLOCK), 0, s,
                       reinterpret_cast<const uint64_t*>(d_part_sorted), n, key_bits - part_bits, n_tiles,
                       reinterpret_cast<uint64_t*>(d_keys_out), d_counts_out,
                       reinterpret_cast<unsigned long long*>(d_state), ablate);
  }
  BNPK_HIP(ctx, hipGetLastError());
  int64_t host[3];
  BNPK_HIP(ctx, hipMemcpyAsync(host, d_state, sizeof(host), hipMemcpyDeviceToHost, s));
  BNPK_HIP(ctx, hipStreamSynchronize(s));
  *h_overflow = host[ST_OVERFLOW] != 0;
  *h_n_unique = host[ST_UNIQUE];
  return BNPK_OK;
}

int bnpk_finish_small(bnpk_ctx* ctx, const int64_t* d_part_sorted, int64_t n, int key_bits, int part_bits,
                      int64_t* d_keys_out, int64_t* d_counts_out, int64_t* d_state, int64_t* h_n_unique,
                      int* h_overflow, void* stream) {
  if (!ctx || n < 0 || key_bits < 1 || key_bits > 62 || part_bits < 1 || part_bits > key_bits || !h_n_unique ||
      !h_overflow || !d_state)
    return BNPK_ERR_ARG;
  *h_n_unique = 0;
  *h_overflow = 0;
  if (n == 0) return BNPK_OK;
  if (!d_part_sorted || !d_keys_out || !d_counts_out || d_keys_out == d_part_sorted) return BNPK_ERR_ARG;
  const int64_t n_chunks = ceil_div(n, FS_CHUNK);
  const int64_t blocks = ceil_div(n_chunks, BNPK_BLOCK / 64);
  if (blocks > BNPK_MAX_BLOCKS) return BNPK_ERR_RANGE;
  hipStream_t s = (hipStream_t)stream;
  // d_state: [0..3] flags, [4 .. 4+n_chunks] per-chunk distinct counts -> offsets (n_chunks+1 entries)
  unsigned long long* flags = reinterpret_cast<unsigned long long*>(d_state);
  int64_t* chunk_off = d_state + 4;
  void* scratch = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, bnpk_scan_scratch_bytes(n_chunks), &scratch));
  BNPK_HIP(ctx, hipMemsetAsync(d_state, 0, 4 * sizeof(int64_t), s));
  const int shift = key_bits - part_bits;
  const uint64_t* A = reinterpret_cast<const uint64_t*>(d_part_sorted);
  {
    bnpk_timer t(ctx, "finish_small_count", s);
    hipLaunchKernelGGL((finish_small_kernel<false>), dim3((unsigned)blocks), dim3(BNPK_BLOCK), 0, s, A, n, shift,
                       n_chunks, chunk_off, (const int64_t*)nullptr, (uint64_t*)nullptr, (int64_t*)nullptr, flags);
    BNPK_HIP(ctx, hipGetLastError());
    BNPK_CHECK(bnpk_scan_launch(ctx, chunk_off, n_chunks, 1, chunk_off, true, (int64_t*)scratch, s));
  }
  int64_t host_flags[3], total = 0;
  BNPK_HIP(ctx, hipMemcpyAsync(host_flags, d_state, sizeof(host_flags), hipMemcpyDeviceToHost, s));
  BNPK_HIP(ctx, hipMemcpyAsync(&total, chunk_off + n_chunks, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  BNPK_HIP(ctx, hipStreamSynchronize(s));
  *h_overflow = host_flags[ST_OVERFLOW] != 0;
  if (*h_overflow) return BNPK_OK;
  *h_n_unique = total;
  bnpk_timer t(ctx, "finish_small_write", s);
  hipLaunchKernelGGL((finish_small_kernel<true>), dim3((unsigned)blocks), dim3(BNPK_BLOCK), 0, s, A, n, shift,
                     n_chunks, (int64_t*)nullptr, (const int64_t*)chunk_off, reinterpret_cast<uint64_t*>(d_keys_out),
                     d_counts_out, flags);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

}  // extern "C"
