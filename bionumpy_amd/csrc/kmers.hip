// 2-bit k-mer hashes (A8), minimizers (A11) and ragged row ids for gfx950.
// Integer / HBM-write bound: 8 output bytes per k-mer against 0.25 input bytes per base.  Output-flat
// mapping: every lane produces two adjacent int64 (one 16-byte store, 1 KiB contiguous per wavefront
// instruction); the packed input words are shared between neighbouring lanes through L1.
#include "common.h"
#include "rows.h"
#include "kmer_gen.h"
#include "scan.h"

namespace {

constexpr int PAIRS = 2;                               // 16-byte stores per lane
constexpr int TILE_OUT = BNPK_BLOCK * 2 * PAIRS;       // 1024 outputs (8 KiB) per workgroup

__device__ __forceinline__ void store_pair(int64_t* __restrict__ out, int64_t o, int64_t n_out, int64_t a,
                                           int64_t b) {
  if (o + 1 < n_out && (((uintptr_t)(out + o)) & 15) == 0) {
    *reinterpret_cast<longlong2*>(out + o) = make_longlong2(a, b);
  } else {
    out[o] = a;
    if (o + 1 < n_out) out[o + 1] = b;
  }
}

// MINIMIZER=false: hash of the k-mer starting at each position.
// MINIMIZER=true : min over the n_kmers = window-k+1 k-mer hashes of each window.
template <bool MINIMIZER>
__global__ __launch_bounds__(BNPK_BLOCK) void kmer_kernel(const uint64_t* __restrict__ W,
                                                          const int64_t* __restrict__ in_off,
                                                          const int64_t* __restrict__ out_off, int64_t n_rows,
                                                          int64_t n_out, int k, int n_kmers,
                                                          const int64_t* __restrict__ tile_rows,
                                                          int64_t* __restrict__ out) {
  int64_t rr[2];
  int64_t tile = (int64_t)blockIdx.x * TILE_OUT;
  if (tile >= n_out) return;
  tile_row_range(tile_rows, blockIdx.x, gridDim.x, n_rows, rr[0], rr[1]);
  const uint64_t mask = (k == 32) ? ~0ull : ((1ull << (2 * k)) - 1ull);
#pragma unroll
  for (int p = 0; p < PAIRS; ++p) {
    int64_t o = tile + (int64_t)p * (BNPK_BLOCK * 2) + 2 * threadIdx.x;
    if (o >= n_out) break;
    row_cursor c = seek_row(in_off, out_off, rr[0], rr[1], o);
    word_window ww;
    int64_t v[2] = {0, 0};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      int64_t oo = o + q;
      if (oo >= n_out) break;
      if (q) next_output(c, in_off, out_off, oo);
      if (!MINIMIZER) {
        v[q] = (int64_t)(bits_at(W, c.in_pos, ww) & mask);
      } else {
        uint64_t m = ~0ull;
        for (int j = 0; j < n_kmers; ++j) {
          uint64_t h = bits_at(W, c.in_pos + j, ww) & mask;
          m = h < m ? h : m;
        }
        v[q] = (int64_t)m;
        ww.wi = -2;   // the next output starts one base later: re-seek the window
      }
    }
    store_pair(out, o, n_out, v[0], v[1]);
  }
}

__global__ __launch_bounds__(BNPK_BLOCK) void row_ids_kernel(const int64_t* __restrict__ off, int64_t n_rows,
                                                             int64_t n, const int64_t* __restrict__ tile_rows,
                                                             int64_t* __restrict__ rows) {
  int64_t rr[2];
  int64_t tile = (int64_t)blockIdx.x * TILE_OUT;
  if (tile >= n) return;
  tile_row_range(tile_rows, blockIdx.x, gridDim.x, n_rows, rr[0], rr[1]);
#pragma unroll
  for (int p = 0; p < PAIRS; ++p) {
    int64_t o = tile + (int64_t)p * (BNPK_BLOCK * 2) + 2 * threadIdx.x;
    if (o >= n) break;
    int64_t r0 = find_row(off, rr[0], rr[1], o);
    int64_t r1 = r0;
    if (o + 1 < n) { while (o + 1 >= off[r1 + 1]) ++r1; }
    store_pair(rows, o, n, r0, r1);
  }
}

// ---- position-flat generation (the fast path of bnpk_kmers / bnpk_minimizers) --------------------------------------
// Items are the flat base positions of the packed stream; a bit mask says at which of them a window (a k-mer, or
// a minimizer window of several k-mers) starts, so no lane ever looks a row up.  A lane owns eight consecutive
// positions: three packed words + one mask byte, one funnel shift for the first k-mer and a 2-bit roll for every
// further one.  The ragged output order equals the position order, so the output index of a window is its rank
// among the mask bits: tile counts (popcounts) -> scan -> ranks inside the tile by wave scans; the values are
// staged in LDS and leave the CU as one contiguous run per tile.
constexpr int WF_ITEMS = 8;
constexpr int WF_TILE = BNPK_BLOCK * WF_ITEMS;            // 2048 positions per workgroup
constexpr int WF_MAX_PER_WINDOW = 26;                     // k-mers per window the 64-bit "next bases" register covers

// number of windows starting in every tile of 2048 positions: one lane per 64-bit mask word, 32 words per tile
__global__ __launch_bounds__(BNPK_BLOCK) void wf_count_kernel(const uint64_t* __restrict__ mask, int64_t n_words,
                                                              int64_t n_tiles, int64_t* __restrict__ counts) {
  const int64_t w = (int64_t)blockIdx.x * BNPK_BLOCK + threadIdx.x;
  const unsigned c = w < n_words ? (unsigned)__popcll(mask[w]) : 0u;
  const unsigned inc = wave_inclusive_scan(c);
  const unsigned s31 = (unsigned)__builtin_amdgcn_readlane((int)inc, 31), s63 = (unsigned)__builtin_amdgcn_readlane((int)inc, 63);
  const int64_t tile = ((int64_t)blockIdx.x * BNPK_BLOCK + (threadIdx.x & ~63)) / 32;    // first tile of this wavefront
  if (lane_id() == 0 && tile < n_tiles) counts[tile] = s31;
  if (lane_id() == 32 && tile + 1 < n_tiles) counts[tile + 1] = s63 - s31;
}

__device__ __forceinline__ uint64_t wf_window(uint64_t w0, uint64_t w1, uint64_t w2, int sh) {   // 64 bits at bit sh (< 128)
  const uint64_t lo = sh < 64 ? w0 : w1, hi = sh < 64 ? w1 : w2;
  const int s6 = sh & 63;
  return s6 ? (lo >> s6) | (hi << (64 - s6)) : lo;
}

// A tile's values leave LDS as one contiguous run out[base .. end): two values per lane and store (16 bytes, aligned:
// an odd base sends its first value alone), streaming — the run is 16 KiB that nothing reads again soon.
__device__ __forceinline__ void wf_store_run(const uint64_t* __restrict__ stage, int64_t base, int64_t end,
                                             int64_t* __restrict__ out) {
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  const unsigned total = (unsigned)(end - base);
  const unsigned head = (unsigned)(base & 1) & (total ? 1u : 0u);
  if (head && threadIdx.x == 0) out[base] = (int64_t)stage[0];
  const unsigned pairs = (total - head) >> 1;
  for (unsigned p = threadIdx.x; p < pairs; p += BNPK_BLOCK) {
    u64x2 v;
    v.x = stage[head + 2 * p];
    v.y = stage[head + 2 * p + 1];
    __builtin_nontemporal_store(v, reinterpret_cast<u64x2*>(out + base + head + 2 * p));
  }
  if (((total - head) & 1u) && threadIdx.x == 0) out[base + total - 1] = (int64_t)stage[total - 1];
}

// per_window == 1: the hash of the k-mer at every marked position; > 1: the minimum over the per_window k-mers of
// the window that starts there
__global__ __launch_bounds__(BNPK_BLOCK) void wf_generate_kernel(const uint64_t* __restrict__ W, int64_t n_words,
                                                                 const uint8_t* __restrict__ mask8, int64_t n_bases,
                                                                 int k, int per_window,
                                                                 const int64_t* __restrict__ tile_off,
                                                                 int64_t* __restrict__ out) {
  __shared__ uint64_t stage[WF_TILE];
  __shared__ unsigned wsum[BNPK_BLOCK / 64];
  const int64_t o = (int64_t)blockIdx.x * WF_TILE + (int64_t)threadIdx.x * WF_ITEMS;
  const unsigned v = o < n_bases ? mask8[o >> 3] : 0u;
  const unsigned cnt = __popc(v);
  const unsigned inc = wave_inclusive_scan(cnt);
  if (lane_id() == 63) wsum[wave_id()] = inc;
  uint64_t vals[WF_ITEMS];
  if (v) {
    const int64_t wi = o >> 5;
    const uint64_t w0 = W[wi], w1 = W[wi + 1], w2 = wi + 2 < n_words ? W[wi + 2] : 0;
    const int sh0 = 2 * (int)(o & 31), top = 2 * k - 2;
    const uint64_t kmask = (1ull << (2 * k)) - 1ull;
    uint64_t h = wf_window(w0, w1, w2, sh0) & kmask;        // k-mer at position o
    const uint64_t next = wf_window(w0, w1, w2, sh0 + 2 * k);   // the bases that enter the k-mers at o+1, o+2, ...
#pragma unroll
    for (int q = 0; q < WF_ITEMS; ++q) {
      if (q) h = (h >> 2) | (((next >> (2 * (q - 1))) & 3ull) << top);
      uint64_t m = h;
      if (per_window > 1 && ((v >> q) & 1u)) {
        uint64_t t = h;
        for (int i = 1; i < per_window; ++i) {
          t = (t >> 2) | (((next >> (2 * (q + i - 1))) & 3ull) << top);
          m = t < m ? t : m;
        }
      }
      vals[q] = m;
    }
  }
  __syncthreads();
  unsigned rank = inc - cnt;
  for (int w = 0; w < wave_id(); ++w) rank += wsum[w];
  if (v) {
#pragma unroll
    for (int q = 0; q < WF_ITEMS; ++q)
      if ((v >> q) & 1u) stage[rank++] = vals[q];
  }
  __syncthreads();
  wf_store_run(stage, tile_off[blockIdx.x], tile_off[blockIdx.x + 1], out);
}

// Minimizers, PW k-mers per window known at compile time.  wf_generate_kernel rolls the PW hashes of every window anew
// — PW rolls and PW comparisons per output, and at PW = 10 (BASELINE config 3) the kernel was bound by those instructions,
// not by its 8 output bytes per window.  A lane's eight windows overlap: together they cover 8 + PW - 1 consecutive
// k-mers.  These are rolled ONCE, and the eight window minima come from a doubling scheme over them —
// a[i] = min(a[i], a[i + 2^j]) for j = 0 .. L - 1 turns a[i] into the minimum of 2^L hashes from i on (2^L <= PW), and
// min(a[q], a[q + PW - 2^L]) is the window's: at PW = 10, 16 rolls and 48 comparisons per lane instead of 80 and 72.
// (Sixteen positions per lane share the rolls and doubling steps among twice the windows, but at 70 registers and 32 KiB
// of staging per workgroup the kernel came out slower, 11.0 against 10.4 ms per 50 M reads: it is not the instructions
// that bound it any more.)
// The smaller of two hashes in ONE instruction: a hash has at most 62 bits, so as the bit pattern of a double it is a
// non-negative number that is neither infinite nor NaN (a denormal below 2^52 — the kernels run with FP64 denormals kept,
// the default), and for those the order of the doubles is the order of the integers.  V_MIN_F64 runs at full rate on
// gfx950 and hands one of its operands back bit for bit; the integer form is a 64-bit compare and two selects.  (Inline
// assembly: llvm.minnum would first canonicalize both operands, two instructions more.)
__device__ __forceinline__ uint64_t min_hash(uint64_t x, uint64_t y) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(__builtin_bit_cast(double, x)), "v"(__builtin_bit_cast(double, y)));
  return __builtin_bit_cast(uint64_t, r);
}

template <int PW>
__global__ __launch_bounds__(BNPK_BLOCK) void wf_minimizer_kernel(const uint64_t* __restrict__ W, int64_t n_words,
                                                                  const uint8_t* __restrict__ mask8, int64_t n_bases, int k,
                                                                  const int64_t* __restrict__ tile_off,
                                                                  int64_t* __restrict__ out) {
  constexpr int H = WF_ITEMS + PW - 1;                       // k-mers under the lane's eight windows
  constexpr int L = PW >= 16 ? 4 : PW >= 8 ? 3 : PW >= 4 ? 2 : 1;
  static_assert(PW >= 2 && PW <= WF_MAX_PER_WINDOW && (1 << L) <= PW, "window of 2 .. 26 k-mers");
  __shared__ uint64_t stage[WF_TILE];
  __shared__ unsigned wsum[BNPK_BLOCK / 64];
  const int64_t o = (int64_t)blockIdx.x * WF_TILE + (int64_t)threadIdx.x * WF_ITEMS;
  const int64_t run_first = tile_off[blockIdx.x], run_end = tile_off[blockIdx.x + 1];   // (asked for now, needed last)
  const unsigned v = o < n_bases ? mask8[o >> 3] : 0u;
  const unsigned cnt = __popc(v);
  const unsigned inc = wave_inclusive_scan(cnt);
  if (lane_id() == 63) wsum[wave_id()] = inc;
  uint64_t a[H];
  if (v) {
    const int64_t wi = o >> 5;
    const uint64_t w0 = W[wi], w1 = W[wi + 1], w2 = wi + 2 < n_words ? W[wi + 2] : 0;
    const int sh0 = 2 * (int)(o & 31), top = 2 * k - 2;
    const uint64_t kmask = (1ull << (2 * k)) - 1ull;
    a[0] = wf_window(w0, w1, w2, sh0) & kmask;              // k-mer at position o
    const uint64_t next = wf_window(w0, w1, w2, sh0 + 2 * k);   // the bases that enter the k-mers at o+1, o+2, ... (2 (H - 1) <= 64 bits)
#pragma unroll
    for (int i = 1; i < H; ++i) a[i] = (a[i - 1] >> 2) | (((next >> (2 * (i - 1))) & 3ull) << top);
    // (positions past a read's end enter hashes that no marked window uses: a window marked at o + q has PW k-mers of its
    // read ahead, and only its own a[q .. q + PW - 1] reach its minimum)
#pragma unroll
    for (int j = 0; j < L; ++j) {
#pragma unroll
      for (int i = 0; i + (1 << j) < H; ++i) a[i] = min_hash(a[i + (1 << j)], a[i]);
    }
  }
  __syncthreads();
  unsigned rank = inc - cnt;
  for (int w = 0; w < wave_id(); ++w) rank += wsum[w];
  if (v) {
#pragma unroll
    for (int q = 0; q < WF_ITEMS; ++q) {
      if ((v >> q) & 1u) {
        stage[rank++] = min_hash(a[q], a[q + PW - (1 << L)]);
      }
    }
  }
  __syncthreads();
  wf_store_run(stage, run_first, run_end, out);
}

// match_string (bionumpy/sequence/string_matcher.py:16-55): for every window of m symbols (marked in the start mask)
// 1 if it equals the pattern, else 0, in the ragged-flat order of the windows.  Same skeleton as wf_generate.
// PACKED: 2-bit symbols, the window is a k-mer hash compared with the pattern's; else bytes compared one by one.
constexpr int WF_MAX_PATTERN = 64;
struct wf_pattern { uint8_t b[WF_MAX_PATTERN]; };

constexpr int WF_MATCH_GROUP = 4;                           // tiles of 2048 positions a workgroup of wf_match takes, one after the other
template <bool PACKED>
__global__ __launch_bounds__(BNPK_BLOCK) void wf_match_kernel(const void* __restrict__ src, int64_t n_words,
                                                              const uint8_t* __restrict__ mask8, int64_t n_items, int m,
                                                              uint64_t pattern_hash, wf_pattern pat,
                                                              const int64_t* __restrict__ tile_off, int64_t n_tiles,
                                                              uint8_t* __restrict__ out) {
  // (the tile's flags are staged at the output's own offset within 16 bytes, so that both the LDS reads and the global stores
  // of the run are aligned 16-byte accesses: one store per lane and 16 flags instead of sixteen single-byte stores.  A tile is
  // 2 KB of output: a workgroup takes WF_MATCH_GROUP of them in a row — 3.7 M workgroups of two barriers each were
  // bound by their dispatch)
  __shared__ __attribute__((aligned(16))) uint8_t stage[WF_TILE + 32];
  __shared__ unsigned wsum[BNPK_BLOCK / 64];
  for (int g = 0; g < WF_MATCH_GROUP; ++g) {
    const int64_t tile = (int64_t)blockIdx.x * WF_MATCH_GROUP + g;
    if (tile >= n_tiles) break;                               // (uniform)
    const int64_t o = tile * WF_TILE + (int64_t)threadIdx.x * WF_ITEMS;
    const unsigned v = o < n_items ? mask8[o >> 3] : 0u;
    const unsigned cnt = __popc(v);
    const unsigned inc = wave_inclusive_scan(cnt);
    if (lane_id() == 63) wsum[wave_id()] = inc;
    unsigned hits = 0;                                        // bit q: the window at o + q matches
    if (v) {
      if (PACKED) {
        const uint64_t* W = reinterpret_cast<const uint64_t*>(src);
        const int64_t wi = o >> 5;
        const uint64_t w0 = W[wi], w1 = W[wi + 1], w2 = wi + 2 < n_words ? W[wi + 2] : 0;
        const int sh0 = 2 * (int)(o & 31), top = 2 * m - 2;
        const uint64_t kmask = (1ull << (2 * m)) - 1ull;
        uint64_t h = wf_window(w0, w1, w2, sh0) & kmask;
        const uint64_t next = wf_window(w0, w1, w2, sh0 + 2 * m);
#pragma unroll
        for (int q = 0; q < WF_ITEMS; ++q) {
          if (q) h = (h >> 2) | (((next >> (2 * (q - 1))) & 3ull) << top);
          hits |= (h == pattern_hash ? 1u : 0u) << q;
        }
      } else {
        const uint8_t* B = reinterpret_cast<const uint8_t*>(src);
#pragma unroll
        for (int q = 0; q < WF_ITEMS; ++q) {
          if ((v >> q) & 1u) {                                // (a marked position has m bytes of its row ahead)
            bool same = true;
            for (int j = 0; j < m; ++j) same = same && B[o + q + j] == pat.b[j];
            hits |= (same ? 1u : 0u) << q;
          }
        }
      }
    }
    __syncthreads();
    const int64_t base = tile_off[tile];
    const unsigned total = (unsigned)(tile_off[tile + 1] - base);
    const unsigned skew = (unsigned)(reinterpret_cast<uintptr_t>(out + base) & 15u);
    unsigned rank = skew + inc - cnt;
    for (int w = 0; w < wave_id(); ++w) rank += wsum[w];
    if (v) {
#pragma unroll
      for (int q = 0; q < WF_ITEMS; ++q)
        if ((v >> q) & 1u) stage[rank++] = (uint8_t)((hits >> q) & 1u);
    }
    __syncthreads();
    // stage[skew + i] -> out[base + i]: the bytes in front of the first 16-byte boundary, whole 16-byte groups, the rest
    uint8_t* dst = out + base - skew;                         // (16-byte aligned; its first `skew` bytes are the tile before's)
    const unsigned end = skew + total;
    const unsigned first16 = skew ? 16u : 0u, last16 = end & ~15u;
    if (threadIdx.x < 16u && threadIdx.x >= skew && threadIdx.x < min(first16, end)) dst[threadIdx.x] = stage[threadIdx.x];
    for (unsigned i = first16 + 16u * threadIdx.x; i + 16u <= last16; i += 16u * BNPK_BLOCK)
      *reinterpret_cast<uint4*>(dst + i) = *reinterpret_cast<const uint4*>(stage + i);
    if (last16 >= first16 && threadIdx.x < (end & 15u) && last16 + threadIdx.x >= skew) dst[last16 + threadIdx.x] = stage[last16 + threadIdx.x];
    __syncthreads();                                          // (the stage and the wave totals are the next tile's)
  }
}

// match_string on 2-bit DNA, one WAVEFRONT per tile of 2048 positions: a lane owns 32 positions — one packed word and the
// one behind it, 16 bytes loaded for 32 windows (the eight-positions-per-lane kernel above loads 24 bytes for 8: its load
// instructions, not its 2 KB of output per tile, set its pace) —, ranks by a wave scan of the mask popcounts, stages its
// flags by predicated byte writes at the output's own offset within 16 bytes, and the wavefront stores the run as aligned
// 16-byte pieces.  No workgroup barrier: the four wavefronts of a workgroup work on four tiles of their own.
__global__ __launch_bounds__(BNPK_BLOCK) void wf_match32_kernel(const uint64_t* __restrict__ W, int64_t n_words,
                                                                const uint32_t* __restrict__ mask32, int64_t n_items, int m,
                                                                uint64_t pattern_hash, const int64_t* __restrict__ tile_off,
                                                                int64_t n_tiles, uint8_t* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) uint8_t stage_all[BNPK_BLOCK / 64][WF_TILE + 48];
  const int lane = lane_id();
  uint8_t* stage = stage_all[wave_id()];
  const int64_t tile = (int64_t)blockIdx.x * (BNPK_BLOCK / 64) + wave_id();
  if (tile >= n_tiles) return;                                // (uniform per wavefront; nothing below synchronises the workgroup)
  const int64_t o = tile * WF_TILE + (int64_t)lane * 32;      // first position of the lane: a word boundary of W and of the mask
  const int64_t wi = o >> 5;
  const uint32_t v = o < n_items ? mask32[wi] : 0u;
  const uint64_t w0 = (v && wi < n_words) ? W[wi] : 0ull, w1 = (v && wi + 1 < n_words) ? W[wi + 1] : 0ull;
  const uint64_t kmask = (1ull << (2 * m)) - 1ull;            // m <= 31
  uint32_t hits = 0;
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const uint64_t h = (q ? (w0 >> (2 * q)) | (w1 << (64 - 2 * q)) : w0) & kmask;
    hits |= (h == pattern_hash ? 1u : 0u) << q;
  }
  const unsigned cnt = __popc(v);
  const unsigned inc = wave_inclusive_scan(cnt);
  const int64_t base = tile_off[tile];
  const unsigned total = (unsigned)(tile_off[tile + 1] - base);
  const unsigned skew = (unsigned)(reinterpret_cast<uintptr_t>(out + base) & 15u);
  const unsigned rank = skew + inc - cnt;
#pragma unroll
  for (int q = 0; q < 32; ++q)
    if ((v >> q) & 1u) stage[rank + __popc(v & ((1u << q) - 1u))] = (uint8_t)((hits >> q) & 1u);
  __builtin_amdgcn_wave_barrier();                            // (the wavefront's LDS writes are in order before its reads)
  __builtin_amdgcn_s_waitcnt(0xc07f);                         // lgkmcnt(0)
  uint8_t* dst = out + base - skew;                           // (16-byte aligned; its first `skew` bytes are the tile before's)
  const unsigned end = skew + total;
  const unsigned first16 = skew ? 16u : 0u, last16 = end & ~15u;
  if ((unsigned)lane < 16u && (unsigned)lane >= skew && (unsigned)lane < min(first16, end)) dst[lane] = stage[lane];
  for (unsigned i = first16 + 16u * (unsigned)lane; i + 16u <= last16; i += 16u * 64u)
    *reinterpret_cast<uint4*>(dst + i) = *reinterpret_cast<const uint4*>(stage + i);
  if (last16 >= first16 && (unsigned)lane < (end & 15u) && last16 + (unsigned)lane >= skew) dst[last16 + lane] = stage[last16 + lane];
}

// Position weight matrix scores (bionumpy/sequence/position_weight_matrix.py:86-104,177-196): for every window of W
// bases score = ((0 + M[0][c0]) + M[1][c1]) + ... in double precision, in that order (the order numpy's
// `scores[:n-offset] += row[codes[offset:]]` accumulates in), so the result is bit-identical to the reference's.
constexpr int WF_MAX_PWM = 64;
struct wf_pwm { double m[WF_MAX_PWM][4]; };          // [position][code]

__global__ __launch_bounds__(BNPK_BLOCK) void wf_pwm_kernel(const uint64_t* __restrict__ W, int64_t n_words,
                                                            const uint8_t* __restrict__ mask8,
                                                            int64_t n_bases, int width, const wf_pwm* __restrict__ pwm,
                                                            const int64_t* __restrict__ tile_off,
                                                            double* __restrict__ out) {
  __shared__ double stage[WF_TILE];
  __shared__ double mat[WF_MAX_PWM][4];
  __shared__ unsigned wsum[BNPK_BLOCK / 64];
  for (int i = threadIdx.x; i < width * 4; i += BNPK_BLOCK) mat[i >> 2][i & 3] = pwm->m[i >> 2][i & 3];
  const int64_t o = (int64_t)blockIdx.x * WF_TILE + (int64_t)threadIdx.x * WF_ITEMS;
  const unsigned v = o < n_bases ? mask8[o >> 3] : 0u;
  const unsigned cnt = __popc(v);
  const unsigned inc = wave_inclusive_scan(cnt);
  if (lane_id() == 63) wsum[wave_id()] = inc;
  __syncthreads();
  unsigned rank = inc - cnt;
  for (int w = 0; w < wave_id(); ++w) rank += wsum[w];
  if (v) {
    // the lane's eight windows lie in four packed words: load them once, cut a 128-bit window per position
    const int64_t wi = o >> 5;
    const uint64_t w0 = W[wi], w1 = W[wi + 1], w2 = wi + 2 < n_words ? W[wi + 2] : 0, w3 = wi + 3 < n_words ? W[wi + 3] : 0;
    const int sh0 = 2 * (int)(o & 31);
#pragma unroll
    for (int q = 0; q < WF_ITEMS; ++q) {
      if ((v >> q) & 1u) {
        const int sh = sh0 + 2 * q;                           // < 128
        uint64_t lo = wf_window(w0, w1, w2, sh);              // bases 0..31 of the window
        uint64_t hi = wf_window(w1, w2, w3, sh);              // bases 32..63
        double score = 0.0;
        const int first = min(width, 32);
        for (int j = 0; j < first; ++j) { score += mat[j][lo & 3ull]; lo >>= 2; }
        for (int j = 32; j < width; ++j) { score += mat[j][hi & 3ull]; hi >>= 2; }
        stage[rank++] = score;
      }
    }
  }
  __syncthreads();
  wf_store_run(reinterpret_cast<const uint64_t*>(stage), tile_off[blockIdx.x], tile_off[blockIdx.x + 1],
               reinterpret_cast<int64_t*>(out));
}

// bits [off[r], off[r+1] - (k-1)) of the mask for every row r with at least k bases: the positions of the flat
// packed stream at which a k-mer starts.  One lane per row; a row touches a handful of 32-bit words.
__global__ void kmer_start_mask_kernel(const int64_t* __restrict__ off, int64_t n_rows, int k,
                                       unsigned* __restrict__ mask32) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; r < n_rows; r += stride) {
    const int64_t s = off[r], e = off[r + 1] - (k - 1);
    for (int64_t p = s; p < e;) {
      const int64_t w = p >> 5;
      const int lo = (int)(p & 31);
      const int hi = (int)min((int64_t)32, e - (w << 5));              // bits [lo, hi) of word w
      const unsigned bits = (hi >= 32 ? ~0u : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
      if (bits == ~0u) mask32[w] = bits; else atomicOr(&mask32[w], bits);   // a full word belongs to one row
      p = (w + 1) << 5;
    }
  }
}

// bit i set on the last element of every non-empty row: what the fused FASTQ decode writes as it goes, for rows that are
// given by their offsets (one atomic OR per row; bnpk_kmer_starts_from_ends turns it into the k-mer start mask with a
// bit-parallel windowed OR — together 3.4 -> ~1 ms per 50 M reads against the walk over every row's words above)
__global__ void row_end_mask_kernel(const int64_t* __restrict__ off, int64_t n_rows, unsigned* __restrict__ ends32) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; r < n_rows; r += stride) {
    const int64_t e = off[r + 1];
    if (e > off[r]) atomicOr(&ends32[(e - 1) >> 5], 1u << ((e - 1) & 31));
  }
}

// A8 for alphabets that are not 4 letters wide (KmerEncoder.__call__ over a sliding window view, sequence/kmers.py:17-27
// + rollable.py:46-66): hash = sum_j code[p + j] * A^j in wrapping int64 arithmetic, exactly what numpy's
// uint8-window.dot(int64 weights) gives.  One lane per output, the row found like in kmer_kernel; the k code bytes of
// neighbouring lanes overlap, so they come out of L1/L2.
__global__ __launch_bounds__(BNPK_BLOCK) void kmer_generic_kernel(const uint8_t* __restrict__ codes,
                                                                  const int64_t* __restrict__ in_off,
                                                                  const int64_t* __restrict__ out_off, int64_t n_rows,
                                                                  int64_t n_out, int k, int kmers_per_window,
                                                                  uint64_t alphabet_size,
                                                                  const int64_t* __restrict__ tile_rows,
                                                                  int64_t* __restrict__ out) {
  int64_t rr[2];
  const int64_t tile = (int64_t)blockIdx.x * TILE_OUT;
  if (tile >= n_out) return;
  tile_row_range(tile_rows, blockIdx.x, gridDim.x, n_rows, rr[0], rr[1]);
  for (int64_t o = tile + threadIdx.x; o < min(tile + (int64_t)TILE_OUT, n_out); o += BNPK_BLOCK) {
    const int64_t row = find_row(out_off, rr[0], rr[1], o);
    const uint8_t* src = codes + in_off[row] + (o - out_off[row]);
    // kmers_per_window > 1: the minimizer of the window — the smallest of its hashes AS numpy compares them, i.e. as
    // signed 64-bit integers (Minimizers.__call__: kmer_hashes.raw().min(axis=-1), sequence/minimizers.py:15-17)
    int64_t best = 0;
    for (int m = 0; m < kmers_per_window; ++m) {
      uint64_t h = 0, w = 1;
      for (int j = 0; j < k; ++j) {
        h += (uint64_t)src[m + j] * w;
        w *= alphabet_size;
      }
      best = (m == 0 || (int64_t)h < best) ? (int64_t)h : best;
    }
    out[o] = best;
  }
}

// match_string(...).sum(axis=-1) / .any(axis=-1) on 2-bit DNA without the flags (string_matcher.py:16-55 of the reference
// returns the flags and its callers reduce them per row): counts[r] = windows of row r equal to the pattern.  One lane per
// row — the rows of neighbouring lanes are neighbours in the packed words, so a wavefront's loads cover one contiguous
// stretch — 32 windows per step, bit-parallel: position j of the pattern against the bases j further on, for all 32
// windows at once on the even bits.  Rows of more than MR_LONG bases are taken by the whole wavefront, a piece per lane.
constexpr int MR_LONG = 4096, MR_PIECE = 1024;

// windows [p, p + n_win) of m bases, p any base index: how many equal the pattern (m <= 31)
__device__ __forceinline__ unsigned mr_count(const uint64_t* __restrict__ W, int64_t n_words, int64_t p, int64_t n_win, int m,
                                             uint64_t pattern_hash) {
  if (n_win <= 0) return 0u;
  int64_t wi = p >> 5;
  const int sh = 2 * (int)(p & 31);
  auto word = [&](int64_t i) -> uint64_t { return i < n_words ? W[i] : 0ull; };
  uint64_t w0 = word(wi), w1 = word(wi + 1);
  auto funnel = [&](uint64_t lo, uint64_t hi) -> uint64_t { return sh ? (lo >> sh) | (hi << (64 - sh)) : lo; };
  uint64_t A = funnel(w0, w1);                                   // bases p .. p + 31
  unsigned n = 0;
  for (int64_t done = 0; done < n_win; done += 32) {
    const uint64_t w2 = word(wi + 2);
    uint64_t B = funnel(w1, w2);                                 // the 32 bases after A's
    uint32_t a0 = (uint32_t)A, a1 = (uint32_t)(A >> 32), b0 = (uint32_t)B, b1 = (uint32_t)(B >> 32);
    uint32_t e0 = 0x55555555u, e1 = 0x55555555u;
    uint64_t h = pattern_hash;
    for (int j = 0; j < m; ++j) {
      const uint32_t rep = (uint32_t)(h & 3u) * 0x55555555u;
      const uint32_t x0 = a0 ^ rep, x1 = a1 ^ rep;
      e0 &= ~(x0 | (x0 >> 1));
      e1 &= ~(x1 | (x1 >> 1));
      a0 = __builtin_amdgcn_alignbit(a1, a0, 2);                 // (a1:a0:b1:b0 as one 128-bit number) >>= 2
      a1 = __builtin_amdgcn_alignbit(b0, a1, 2);
      b0 = __builtin_amdgcn_alignbit(b1, b0, 2);
      b1 >>= 2;
      h >>= 2;
    }
    const int64_t left = n_win - done;
    if (left < 32) {
      const uint64_t keep = (1ull << (2 * left)) - 1ull;
      e0 &= (uint32_t)keep;
      e1 &= (uint32_t)(keep >> 32);
    }
    n += __popc(e0) + __popc(e1);
    A = B;
    w1 = w2;
    ++wi;
  }
  return n;
}

__global__ __launch_bounds__(BNPK_BLOCK) void match_rows_kernel(const uint64_t* __restrict__ W, int64_t n_words,
                                                               const int64_t* __restrict__ off, int64_t n_rows, int m,
                                                               uint64_t pattern_hash, int64_t* __restrict__ counts) {
  const int lane = lane_id();
  const int64_t n_groups = (n_rows + 63) / 64;
  for (int64_t g = (int64_t)blockIdx.x * (BNPK_BLOCK / 64) + wave_id(); g < n_groups; g += (int64_t)gridDim.x * (BNPK_BLOCK / 64)) {
    const int64_t r = g * 64 + lane;
    const int64_t lo = r < n_rows ? off[r] : 0, hi = r < n_rows ? off[r + 1] : 0;
    const int64_t n_win = hi - lo - (m - 1);
    const bool is_long = hi - lo > MR_LONG;
    int64_t n = is_long ? 0 : (int64_t)mr_count(W, n_words, lo, n_win, m, pattern_hash);
    uint64_t todo = __ballot(is_long);
    while (todo) {                                             // (uniform: every lane sees the same ballot)
      const int src = __builtin_ctzll(todo);
      todo &= todo - 1;
      const int64_t rlo = __shfl((long long)lo, src), rwin = __shfl((long long)n_win, src);
      int64_t part = 0;
      for (int64_t c = (int64_t)lane * MR_PIECE; c < rwin; c += 64 * MR_PIECE)
        part += mr_count(W, n_words, rlo + c, min((int64_t)MR_PIECE, rwin - c), m, pattern_hash);
      part = __shfl(wave_reduce_sum(part), 0);
      if (lane == src) n = part;
    }
    if (r < n_rows) counts[r] = n;
  }
}

}  // namespace

extern "C" {

int bnpk_row_end_mask(bnpk_ctx* ctx, const int64_t* d_offsets, int64_t n_rows, int64_t total, uint64_t* d_ends, void* stream) {
  if (!ctx || n_rows < 0 || total < 0 || !d_ends || (n_rows > 0 && !d_offsets)) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  bnpk_timer t(ctx, "row_end_mask", s);
  BNPK_HIP(ctx, hipMemsetAsync(d_ends, 0, (size_t)(total / 64 + 2) * 8, s));
  if (n_rows > 0)
    hipLaunchKernelGGL(row_end_mask_kernel, dim3(grid_for(ceil_div(n_rows, 256))), dim3(256), 0, s, d_offsets, n_rows,
                       reinterpret_cast<unsigned*>(d_ends));
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_kmer_start_mask(bnpk_ctx* ctx, const int64_t* d_offsets, int64_t n_rows, int64_t total, int k,
                         uint64_t* d_mask, void* stream) {
  if (!ctx || n_rows < 0 || total < 0 || k < 1 || !d_mask || (n_rows > 0 && !d_offsets)) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  bnpk_timer t(ctx, "kmer_start_mask", s);
  BNPK_HIP(ctx, hipMemsetAsync(d_mask, 0, (size_t)(total / 64 + 2) * 8, s));
  if (n_rows > 0)
    hipLaunchKernelGGL(kmer_start_mask_kernel, dim3(grid_for(ceil_div(n_rows, 256))), dim3(256), 0, s, d_offsets,
                       n_rows, k, reinterpret_cast<unsigned*>(d_mask));
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_windows_flat(bnpk_ctx* ctx, const uint64_t* d_packed, const uint64_t* d_start_mask, int64_t n_bases, int k,
                      int kmers_per_window, int64_t n_out, int64_t* d_out, void* stream) {
  if (!ctx || k < 1 || k > 31 || kmers_per_window < 1 || n_bases < 0 || n_out < 0) return BNPK_ERR_ARG;
  if (kmers_per_window > WF_MAX_PER_WINDOW) return BNPK_ERR_RANGE;
  if (n_out == 0 || n_bases == 0) return BNPK_OK;
  if (!d_packed || !d_start_mask || !d_out) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int64_t n_tiles = ceil_div(n_bases, WF_TILE), n_mask_words = ceil_div(n_bases, 64);
  if (n_tiles > BNPK_MAX_BLOCKS) return BNPK_ERR_RANGE;
  void* scratch = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, (size_t)(n_tiles + 1) * 8 + bnpk_scan_scratch_bytes(n_tiles), &scratch, (hipStream_t)stream));
  int64_t* tile_off = (int64_t*)scratch;
  int64_t* scan_scratch = tile_off + n_tiles + 1;
  bnpk_timer t(ctx, kmers_per_window == 1 ? "kmers_flat" : "minimizers_flat", s);
  hipLaunchKernelGGL(wf_count_kernel, dim3((unsigned)ceil_div(n_mask_words, BNPK_BLOCK)), dim3(BNPK_BLOCK), 0, s,
                     d_start_mask, n_mask_words, n_tiles, tile_off);
  BNPK_HIP(ctx, hipGetLastError());
  BNPK_CHECK(bnpk_scan_launch(ctx, tile_off, n_tiles, 1, tile_off, true, scan_scratch, s));
  const uint8_t* mask8 = reinterpret_cast<const uint8_t*>(d_start_mask);
#define WF_MIN_CASE(PW)                                                                                                    \
  case PW:                                                                                                                 \
    hipLaunchKernelGGL(wf_minimizer_kernel<PW>, dim3((unsigned)n_tiles), dim3(BNPK_BLOCK), 0, s, d_packed, n_bases / 32 + 2, \
                       mask8, n_bases, k, (const int64_t*)tile_off, d_out);                                                \
    break;
  switch (kmers_per_window) {
    WF_MIN_CASE(2) WF_MIN_CASE(3) WF_MIN_CASE(4) WF_MIN_CASE(5) WF_MIN_CASE(6) WF_MIN_CASE(7) WF_MIN_CASE(8) WF_MIN_CASE(9)
    WF_MIN_CASE(10) WF_MIN_CASE(11) WF_MIN_CASE(12) WF_MIN_CASE(13) WF_MIN_CASE(14) WF_MIN_CASE(15) WF_MIN_CASE(16)
    WF_MIN_CASE(17) WF_MIN_CASE(18) WF_MIN_CASE(19) WF_MIN_CASE(20) WF_MIN_CASE(21) WF_MIN_CASE(22) WF_MIN_CASE(23)
    WF_MIN_CASE(24) WF_MIN_CASE(25) WF_MIN_CASE(26)
    default:                                                   // one k-mer per window: the k-mers themselves
      hipLaunchKernelGGL(wf_generate_kernel, dim3((unsigned)n_tiles), dim3(BNPK_BLOCK), 0, s, d_packed, n_bases / 32 + 2, mask8,
                         n_bases, k, kmers_per_window, (const int64_t*)tile_off, d_out);
  }
#undef WF_MIN_CASE
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

static int match_windows(bnpk_ctx* ctx, bool packed, const void* d_src, const uint64_t* d_start_mask, int64_t n_items, int m,
                         uint64_t pattern_hash, const uint8_t* h_pattern, int64_t n_out, uint8_t* d_out, void* stream) {
  if (n_out == 0 || n_items == 0) return BNPK_OK;
  if (!d_src || !d_start_mask || !d_out) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int64_t n_tiles = ceil_div(n_items, WF_TILE), n_mask_words = ceil_div(n_items, 64);
  if (n_tiles > BNPK_MAX_BLOCKS) return BNPK_ERR_RANGE;
  void* scratch = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, (size_t)(n_tiles + 1) * 8 + bnpk_scan_scratch_bytes(n_tiles), &scratch, (hipStream_t)stream));
  int64_t* tile_off = (int64_t*)scratch;
  int64_t* scan_scratch = tile_off + n_tiles + 1;
  wf_pattern pat;
  memset(&pat, 0, sizeof(pat));
  if (h_pattern) memcpy(pat.b, h_pattern, (size_t)m);
  bnpk_timer t(ctx, packed ? "match_windows_packed" : "match_windows_bytes", s);
  hipLaunchKernelGGL(wf_count_kernel, dim3((unsigned)ceil_div(n_mask_words, BNPK_BLOCK)), dim3(BNPK_BLOCK), 0, s,
                     d_start_mask, n_mask_words, n_tiles, tile_off);
  BNPK_HIP(ctx, hipGetLastError());
  BNPK_CHECK(bnpk_scan_launch(ctx, tile_off, n_tiles, 1, tile_off, true, scan_scratch, s));
  if (packed)
    hipLaunchKernelGGL(wf_match32_kernel, dim3((unsigned)ceil_div(n_tiles, BNPK_BLOCK / 64)), dim3(BNPK_BLOCK), 0, s,
                       reinterpret_cast<const uint64_t*>(d_src), n_items / 32 + 2, reinterpret_cast<const uint32_t*>(d_start_mask),
                       n_items, m, pattern_hash, (const int64_t*)tile_off, n_tiles, d_out);
  else
    hipLaunchKernelGGL((wf_match_kernel<false>), dim3((unsigned)ceil_div(n_tiles, WF_MATCH_GROUP)), dim3(BNPK_BLOCK), 0, s, d_src, (int64_t)0,
                       reinterpret_cast<const uint8_t*>(d_start_mask), n_items, m, pattern_hash, pat,
                       (const int64_t*)tile_off, n_tiles, d_out);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_pwm_scores(bnpk_ctx* ctx, const uint64_t* d_packed, const uint64_t* d_start_mask, int64_t n_bases, int width,
                    const double* h_matrix, int64_t n_out, double* d_out, void* stream) {
  if (!ctx || width < 1 || width > WF_MAX_PWM || n_bases < 0 || n_out < 0 || !h_matrix) return BNPK_ERR_ARG;
  if (n_out == 0 || n_bases == 0) return BNPK_OK;
  if (!d_packed || !d_start_mask || !d_out) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int64_t n_tiles = ceil_div(n_bases, WF_TILE), n_mask_words = ceil_div(n_bases, 64);
  if (n_tiles > BNPK_MAX_BLOCKS) return BNPK_ERR_RANGE;
  void* scratch = nullptr;
  const size_t pwm_bytes = (sizeof(wf_pwm) + 63) & ~(size_t)63;
  BNPK_CHECK(bnpk_scratch(ctx, pwm_bytes + (size_t)(n_tiles + 1) * 8 + bnpk_scan_scratch_bytes(n_tiles), &scratch, (hipStream_t)stream));
  wf_pwm* d_pwm = (wf_pwm*)scratch;
  int64_t* tile_off = (int64_t*)((char*)scratch + pwm_bytes);
  int64_t* scan_scratch = tile_off + n_tiles + 1;
  BNPK_HIP(ctx, hipMemcpyAsync(d_pwm, h_matrix, (size_t)width * 4 * sizeof(double), hipMemcpyHostToDevice, s));
  BNPK_HIP(ctx, hipStreamSynchronize(s));                      // (h_matrix is the caller's pageable memory)
  bnpk_timer t(ctx, "pwm_scores", s);
  hipLaunchKernelGGL(wf_count_kernel, dim3((unsigned)ceil_div(n_mask_words, BNPK_BLOCK)), dim3(BNPK_BLOCK), 0, s,
                     d_start_mask, n_mask_words, n_tiles, tile_off);
  BNPK_HIP(ctx, hipGetLastError());
  BNPK_CHECK(bnpk_scan_launch(ctx, tile_off, n_tiles, 1, tile_off, true, scan_scratch, s));
  hipLaunchKernelGGL(wf_pwm_kernel, dim3((unsigned)n_tiles), dim3(BNPK_BLOCK), 0, s, d_packed, n_bases / 32 + 2,
                     reinterpret_cast<const uint8_t*>(d_start_mask), n_bases, width, (const wf_pwm*)d_pwm,
                     (const int64_t*)tile_off, d_out);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_match_windows_packed(bnpk_ctx* ctx, const uint64_t* d_packed, const uint64_t* d_start_mask, int64_t n_bases, int m,
                              uint64_t pattern_hash, int64_t n_out, uint8_t* d_out, void* stream) {
  if (!ctx || m < 1 || m > 31 || n_bases < 0 || n_out < 0) return BNPK_ERR_ARG;
  return match_windows(ctx, true, d_packed, d_start_mask, n_bases, m, pattern_hash, nullptr, n_out, d_out, stream);
}

int bnpk_match_rows_packed(bnpk_ctx* ctx, const uint64_t* d_packed, int64_t n_bases, const int64_t* d_offsets, int64_t n_rows,
                           int m, uint64_t pattern_hash, int64_t* d_counts, void* stream) {
  if (!ctx || m < 1 || m > 31 || n_bases < 0 || n_rows < 0) return BNPK_ERR_ARG;
  if (n_rows == 0) return BNPK_OK;
  if (!d_offsets || !d_counts || (n_bases > 0 && !d_packed)) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  bnpk_timer t(ctx, "match_rows_packed", s);
  hipLaunchKernelGGL(match_rows_kernel, dim3(grid_for(ceil_div(n_rows, (int64_t)BNPK_BLOCK))), dim3(BNPK_BLOCK), 0, s, d_packed,
                     ceil_div(n_bases, (int64_t)32), d_offsets, n_rows, m, pattern_hash, d_counts);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_match_windows_bytes(bnpk_ctx* ctx, const uint8_t* d_bytes, const uint64_t* d_start_mask, int64_t n_bytes, int m,
                             const uint8_t* h_pattern, int64_t n_out, uint8_t* d_out, void* stream) {
  if (!ctx || m < 1 || m > WF_MAX_PATTERN || n_bytes < 0 || n_out < 0 || !h_pattern) return BNPK_ERR_ARG;
  return match_windows(ctx, false, d_bytes, d_start_mask, n_bytes, m, 0, h_pattern, n_out, d_out, stream);
}

int bnpk_kmers(bnpk_ctx* ctx, const uint64_t* d_packed, const int64_t* d_in_offsets, const int64_t* d_out_offsets,
               int64_t n_rows, int64_t n_out, int k, int64_t* d_hashes, void* stream) {
  if (!ctx || k < 1 || k > 31 || n_rows < 0 || n_out < 0) return BNPK_ERR_ARG;
  if (n_out == 0) return BNPK_OK;
  if (!d_packed || !d_in_offsets || !d_out_offsets || !d_hashes || n_rows == 0) return BNPK_ERR_ARG;
  int64_t blocks = ceil_div(n_out, TILE_OUT);
  if (blocks > BNPK_MAX_BLOCKS) return BNPK_ERR_RANGE;
  hipStream_t s = (hipStream_t)stream;
  void* table = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, tile_rows_bytes(blocks), &table, (hipStream_t)stream));
  bnpk_timer t(ctx, "kmers", s);
  BNPK_CHECK(build_tile_rows(ctx, d_out_offsets, n_rows, TILE_OUT, (int64_t*)table, s));
  hipLaunchKernelGGL((kmer_kernel<false>), dim3((unsigned)blocks), dim3(BNPK_BLOCK), 0, s, d_packed, d_in_offsets,
                     d_out_offsets, n_rows, n_out, k, 1, (const int64_t*)table, d_hashes);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_kmers_generic(bnpk_ctx* ctx, const uint8_t* d_codes, const int64_t* d_in_offsets, const int64_t* d_out_offsets,
                       int64_t n_rows, int64_t n_out, int k, int alphabet_size, int64_t* d_hashes, void* stream) {
  if (!ctx || k < 1 || k > 31 || alphabet_size < 1 || alphabet_size > 255 || n_rows < 0 || n_out < 0) return BNPK_ERR_ARG;
  if (n_out == 0) return BNPK_OK;
  if (!d_codes || !d_in_offsets || !d_out_offsets || !d_hashes || n_rows == 0) return BNPK_ERR_ARG;
  int64_t blocks = ceil_div(n_out, TILE_OUT);
  if (blocks > BNPK_MAX_BLOCKS) return BNPK_ERR_RANGE;
  hipStream_t s = (hipStream_t)stream;
  void* table = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, tile_rows_bytes(blocks), &table, (hipStream_t)stream));
  bnpk_timer t(ctx, "kmers_generic", s);
  BNPK_CHECK(build_tile_rows(ctx, d_out_offsets, n_rows, TILE_OUT, (int64_t*)table, s));
  hipLaunchKernelGGL(kmer_generic_kernel, dim3((unsigned)blocks), dim3(BNPK_BLOCK), 0, s, d_codes, d_in_offsets,
                     d_out_offsets, n_rows, n_out, k, 1, (uint64_t)alphabet_size, (const int64_t*)table, d_hashes);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_minimizers_generic(bnpk_ctx* ctx, const uint8_t* d_codes, const int64_t* d_in_offsets, const int64_t* d_out_offsets,
                            int64_t n_rows, int64_t n_out, int k, int window_size, int alphabet_size, int64_t* d_out,
                            void* stream) {
  if (!ctx || k < 1 || window_size < k || alphabet_size < 1 || alphabet_size > 255 || n_rows < 0 || n_out < 0) return BNPK_ERR_ARG;
  if (n_out == 0) return BNPK_OK;
  if (!d_codes || !d_in_offsets || !d_out_offsets || !d_out || n_rows == 0) return BNPK_ERR_ARG;
  int64_t blocks = ceil_div(n_out, TILE_OUT);
  if (blocks > BNPK_MAX_BLOCKS) return BNPK_ERR_RANGE;
  hipStream_t s = (hipStream_t)stream;
  void* table = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, tile_rows_bytes(blocks), &table, (hipStream_t)stream));
  bnpk_timer t(ctx, "minimizers_generic", s);
  BNPK_CHECK(build_tile_rows(ctx, d_out_offsets, n_rows, TILE_OUT, (int64_t*)table, s));
  hipLaunchKernelGGL(kmer_generic_kernel, dim3((unsigned)blocks), dim3(BNPK_BLOCK), 0, s, d_codes, d_in_offsets,
                     d_out_offsets, n_rows, n_out, k, window_size - k + 1, (uint64_t)alphabet_size, (const int64_t*)table,
                     d_out);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_minimizers(bnpk_ctx* ctx, const uint64_t* d_packed, const int64_t* d_in_offsets,
                    const int64_t* d_out_offsets, int64_t n_rows, int64_t n_out, int k, int window_size,
                    int64_t* d_out, void* stream) {
  if (!ctx || k < 1 || k > 31 || window_size < k || n_rows < 0 || n_out < 0) return BNPK_ERR_ARG;
  if (n_out == 0) return BNPK_OK;
  if (!d_packed || !d_in_offsets || !d_out_offsets || !d_out || n_rows == 0) return BNPK_ERR_ARG;
  int64_t blocks = ceil_div(n_out, TILE_OUT);
  if (blocks > BNPK_MAX_BLOCKS) return BNPK_ERR_RANGE;
  hipStream_t s = (hipStream_t)stream;
  void* table = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, tile_rows_bytes(blocks), &table, (hipStream_t)stream));
  bnpk_timer t(ctx, "minimizers", s);
  BNPK_CHECK(build_tile_rows(ctx, d_out_offsets, n_rows, TILE_OUT, (int64_t*)table, s));
  hipLaunchKernelGGL((kmer_kernel<true>), dim3((unsigned)blocks), dim3(BNPK_BLOCK), 0, s, d_packed, d_in_offsets,
                     d_out_offsets, n_rows, n_out, k, window_size - k + 1, (const int64_t*)table, d_out);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_row_ids(bnpk_ctx* ctx, const int64_t* d_offsets, int64_t n_rows, int64_t n, int64_t* d_rows,
                 void* stream) {
  if (!ctx || n_rows < 0 || n < 0) return BNPK_ERR_ARG;
  if (n == 0) return BNPK_OK;
  if (!d_offsets || !d_rows || n_rows == 0) return BNPK_ERR_ARG;
  int64_t blocks = ceil_div(n, TILE_OUT);
  if (blocks > BNPK_MAX_BLOCKS) return BNPK_ERR_RANGE;
  hipStream_t s = (hipStream_t)stream;
  void* table = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, tile_rows_bytes(blocks), &table, (hipStream_t)stream));
  bnpk_timer t(ctx, "row_ids", s);
  BNPK_CHECK(build_tile_rows(ctx, d_offsets, n_rows, TILE_OUT, (int64_t*)table, s));
  hipLaunchKernelGGL(row_ids_kernel, dim3((unsigned)blocks), dim3(BNPK_BLOCK), 0, s, d_offsets, n_rows, n,
                     (const int64_t*)table, d_rows);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

}  // extern "C"
