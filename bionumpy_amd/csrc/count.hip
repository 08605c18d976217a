// Counting (A9): dense LDS/global-atomic histograms for small k, sort + run-length for k up to 31,
// plus the sorted-search used by the k-mer index (A12).
#include "common.h"
#include "scan.h"

#include <algorithm>
#include <rocprim/rocprim.hpp>

namespace {

// ------------------------------------------------------------------------------------ dense histogram
constexpr int LDS_MAX_BINS = 16384;     // 64 KiB of int32 bins per workgroup (k <= 7)

__global__ __launch_bounds__(BNPK_BLOCK) void hist_lds_kernel(const int64_t* __restrict__ v, int64_t n,
                                                              int n_bins, int64_t per_block,
                                                              unsigned long long* __restrict__ hist) {
  extern __shared__ __attribute__((aligned(16))) unsigned int bins[];
  for (int i = threadIdx.x; i < n_bins; i += BNPK_BLOCK) bins[i] = 0;
  __syncthreads();
  int64_t begin = (int64_t)blockIdx.x * per_block;          // per_block is a multiple of 2
  int64_t end = min(begin + per_block, n);
  for (int64_t i = begin + 2 * (int64_t)threadIdx.x; i < end; i += 2 * BNPK_BLOCK) {
    int64_t a, b = -1;
    if (i + 1 < end) {
      longlong2 p = *reinterpret_cast<const longlong2*>(v + i);
      a = p.x; b = p.y;
    } else {
      a = v[i];
    }
    if ((uint64_t)a < (uint64_t)n_bins) atomicAdd(&bins[a], 1u);
    if ((uint64_t)b < (uint64_t)n_bins) atomicAdd(&bins[b], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n_bins; i += BNPK_BLOCK) {
    unsigned int c = bins[i];
    if (c) atomicAdd(&hist[i], (unsigned long long)c);
  }
}

__global__ __launch_bounds__(BNPK_BLOCK) void hist_global_kernel(const int64_t* __restrict__ v, int64_t n,
                                                                 int64_t n_bins,
                                                                 unsigned long long* __restrict__ hist) {
  int64_t i = (int64_t)blockIdx.x * BNPK_BLOCK + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * BNPK_BLOCK;
  for (; i < n; i += stride) {
    int64_t a = v[i];
    if ((uint64_t)a < (uint64_t)n_bins) atomicAdd(&hist[a], 1ull);
  }
}

__global__ __launch_bounds__(BNPK_BLOCK) void hist_rows_kernel(const int64_t* __restrict__ v,
                                                               const int64_t* __restrict__ off, int64_t n_rows,
                                                               int64_t n, int64_t n_bins,
                                                               unsigned long long* __restrict__ hist) {
  int64_t i = (int64_t)blockIdx.x * BNPK_BLOCK + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * BNPK_BLOCK;
  for (; i < n; i += stride) {
    int64_t row = find_row(off, 0, n_rows - 1, i);
    int64_t a = v[i];
    if ((uint64_t)a < (uint64_t)n_bins) atomicAdd(&hist[row * n_bins + a], 1ull);
  }
}

// ------------------------------------------------------------------------------------ histograms of LETTERS (uint8 codes)
// count_encoded / np.bincount over the codes of an encoded array (bionumpy/sequence/count_encoded.py:166-182,
// encoded_array.py:463-466) without widening them to int64 first.  Few bins (<= 8: every alphabet of the sequence path but
// amino acids): a lane keeps eight BYTE counters in one 64-bit register — a code c adds 1 << 8c — and spills them into
// eight 32-bit counters every 15 vectors (15 x 16 = 240 < 256); no atomics, no LDS, ~70 vector instructions per 16 bytes.
// A word with a byte >= 8 (never in an encoded array; the bound is checked like np.bincount's minlength would) takes the
// byte-by-byte path.  The wave totals leave as eight 64-bit global atomics per wavefront.
__device__ __forceinline__ void byte_counters_add(uint64_t& acc, uint32_t x, int n_bins) {
  if (x & 0xF8F8F8F8u) {                                     // some byte is not a code below 8
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t c = (x >> (8 * j)) & 0xffu;
      if (c < (uint32_t)n_bins) acc += 1ull << (8 * c);
    }
    return;
  }
  const uint32_t sh = (x & 0x07070707u) << 3;
#pragma unroll
  for (int j = 0; j < 4; ++j) acc += 1ull << ((sh >> (8 * j)) & 0xffu);
}

__device__ __forceinline__ void byte_counters_spill(uint64_t& acc, uint32_t (&cnt)[8]) {
#pragma unroll
  for (int b = 0; b < 8; ++b) cnt[b] += (uint32_t)(acc >> (8 * b)) & 0xffu;
  acc = 0;
}

__global__ __launch_bounds__(BNPK_BLOCK) void hist_bytes_small_kernel(const uint8_t* __restrict__ v, int64_t n, int n_bins,
                                                                      unsigned long long* __restrict__ hist) {
  // [head | 16-byte vectors | tail]: the vectors start at the first 16-byte boundary of the buffer
  const int64_t head = min<int64_t>(n, (16 - ((uintptr_t)v & 15)) & 15);
  const int64_t n_vec = (n - head) >> 4;
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  const u32x4_t* __restrict__ vec = reinterpret_cast<const u32x4_t*>(v + head);
  uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t acc = 0;
  int pending = 0;
  const int64_t stride = (int64_t)gridDim.x * BNPK_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BNPK_BLOCK + threadIdx.x; i < n_vec; i += stride) {
    const u32x4_t q = __builtin_nontemporal_load(vec + i);
    byte_counters_add(acc, q.x, n_bins);
    byte_counters_add(acc, q.y, n_bins);
    byte_counters_add(acc, q.z, n_bins);
    byte_counters_add(acc, q.w, n_bins);
    if (++pending == 15) { byte_counters_spill(acc, cnt); pending = 0; }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {                 // the unaligned ends: at most 30 bytes
    for (int64_t i = 0; i < head; ++i) {
      if (++pending == 200) { byte_counters_spill(acc, cnt); pending = 0; }
      if (v[i] < n_bins) acc += 1ull << (8 * v[i]);
    }
    for (int64_t i = head + (n_vec << 4); i < n; ++i) {
      if (++pending == 200) { byte_counters_spill(acc, cnt); pending = 0; }
      if (v[i] < n_bins) acc += 1ull << (8 * v[i]);
    }
  }
  byte_counters_spill(acc, cnt);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const uint32_t t = wave_reduce_sum(cnt[b]);
    if (lane_id() == 0 && t && b < n_bins) atomicAdd(&hist[b], (unsigned long long)t);
  }
}

// up to 256 bins: a histogram per wavefront in LDS
__global__ __launch_bounds__(BNPK_BLOCK) void hist_bytes_lds_kernel(const uint8_t* __restrict__ v, int64_t n, int n_bins,
                                                                    unsigned long long* __restrict__ hist) {
  __shared__ unsigned int bins[BNPK_BLOCK / 64][256];
  for (int i = threadIdx.x; i < (BNPK_BLOCK / 64) * 256; i += BNPK_BLOCK) (&bins[0][0])[i] = 0;
  __syncthreads();
  unsigned int* mine = bins[wave_id()];
  const int64_t stride = (int64_t)gridDim.x * BNPK_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BNPK_BLOCK + threadIdx.x; i < n; i += stride) {
    const uint32_t c = v[i];
    if (c < (uint32_t)n_bins) atomicAdd(&mine[c], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n_bins; i += BNPK_BLOCK) {
    unsigned long long t = 0;
#pragma unroll
    for (int w = 0; w < BNPK_BLOCK / 64; ++w) t += bins[w][i];
    if (t) atomicAdd(&hist[i], t);
  }
}

// the four letters of packed 2-bit DNA (32 bases per word): with l = the low bits and h = the high bits of the codes,
// #3 = popc(l & h), #1 = popc(l) - #3, #2 = popc(h) - #3, and #0 is what is left of n_bases — 1.9 GB read per 50 M reads
// instead of the 7.5 GB of their unpacked codes
__global__ __launch_bounds__(BNPK_BLOCK) void hist_packed2_kernel(const uint64_t* __restrict__ w, int64_t n_bases,
                                                                  unsigned long long* __restrict__ hist) {
  const int64_t n_words = (n_bases + 31) >> 5;
  const int tail = (int)(n_bases & 31);
  const int64_t stride = (int64_t)gridDim.x * BNPK_BLOCK;
  uint64_t c1 = 0, c2 = 0, c3 = 0;
  for (int64_t i = (int64_t)blockIdx.x * BNPK_BLOCK + threadIdx.x; i < n_words; i += stride) {
    uint64_t x = w[i];
    if (tail && i == n_words - 1) x &= (1ull << (2 * tail)) - 1ull;    // (whatever lies behind the last base)
    const uint64_t l = x & 0x5555555555555555ull, h = (x >> 1) & 0x5555555555555555ull;
    const int p3 = __popcll(l & h);
    c3 += p3;
    c1 += __popcll(l) - p3;
    c2 += __popcll(h) - p3;
  }
  c1 = wave_reduce_sum(c1);
  c2 = wave_reduce_sum(c2);
  c3 = wave_reduce_sum(c3);
  if (lane_id() == 0) {
    if (c1) atomicAdd(&hist[1], (unsigned long long)c1);
    if (c2) atomicAdd(&hist[2], (unsigned long long)c2);
    if (c3) atomicAdd(&hist[3], (unsigned long long)c3);
    // hist[0] += this wavefront's bases - c1 - c2 - c3 is added by the host call as n_bases - (the three others): one
    // subtraction over the finished counters (hist_packed2_zero_kernel)
  }
}

__global__ void hist_packed2_zero_kernel(unsigned long long* __restrict__ hist, const unsigned long long* __restrict__ before,
                                         int64_t n_bases) {
  // #0 = n_bases - (what this call added to the other three)
  if (threadIdx.x == 0 && blockIdx.x == 0)
    hist[0] += (unsigned long long)n_bases - ((hist[1] - before[1]) + (hist[2] - before[2]) + (hist[3] - before[3]));
}

// one histogram PER ROW of ragged uint8 codes (count_encoded(ragged, axis=-1): count_encoded.py:176-182), n_bins <= 8:
// four lanes per row, 16 bytes per lane and step, the byte counters of the four added up by two DPP steps; the row's
// counts leave as n_bins 64-bit stores — a row belongs to one group, nothing is atomic
__global__ __launch_bounds__(BNPK_BLOCK) void hist_bytes_rows_kernel(const uint8_t* __restrict__ v, const int64_t* __restrict__ off,
                                                                     int64_t n_rows, int64_t total, int n_bins,
                                                                     int64_t* __restrict__ hist) {
  const int q = threadIdx.x & 3;
  const int64_t stride = (int64_t)gridDim.x * (BNPK_BLOCK / 4);
  for (int64_t r = (int64_t)blockIdx.x * (BNPK_BLOCK / 4) + (threadIdx.x >> 2); r < ((n_rows + 15) & ~(int64_t)15); r += stride) {
    // (all four lanes of a group — and whole wavefronts — stay in step for the shuffles: rows beyond the end are empty)
    const int64_t s = r < n_rows ? off[r] : 0, e = r < n_rows ? off[r + 1] : 0;
    uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t acc = 0;
    int pending = 0;
    for (int64_t p = s + 16 * q; p < e; p += 64) {
      if (p + 16 <= e && p + 16 <= total) {
        uint4 x;
        __builtin_memcpy(&x, v + p, 16);
        byte_counters_add(acc, x.x, n_bins);
        byte_counters_add(acc, x.y, n_bins);
        byte_counters_add(acc, x.z, n_bins);
        byte_counters_add(acc, x.w, n_bins);
      } else {
        for (int64_t i = p; i < e; ++i)
          if (v[i] < n_bins) acc += 1ull << (8 * v[i]);
      }
      if (++pending == 15) { byte_counters_spill(acc, cnt); pending = 0; }
    }
    byte_counters_spill(acc, cnt);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      uint32_t t = cnt[b];
      t += __shfl_xor(t, 1);
      t += __shfl_xor(t, 2);
      if (q == 0 && b < n_bins && r < n_rows) hist[r * n_bins + b] = (int64_t)t;
    }
  }
}

// ------------------------------------------------------------------------------------ runs of equal keys
constexpr int RUN_ITEMS = 8;
constexpr int RUN_TILE = BNPK_BLOCK * RUN_ITEMS;

__device__ __forceinline__ bool is_head(const int64_t* __restrict__ a, const int64_t* __restrict__ b, int64_t i) {
  if (i == 0) return true;
  if (a[i] != a[i - 1]) return true;
  return b != nullptr && b[i] != b[i - 1];
}

__global__ __launch_bounds__(BNPK_BLOCK) void run_census_kernel(const int64_t* __restrict__ a,
                                                                const int64_t* __restrict__ b, int64_t n,
                                                                int64_t* __restrict__ tile_counts) {
  __shared__ int smem[BNPK_BLOCK / 64];
  int64_t base = (int64_t)blockIdx.x * RUN_TILE + (int64_t)threadIdx.x * RUN_ITEMS;
  int c = 0;
#pragma unroll
  for (int j = 0; j < RUN_ITEMS; ++j) {
    int64_t i = base + j;
    if (i < n && is_head(a, b, i)) ++c;
  }
  c = wave_reduce_sum(c);
  if (lane_id() == 0) smem[wave_id()] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < BNPK_BLOCK / 64; ++w) t += smem[w];
    tile_counts[blockIdx.x] = t;
  }
}

// The heads of a tile leave through LDS: a lane's heads are neighbours in the output, its neighbours' heads are not, and three
// arrays written lane by lane were 0.65 ms for the 12 M (k-mer, row) pairs of a yeast genome's index, where nearly every pair is
// a head (the census next to it, which reads the same way and writes nothing: 0.07).  Staged, the tile's heads go out as one
// contiguous run per array.
__global__ __launch_bounds__(BNPK_BLOCK) void run_heads_kernel(const int64_t* __restrict__ a,
                                                               const int64_t* __restrict__ b, int64_t n,
                                                               const int64_t* __restrict__ tile_offsets,
                                                               int64_t n_runs, int64_t* __restrict__ keys_out,
                                                               int64_t* __restrict__ second_out,
                                                               int64_t* __restrict__ run_starts) {
  __shared__ int smem[BNPK_BLOCK / 64 + 1];
  __shared__ int64_t stage[RUN_TILE];
  const int64_t base = (int64_t)blockIdx.x * RUN_TILE + (int64_t)threadIdx.x * RUN_ITEMS;
  unsigned flags = 0;
  int c = 0;
  int64_t va[RUN_ITEMS], vb[RUN_ITEMS];
#pragma unroll
  for (int j = 0; j < RUN_ITEMS; ++j) {
    const int64_t i = base + j;
    va[j] = vb[j] = 0;
    if (i < n) {
      va[j] = a[i];
      if (b) vb[j] = b[i];
      if (is_head(a, b, i)) { flags |= 1u << j; ++c; }
    }
  }
  int total;
  const int ex = block_exclusive_scan(c, smem, &total);
  const int64_t out0 = tile_offsets[blockIdx.x];
  auto leave = [&](int64_t* __restrict__ dst, int which) {
    int r = ex;
#pragma unroll
    for (int j = 0; j < RUN_ITEMS; ++j)
      if (flags & (1u << j)) stage[r++] = which == 0 ? va[j] : which == 1 ? vb[j] : base + j;
    __syncthreads();
    for (int t = threadIdx.x; t < total; t += BNPK_BLOCK) dst[out0 + t] = stage[t];
    __syncthreads();
  };
  leave(keys_out, 0);
  if (second_out) leave(second_out, 1);
  leave(run_starts, 2);
  if (blockIdx.x == 0 && threadIdx.x == 0) run_starts[n_runs] = n;
}

__global__ void run_sums_kernel(const int64_t* __restrict__ run_starts, int64_t n_runs,
                                const int64_t* __restrict__ prefix, int64_t* __restrict__ counts) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; j < n_runs; j += stride) {
    int64_t s = run_starts[j], e = run_starts[j + 1];
    counts[j] = prefix ? prefix[e] - prefix[s] : e - s;
  }
}

__global__ void search_sorted_kernel(const int64_t* __restrict__ sorted, int64_t n, const int64_t* __restrict__ q,
                                     int64_t m, int upper, int64_t* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < m; i += stride) {
    int64_t key = q[i];
    int64_t lo = 0, hi = n;
    while (lo < hi) {
      int64_t mid = lo + ((hi - lo) >> 1);
      int64_t v = sorted[mid];
      bool right = upper ? (v <= key) : (v < key);
      if (right) lo = mid + 1; else hi = mid;
    }
    out[i] = lo;
  }
}

// A12 without a key-value sort: the distinct (kmer, row) pairs of KmerIndex.create_index (kmer_indexing.py:24-47) are the
// distinct values of ONE integer per occurrence, id = rank(kmer) * n_rows + row with rank = index of the k-mer among the
// sorted distinct k-mers — ascending ids are ascending (kmer, row) — so the index is built by the sparse counting path
// twice (k-mers, then ids) and two element-wise kernels.
__global__ void pair_compose_kernel(const int64_t* __restrict__ rank, const int64_t* __restrict__ rows, int64_t n, int64_t n_rows,
                                    int64_t* __restrict__ ids) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) ids[i] = rank[i] * n_rows + rows[i];
}

__global__ void pair_split_kernel(const int64_t* __restrict__ ids, int64_t n, int64_t n_rows, const int64_t* __restrict__ keys,
                                  int64_t* __restrict__ keys_out, int64_t* __restrict__ rows_out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const int64_t id = ids[i], r = id / n_rows;
    keys_out[i] = keys[r];
    rows_out[i] = id - r * n_rows;
  }
}


// ---- weighted counts: np.bincount(values, weights=w) (bionumpy/sequence/count_encoded.py:166-187) ------------------------
// hist[r][values[r * vstride + i]] += w[r * wstride + i]: one histogram per row r — of the weights (2-D weights over flat
// values: vstride 0), of the values (a matrix of values under 1-D weights: wstride 0), or one row altogether.  T = int64_t
// for integer / bool weights (exact), double for floating-point ones.  Few bins: a private histogram in LDS per
// workgroup (64-bit LDS atomics), flushed with global atomics; many bins: global atomics (the bins then see little
// contention).  blockIdx.y = row.
constexpr int HW_LDS_BINS = 4096;
template <typename T>
__device__ __forceinline__ void hw_add(T* p, T v) { atomicAdd(p, v); }
template <>
__device__ __forceinline__ void hw_add<int64_t>(int64_t* p, int64_t v) {
  atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v);
}
template <typename T, bool LDS>
__global__ __launch_bounds__(BNPK_BLOCK) void hist_weighted_kernel(const int64_t* __restrict__ values, const T* __restrict__ weights,
                                                                   int64_t n, int64_t vstride, int64_t wstride, int64_t n_bins,
                                                                   T* __restrict__ hist, unsigned long long* __restrict__ bad) {
  __shared__ T local[LDS ? HW_LDS_BINS : 1];
  const int64_t r = blockIdx.y;
  const int64_t* v = values + r * vstride;
  const T* w = weights + r * wstride;
  T* out = hist + r * n_bins;
  if (LDS) {
    for (int64_t c = threadIdx.x; c < n_bins; c += BNPK_BLOCK) local[c] = T(0);
    __syncthreads();
  }
  const int64_t stride = (int64_t)gridDim.x * BNPK_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BNPK_BLOCK + threadIdx.x; i < n; i += stride) {
    const int64_t b = v[i];
    if (b < 0 || b >= n_bins) { atomicOr(bad, 1ull); continue; }
    hw_add<T>(LDS ? &local[b] : &out[b], w[i]);
  }
  if (LDS) {
    __syncthreads();
    for (int64_t c = threadIdx.x; c < n_bins; c += BNPK_BLOCK)
      if (local[c] != T(0)) hw_add<T>(&out[c], local[c]);
  }
}

}  // namespace

extern "C" {

int bnpk_count_bytes(bnpk_ctx* ctx, const uint8_t* d_values, int64_t n, int n_bins, int64_t* d_hist, void* stream) {
  if (!ctx || n < 0 || n_bins < 1 || n_bins > 256 || !d_hist) return BNPK_ERR_ARG;
  if (n == 0) return BNPK_OK;
  if (!d_values) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  auto* hist = reinterpret_cast<unsigned long long*>(d_hist);
  bnpk_timer t(ctx, "count_bytes", s);
  if (n_bins <= 8) {
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>(ceil_div(n, 16 * BNPK_BLOCK), (int64_t)ctx->compute_units * 8));
    hipLaunchKernelGGL(hist_bytes_small_kernel, dim3((unsigned)blocks), dim3(BNPK_BLOCK), 0, s, d_values, n, n_bins, hist);
  } else {
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>(ceil_div(n, 8 * BNPK_BLOCK), (int64_t)ctx->compute_units * 8));
    hipLaunchKernelGGL(hist_bytes_lds_kernel, dim3((unsigned)blocks), dim3(BNPK_BLOCK), 0, s, d_values, n, n_bins, hist);
  }
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_count_packed2(bnpk_ctx* ctx, const uint64_t* d_packed, int64_t n_bases, int64_t* d_hist4, void* stream) {
  if (!ctx || n_bases < 0 || !d_hist4) return BNPK_ERR_ARG;
  if (n_bases == 0) return BNPK_OK;
  if (!d_packed) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  void* before = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, 4 * sizeof(int64_t), &before, s));
  BNPK_HIP(ctx, hipMemcpyAsync(before, d_hist4, 4 * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
  auto* hist = reinterpret_cast<unsigned long long*>(d_hist4);
  const int64_t n_words = (n_bases + 31) >> 5;
  const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>(ceil_div(n_words, 4 * BNPK_BLOCK), (int64_t)ctx->compute_units * 8));
  bnpk_timer t(ctx, "count_packed2", s);
  hipLaunchKernelGGL(hist_packed2_kernel, dim3((unsigned)blocks), dim3(BNPK_BLOCK), 0, s, d_packed, n_bases, hist);
  hipLaunchKernelGGL(hist_packed2_zero_kernel, dim3(1), dim3(64), 0, s, hist, (const unsigned long long*)before, n_bases);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_count_bytes_rows(bnpk_ctx* ctx, const uint8_t* d_values, const int64_t* d_offsets, int64_t n_rows, int64_t total,
                          int n_bins, int64_t* d_hist, void* stream) {
  if (!ctx || n_rows < 0 || total < 0 || n_bins < 1 || n_bins > 8 || !d_hist || !d_offsets) return BNPK_ERR_ARG;
  if (n_rows == 0) return BNPK_OK;
  if (total > 0 && !d_values) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  bnpk_timer t(ctx, "count_bytes_rows", s);
  const int64_t blocks = ceil_div(n_rows, BNPK_BLOCK / 4);
  hipLaunchKernelGGL(hist_bytes_rows_kernel, dim3(grid_for(blocks)), dim3(BNPK_BLOCK), 0, s, d_values, d_offsets, n_rows, total,
                     n_bins, d_hist);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_count_dense(bnpk_ctx* ctx, const int64_t* d_values, int64_t n, int64_t n_bins, int64_t* d_hist,
                     void* stream) {
  if (!ctx || n < 0 || n_bins < 1 || !d_hist) return BNPK_ERR_ARG;
  if (n == 0) return BNPK_OK;
  if (!d_values) return BNPK_ERR_ARG;
  if (((uintptr_t)d_values & 15) != 0) return BNPK_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  auto* hist = reinterpret_cast<unsigned long long*>(d_hist);
  if (n_bins <= LDS_MAX_BINS) {
    // enough workgroups to fill 256 CUs a few times over, each streaming a contiguous slab
    int64_t blocks = std::min<int64_t>(ceil_div(n, 4 * 2 * BNPK_BLOCK), (int64_t)ctx->compute_units * 8);
    int64_t per_block = ceil_div(ceil_div(n, blocks), 2) * 2;
    while (per_block >= 0x7fffffffLL) { blocks *= 2; per_block = ceil_div(ceil_div(n, blocks), 2) * 2; }
    blocks = ceil_div(n, per_block);
    bnpk_timer t(ctx, "count_dense_lds", s);
    hipLaunchKernelGGL(hist_lds_kernel, dim3((unsigned)blocks), dim3(BNPK_BLOCK), (size_t)n_bins * sizeof(unsigned),
                       s, d_values, n, (int)n_bins, per_block, hist);
  } else {
    int64_t blocks = std::min<int64_t>(ceil_div(n, BNPK_BLOCK), (int64_t)ctx->compute_units * 16);
    bnpk_timer t(ctx, "count_dense_global", s);
    hipLaunchKernelGGL(hist_global_kernel, dim3((unsigned)blocks), dim3(BNPK_BLOCK), 0, s, d_values, n, n_bins, hist);
  }
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_count_dense_rows(bnpk_ctx* ctx, const int64_t* d_values, const int64_t* d_offsets, int64_t n_rows,
                          int64_t n, int64_t n_bins, int64_t* d_hist, void* stream) {
  if (!ctx || n < 0 || n_rows < 0 || n_bins < 1) return BNPK_ERR_ARG;
  if (n == 0) return BNPK_OK;
  if (!d_values || !d_offsets || !d_hist || n_rows == 0) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  bnpk_timer t(ctx, "count_dense_rows", s);
  hipLaunchKernelGGL(hist_rows_kernel, dim3(grid_for(ceil_div(n, BNPK_BLOCK))), dim3(BNPK_BLOCK), 0, s, d_values, d_offsets, n_rows, n,
                     n_bins, reinterpret_cast<unsigned long long*>(d_hist));
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_count_weighted(bnpk_ctx* ctx, const int64_t* d_values, const void* d_weights, int weights_f64, int64_t n,
                        int64_t n_rows, int64_t value_stride, int64_t weight_stride, int64_t n_bins, void* d_hist,
                        int* h_out_of_range, void* stream) {
  if (!ctx || n < 0 || n_rows < 1 || n_bins < 1 || value_stride < 0 || weight_stride < 0 || !h_out_of_range || !d_hist)
    return BNPK_ERR_ARG;
  *h_out_of_range = 0;
  if (n_rows > 65535) return BNPK_ERR_RANGE;              // (gridDim.y)
  if (n == 0) return BNPK_OK;
  if (!d_values || !d_weights) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  void* flag = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, 64, &flag, s));
  BNPK_HIP(ctx, hipMemsetAsync(flag, 0, 8, s));
  {
    bnpk_timer t(ctx, "count_weighted", s);
    const unsigned gx = grid_for(std::min<int64_t>(ceil_div(n, BNPK_BLOCK), std::max<int64_t>(1, (int64_t)ctx->compute_units * 16 / n_rows)));
    const dim3 grid(gx, (unsigned)n_rows);
    unsigned long long* bad = reinterpret_cast<unsigned long long*>(flag);
    const bool lds = n_bins <= HW_LDS_BINS;
    if (weights_f64) {
      const double* w = reinterpret_cast<const double*>(d_weights);
      double* h = reinterpret_cast<double*>(d_hist);
      if (lds) hipLaunchKernelGGL((hist_weighted_kernel<double, true>), grid, dim3(BNPK_BLOCK), 0, s, d_values, w, n, value_stride, weight_stride, n_bins, h, bad);
      else hipLaunchKernelGGL((hist_weighted_kernel<double, false>), grid, dim3(BNPK_BLOCK), 0, s, d_values, w, n, value_stride, weight_stride, n_bins, h, bad);
    } else {
      const int64_t* w = reinterpret_cast<const int64_t*>(d_weights);
      int64_t* h = reinterpret_cast<int64_t*>(d_hist);
      if (lds) hipLaunchKernelGGL((hist_weighted_kernel<int64_t, true>), grid, dim3(BNPK_BLOCK), 0, s, d_values, w, n, value_stride, weight_stride, n_bins, h, bad);
      else hipLaunchKernelGGL((hist_weighted_kernel<int64_t, false>), grid, dim3(BNPK_BLOCK), 0, s, d_values, w, n, value_stride, weight_stride, n_bins, h, bad);
    }
    BNPK_HIP(ctx, hipGetLastError());
  }
  unsigned long long host_flag = 0;
  BNPK_HIP(ctx, hipMemcpyAsync(&host_flag, flag, 8, hipMemcpyDeviceToHost, s));
  BNPK_HIP(ctx, hipStreamSynchronize(s));
  *h_out_of_range = host_flag ? 1 : 0;
  return BNPK_OK;
}

int bnpk_sort_keys(bnpk_ctx* ctx, int64_t* d_keys, int64_t* d_alt, int64_t n, int begin_bit, int end_bit,
                   int* h_in_alt, void* stream) {
  if (!ctx || n < 0 || begin_bit < 0 || end_bit <= begin_bit || end_bit > 64 || !h_in_alt) return BNPK_ERR_ARG;
  *h_in_alt = 0;
  if (n <= 1) return BNPK_OK;
  if (!d_keys || !d_alt) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  rocprim::double_buffer<uint64_t> keys(reinterpret_cast<uint64_t*>(d_keys), reinterpret_cast<uint64_t*>(d_alt));
  size_t temp_bytes = 0;
  BNPK_HIP(ctx, rocprim::radix_sort_keys(nullptr, temp_bytes, keys, (size_t)n, (unsigned)begin_bit,
                                         (unsigned)end_bit, s));
  void* temp = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, temp_bytes, &temp, (hipStream_t)stream));
  bnpk_timer t(ctx, begin_bit == 0 ? "sort_keys" : "partition_keys", s);
  BNPK_HIP(ctx, rocprim::radix_sort_keys(temp, temp_bytes, keys, (size_t)n, (unsigned)begin_bit,
                                         (unsigned)end_bit, s));
  *h_in_alt = (keys.current() == reinterpret_cast<uint64_t*>(d_alt)) ? 1 : 0;
  return BNPK_OK;
}

int bnpk_sort_pairs(bnpk_ctx* ctx, int64_t* d_keys, int64_t* d_keys_alt, int64_t* d_vals, int64_t* d_vals_alt,
                    int64_t n, int key_bits, int* h_in_alt, void* stream) {
  if (!ctx || n < 0 || key_bits < 1 || key_bits > 64 || !h_in_alt) return BNPK_ERR_ARG;
  *h_in_alt = 0;
  if (n <= 1) return BNPK_OK;
  if (!d_keys || !d_keys_alt || !d_vals || !d_vals_alt) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  rocprim::double_buffer<uint64_t> keys(reinterpret_cast<uint64_t*>(d_keys), reinterpret_cast<uint64_t*>(d_keys_alt));
  rocprim::double_buffer<int64_t> vals(d_vals, d_vals_alt);
  size_t temp_bytes = 0;
  BNPK_HIP(ctx, rocprim::radix_sort_pairs(nullptr, temp_bytes, keys, vals, (size_t)n, 0u, (unsigned)key_bits, s));
  void* temp = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, temp_bytes, &temp, (hipStream_t)stream));
  bnpk_timer t(ctx, "sort_pairs", s);
  BNPK_HIP(ctx, rocprim::radix_sort_pairs(temp, temp_bytes, keys, vals, (size_t)n, 0u, (unsigned)key_bits, s));
  *h_in_alt = (keys.current() == reinterpret_cast<uint64_t*>(d_keys_alt)) ? 1 : 0;
  return BNPK_OK;
}

int64_t bnpk_run_tiles(int64_t n) { return n <= 0 ? 0 : ceil_div(n, RUN_TILE); }

int bnpk_run_census(bnpk_ctx* ctx, const int64_t* d_sorted, const int64_t* d_second, int64_t n,
                    int64_t* d_tile_offsets, int64_t* h_n_runs, void* stream) {
  if (!ctx || n < 0 || !d_tile_offsets || !h_n_runs || (n > 0 && !d_sorted)) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  int64_t tiles = bnpk_run_tiles(n);
  if (tiles > BNPK_MAX_BLOCKS) return BNPK_ERR_RANGE;
  void* scratch = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, bnpk_scan_scratch_bytes(tiles), &scratch, (hipStream_t)stream));
  {
    bnpk_timer t(ctx, "run_census", s);
    if (tiles > 0)
      hipLaunchKernelGGL(run_census_kernel, dim3((unsigned)tiles), dim3(BNPK_BLOCK), 0, s, d_sorted, d_second, n,
                         d_tile_offsets);
    BNPK_HIP(ctx, hipGetLastError());
    BNPK_CHECK(bnpk_scan_launch(ctx, d_tile_offsets, tiles, 1, d_tile_offsets, true, (int64_t*)scratch, s));
  }
  BNPK_HIP(ctx, hipMemcpyAsync(h_n_runs, d_tile_offsets + tiles, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  BNPK_HIP(ctx, hipStreamSynchronize(s));
  return BNPK_OK;
}

int bnpk_run_heads(bnpk_ctx* ctx, const int64_t* d_sorted, const int64_t* d_second, int64_t n,
                   const int64_t* d_tile_offsets, int64_t n_runs, int64_t* d_keys_out, int64_t* d_second_out,
                   int64_t* d_run_starts, void* stream) {
  if (!ctx || n < 0 || n_runs < 0 || !d_run_starts) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) return bnpk_fill_i64(ctx, d_run_starts, 1, 0, stream);
  if (!d_sorted || !d_tile_offsets || !d_keys_out || (d_second_out && !d_second)) return BNPK_ERR_ARG;
  if (bnpk_run_tiles(n) > BNPK_MAX_BLOCKS) return BNPK_ERR_RANGE;
  bnpk_timer t(ctx, "run_heads", s);
  hipLaunchKernelGGL(run_heads_kernel, dim3((unsigned)bnpk_run_tiles(n)), dim3(BNPK_BLOCK), 0, s, d_sorted, d_second, n,
                     d_tile_offsets, n_runs, d_keys_out, d_second_out, d_run_starts);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_run_sums(bnpk_ctx* ctx, const int64_t* d_run_starts, int64_t n_runs, const int64_t* d_weight_prefix,
                  int64_t* d_counts, void* stream) {
  if (!ctx || n_runs < 0) return BNPK_ERR_ARG;
  if (n_runs == 0) return BNPK_OK;
  if (!d_run_starts || !d_counts) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  bnpk_timer t(ctx, "run_sums", s);
  hipLaunchKernelGGL(run_sums_kernel, dim3(grid_for(ceil_div(n_runs, 256))), dim3(256), 0, s, d_run_starts, n_runs,
                     d_weight_prefix, d_counts);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_search_sorted(bnpk_ctx* ctx, const int64_t* d_sorted, int64_t n, const int64_t* d_queries, int64_t m,
                       int upper, int64_t* d_out, void* stream) {
  if (!ctx || n < 0 || m < 0) return BNPK_ERR_ARG;
  if (m == 0) return BNPK_OK;
  if (!d_queries || !d_out || (n > 0 && !d_sorted)) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  bnpk_timer t(ctx, "search_sorted", s);
  hipLaunchKernelGGL(search_sorted_kernel, dim3(grid_for(ceil_div(m, 256))), dim3(256), 0, s, d_sorted, n, d_queries, m,
                     upper, d_out);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_pair_compose(bnpk_ctx* ctx, const int64_t* d_rank, const int64_t* d_rows, int64_t n, int64_t n_rows, int64_t* d_ids,
                      void* stream) {
  if (!ctx || n < 0 || n_rows < 1) return BNPK_ERR_ARG;
  if (n == 0) return BNPK_OK;
  if (!d_rank || !d_rows || !d_ids) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  bnpk_timer t(ctx, "pair_compose", s);
  hipLaunchKernelGGL(pair_compose_kernel, dim3(grid_for(std::min<int64_t>(ceil_div(n, 256), (int64_t)ctx->compute_units * 32))),
                     dim3(256), 0, s, d_rank, d_rows, n, n_rows, d_ids);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_pair_split(bnpk_ctx* ctx, const int64_t* d_ids, int64_t n, int64_t n_rows, const int64_t* d_sorted_keys,
                    int64_t* d_keys_out, int64_t* d_rows_out, void* stream) {
  if (!ctx || n < 0 || n_rows < 1) return BNPK_ERR_ARG;
  if (n == 0) return BNPK_OK;
  if (!d_ids || !d_sorted_keys || !d_keys_out || !d_rows_out) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  bnpk_timer t(ctx, "pair_split", s);
  hipLaunchKernelGGL(pair_split_kernel, dim3(grid_for(std::min<int64_t>(ceil_div(n, 256), (int64_t)ctx->compute_units * 32))),
                     dim3(256), 0, s, d_ids, n, n_rows, d_sorted_keys, d_keys_out, d_rows_out);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

}  // extern "C"
