// k-mer generation fused with the first radix pass of the histogram sort (A8 + A9, sparse path).
// Instead of materialising the k-mer hashes in row order (kmers.hip) and letting the sort read them back,
// this pair of kernels generates every k-mer from the packed reads twice — once to count 8-bit digits per
// slab, once to scatter — and writes each hash exactly once, already partitioned by the chosen digit.
// The first pass of an LSD radix sort has no prior order to preserve, so ranks come from plain LDS atomics
// (no stable multi-split needed).  Tiles of 8192 k-mers are staged through 64 KiB of LDS so that every
// bucket's keys leave the CU as one contiguous run (32 keys = 256 B on average).
#include <algorithm>

#include "common.h"
#include "kmer_gen.h"
#include "rows.h"
#include "scan.h"

namespace {

constexpr int PT_THREADS = 1024;
constexpr int PT_ITEMS = 8;                          // consecutive k-mers per lane
constexpr int PT_TILE = PT_THREADS * PT_ITEMS;       // 8192
constexpr int PT_BINS = 256;

// the PT_ITEMS k-mers starting at output index o (row cursor walk + funnel shifts), o < n_out
__device__ __forceinline__ int gen_items(const uint64_t* __restrict__ W, const int64_t* __restrict__ in_off,
                                         const int64_t* __restrict__ out_off, int64_t rlo, int64_t rhi, int64_t o,
                                         int64_t hi, uint64_t mask, uint64_t v[PT_ITEMS]) {
  row_cursor c = seek_row(in_off, out_off, rlo, rhi, o);
  word_window ww;
  int cnt = 0;
#pragma unroll
  for (int q = 0; q < PT_ITEMS; ++q) {
    int64_t oo = o + q;
    if (oo >= hi) break;
    if (q) next_output(c, in_off, out_off, oo);
    v[q] = bits_at(W, c.in_pos, ww) & mask;
    ++cnt;
  }
  return cnt;
}

__global__ __launch_bounds__(PT_THREADS) void gen_hist_kernel(const uint64_t* __restrict__ W,
                                                              const int64_t* __restrict__ in_off,
                                                              const int64_t* __restrict__ out_off, int64_t n_rows,
                                                              int64_t n_out, int k, int shift, int64_t slab,
                                                              const int64_t* __restrict__ tile_rows, int64_t n_tiles,
                                                              int64_t* __restrict__ H, int nb) {
  __shared__ unsigned h[PT_BINS];
  if (threadIdx.x < PT_BINS) h[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t mask = (1ull << (2 * k)) - 1ull;
  const int64_t lo = (int64_t)blockIdx.x * slab, hi = min(lo + slab, n_out);
  for (int64_t t0 = lo; t0 < hi; t0 += PT_TILE) {
    int64_t rlo, rhi;
    tile_row_range(tile_rows, t0 / PT_TILE, n_tiles, n_rows, rlo, rhi);
    int64_t o = t0 + (int64_t)threadIdx.x * PT_ITEMS;
    if (o < hi) {
      uint64_t v[PT_ITEMS];
      int cnt = gen_items(W, in_off, out_off, rlo, rhi, o, hi, mask, v);
#pragma unroll
      for (int q = 0; q < PT_ITEMS; ++q)
        if (q < cnt) atomicAdd(&h[(v[q] >> shift) & (PT_BINS - 1)], 1u);
    }
  }
  __syncthreads();
  if (threadIdx.x < PT_BINS) H[(int64_t)threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(PT_THREADS) void gen_scatter_kernel(const uint64_t* __restrict__ W,
                                                                 const int64_t* __restrict__ in_off,
                                                                 const int64_t* __restrict__ out_off, int64_t n_rows,
                                                                 int64_t n_out, int k, int shift, int64_t slab,
                                                                 const int64_t* __restrict__ tile_rows,
                                                                 int64_t n_tiles, const int64_t* __restrict__ offs,
                                                                 int nb, uint64_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* stage = reinterpret_cast<uint64_t*>(smem);                  // PT_TILE keys
  int64_t* cursor = reinterpret_cast<int64_t*>(stage + PT_TILE);       // PT_BINS global write cursors
  unsigned* cnt = reinterpret_cast<unsigned*>(cursor + PT_BINS);        // PT_BINS per-tile counts
  unsigned* start = cnt + PT_BINS;                                       // PT_BINS per-tile bucket starts
  unsigned* wsum = start + PT_BINS;                                      // 4 wave totals
  const int tid = threadIdx.x;
  if (tid < PT_BINS) { cursor[tid] = offs[(int64_t)tid * nb + blockIdx.x]; cnt[tid] = 0; }
  __syncthreads();
  const uint64_t mask = (1ull << (2 * k)) - 1ull;
  const int64_t lo = (int64_t)blockIdx.x * slab, hi = min(lo + slab, n_out);
  for (int64_t t0 = lo; t0 < hi; t0 += PT_TILE) {
    int64_t rlo, rhi;
    tile_row_range(tile_rows, t0 / PT_TILE, n_tiles, n_rows, rlo, rhi);
    uint64_t v[PT_ITEMS];
    unsigned r[PT_ITEMS];
    int n_items = 0;
    int64_t o = t0 + (int64_t)tid * PT_ITEMS;
    if (o < hi) {
      n_items = gen_items(W, in_off, out_off, rlo, rhi, o, hi, mask, v);
#pragma unroll
      for (int q = 0; q < PT_ITEMS; ++q)
        if (q < n_items) r[q] = atomicAdd(&cnt[(v[q] >> shift) & (PT_BINS - 1)], 1u);   // rank inside the tile's bucket
    }
    __syncthreads();
    // exclusive scan of the 256 bucket counts: the first 4 wavefronts hold one bin per lane
    unsigned c = (tid < PT_BINS) ? cnt[tid] : 0;
    unsigned inc = wave_inclusive_scan(c);
    if (tid < PT_BINS && (tid & 63) == 63) wsum[tid >> 6] = inc;
    __syncthreads();
    if (tid < PT_BINS) {
      unsigned base = 0;
      for (int w = 0; w < (tid >> 6); ++w) base += wsum[w];
      start[tid] = base + inc - c;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PT_ITEMS; ++q)
      if (q < n_items) stage[start[(v[q] >> shift) & (PT_BINS - 1)] + r[q]] = v[q];
    __syncthreads();
    const int m = (int)min((int64_t)PT_TILE, hi - t0);
    for (int i = tid; i < m; i += PT_THREADS) {   // bucket-contiguous runs leave the CU coalesced
      uint64_t key = stage[i];
      unsigned d = (unsigned)(key >> shift) & (PT_BINS - 1);
      out[cursor[d] + (i - (int)start[d])] = key;
    }
    __syncthreads();
    if (tid < PT_BINS) { cursor[tid] += cnt[tid]; cnt[tid] = 0; }
    __syncthreads();
  }
}

constexpr size_t PT_LDS = (size_t)PT_TILE * 8 + PT_BINS * 8 + PT_BINS * 4 * 2 + 64;

}  // namespace

extern "C" {

int bnpk_kmers_partition(bnpk_ctx* ctx, const uint64_t* d_packed, const int64_t* d_in_offsets,
                         const int64_t* d_out_offsets, int64_t n_rows, int64_t n_out, int k, int digit_shift,
                         int64_t* d_out, void* stream) {
  if (!ctx || k < 1 || k > 31 || n_rows < 0 || n_out < 0 || digit_shift < 0 || digit_shift > 2 * k - 1)
    return BNPK_ERR_ARG;
  if (n_out == 0) return BNPK_OK;
  if (!d_packed || !d_in_offsets || !d_out_offsets || !d_out || n_rows == 0) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int64_t n_tiles = ceil_div(n_out, PT_TILE);
  const int64_t slab_tiles = std::max<int64_t>(1, ceil_div(n_tiles, 4096));
  const int64_t slab = slab_tiles * PT_TILE;
  const int64_t nb = ceil_div(n_out, slab);
  if (nb > BNPK_MAX_BLOCKS / 4) return BNPK_ERR_RANGE;
  const int64_t hn = (int64_t)PT_BINS * nb;
  const size_t table_bytes = (tile_rows_bytes(n_tiles) + 63) & ~(size_t)63;
  void* scratch = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, table_bytes + (size_t)hn * 8 + bnpk_scan_scratch_bytes(hn), &scratch));
  int64_t* table = (int64_t*)scratch;
  int64_t* H = (int64_t*)((char*)scratch + table_bytes);
  int64_t* scan_scratch = H + hn;
  static bool attr_set = false;
  if (!attr_set) {
    BNPK_HIP(ctx, hipFuncSetAttribute((const void*)gen_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)PT_LDS));
    attr_set = true;
  }
  {
    bnpk_timer t(ctx, "kmers_partition_hist", s);
    BNPK_CHECK(build_tile_rows(ctx, d_out_offsets, n_rows, PT_TILE, table, s));
    hipLaunchKernelGGL(gen_hist_kernel, dim3((unsigned)nb), dim3(PT_THREADS), 0, s, d_packed, d_in_offsets,
                       d_out_offsets, n_rows, n_out, k, digit_shift, slab, (const int64_t*)table, n_tiles, H, (int)nb);
    BNPK_HIP(ctx, hipGetLastError());
    BNPK_CHECK(bnpk_scan_launch(ctx, H, hn, 1, H, false, scan_scratch, s));
  }
  bnpk_timer t(ctx, "kmers_partition_scatter", s);
  hipLaunchKernelGGL(gen_scatter_kernel, dim3((unsigned)nb), dim3(PT_THREADS), PT_LDS, s, d_packed, d_in_offsets,
                     d_out_offsets, n_rows, n_out, k, digit_shift, slab, (const int64_t*)table, n_tiles,
                     (const int64_t*)H, (int)nb, reinterpret_cast<uint64_t*>(d_out));
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

}  // extern "C"
