// 2-bit k-mer hashes (A8), minimizers (A11) and ragged row ids for gfx950.
// Integer / HBM-write bound: 8 output bytes per k-mer against 0.25 input bytes per base.  Output-flat
// mapping: every lane produces two adjacent int64 (one 16-byte store, 1 KiB contiguous per wavefront
// instruction); the packed input words are shared between neighbouring lanes through L1.
#include "common.h"
#include "rows.h"
#include "kmer_gen.h"

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

}  // namespace

extern "C" {

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

int bnpk_kmers(bnpk_ctx* ctx, const uint64_t* d_packed, const int64_t* d_in_offsets, const int64_t* d_out_offsets,
               int64_t n_rows, int64_t n_out, int k, int64_t* d_hashes, void* stream) {
  if (!ctx || k < 1 || k > 31 || n_rows < 0 || n_out < 0) return BNPK_ERR_ARG;
  if (n_out == 0) return BNPK_OK;
  if (!d_packed || !d_in_offsets || !d_out_offsets || !d_hashes || n_rows == 0) return BNPK_ERR_ARG;
  int64_t blocks = ceil_div(n_out, TILE_OUT);
  if (blocks > BNPK_MAX_BLOCKS) return BNPK_ERR_RANGE;
  hipStream_t s = (hipStream_t)stream;
  void* table = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, tile_rows_bytes(blocks), &table));
  bnpk_timer t(ctx, "kmers", s);
  BNPK_CHECK(build_tile_rows(ctx, d_out_offsets, n_rows, TILE_OUT, (int64_t*)table, s));
  hipLaunchKernelGGL((kmer_kernel<false>), dim3((unsigned)blocks), dim3(BNPK_BLOCK), 0, s, d_packed, d_in_offsets,
                     d_out_offsets, n_rows, n_out, k, 1, (const int64_t*)table, d_hashes);
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
  BNPK_CHECK(bnpk_scratch(ctx, tile_rows_bytes(blocks), &table));
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
  BNPK_CHECK(bnpk_scratch(ctx, tile_rows_bytes(blocks), &table));
  bnpk_timer t(ctx, "row_ids", s);
  BNPK_CHECK(build_tile_rows(ctx, d_offsets, n_rows, TILE_OUT, (int64_t*)table, s));
  hipLaunchKernelGGL(row_ids_kernel, dim3((unsigned)blocks), dim3(BNPK_BLOCK), 0, s, d_offsets, n_rows, n,
                     (const int64_t*)table, d_rows);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

}  // extern "C"
