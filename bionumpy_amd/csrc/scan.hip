// int64 exclusive scans (row offsets, tile offsets) — reduce / scan-partials / apply, 2048 items per block.
#include "scan.h"

namespace {

constexpr int ITEMS = 8;
constexpr int TILE = BNPK_BLOCK * ITEMS;

// rows after `ragged[..., :-(window-1)]` keep max(0, len - (window-1)) elements
__device__ __forceinline__ int64_t xform(int64_t v, int window) {
  if (window > 1) { v -= (window - 1); if (v < 0) v = 0; }
  return v;
}

__global__ __launch_bounds__(BNPK_BLOCK) void scan_reduce_kernel(const int64_t* __restrict__ in, int64_t n,
                                                                 int window, int64_t* __restrict__ partials) {
  __shared__ int64_t smem[BNPK_BLOCK / 64];
  int64_t base = (int64_t)blockIdx.x * TILE + (int64_t)threadIdx.x * ITEMS;
  int64_t s = 0;
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    int64_t i = base + j;
    if (i < n) s += xform(in[i], window);
  }
  s = wave_reduce_sum(s);
  if (lane_id() == 0) smem[wave_id()] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t t = 0;
    for (int w = 0; w < BNPK_BLOCK / 64; ++w) t += smem[w];
    partials[blockIdx.x] = t;
  }
}

// out[i] = tile_base + exclusive prefix within the tile; if write_total, out[n] = grand total
__global__ __launch_bounds__(BNPK_BLOCK) void scan_apply_kernel(const int64_t* __restrict__ in, int64_t n,
                                                                int window,
                                                                const int64_t* __restrict__ tile_base,
                                                                int64_t* __restrict__ out, int write_total) {
  __shared__ int64_t smem[BNPK_BLOCK / 64 + 1];
  int64_t base = (int64_t)blockIdx.x * TILE + (int64_t)threadIdx.x * ITEMS;
  int64_t v[ITEMS];
  int64_t s = 0;
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    int64_t i = base + j;
    v[j] = (i < n) ? xform(in[i], window) : 0;
    s += v[j];
  }
  int64_t total;
  int64_t ex = block_exclusive_scan(s, smem, &total);
  int64_t run = ex + (tile_base ? tile_base[blockIdx.x] : 0);
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    int64_t i = base + j;
    if (i < n) {
      out[i] = run;
      run += v[j];
      if (write_total && i == n - 1) out[n] = run;
    }
  }
}

__global__ void fill_kernel(int64_t* p, int64_t n, int64_t value) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = value;
}

}  // namespace

size_t bnpk_scan_scratch_bytes(int64_t n) {
  size_t total = 0;
  int64_t m = n;
  while (m > TILE) {
    m = ceil_div(m, TILE);
    total += (size_t)m * sizeof(int64_t);
  }
  return total + 64;
}

int bnpk_scan_launch(bnpk_ctx* ctx, const int64_t* d_in, int64_t n, int window, int64_t* d_out,
                     bool write_total, int64_t* d_scratch, hipStream_t stream) {
  if (n <= 0) {
    if (write_total) {
      hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(64), 0, stream, d_out, (int64_t)1, (int64_t)0);
    }
    return BNPK_OK;
  }
  int64_t nb = ceil_div(n, TILE);
  if (nb > BNPK_MAX_BLOCKS) return BNPK_ERR_RANGE;
  int64_t* partials = nullptr;
  if (nb > 1) {
    partials = d_scratch;
    hipLaunchKernelGGL(scan_reduce_kernel, dim3((unsigned)nb), dim3(BNPK_BLOCK), 0, stream, d_in, n, window,
                       partials);
    // exclusive scan of the partials in place (recursion uses the scratch after them)
    BNPK_CHECK(bnpk_scan_launch(ctx, partials, nb, 1, partials, false, d_scratch + nb, stream));
  }
  hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nb), dim3(BNPK_BLOCK), 0, stream, d_in, n, window,
                     (const int64_t*)partials, d_out, write_total ? 1 : 0);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

extern "C" {

int bnpk_exclusive_scan_i64(bnpk_ctx* ctx, const int64_t* d_in, int64_t n, int64_t* d_out, void* stream) {
  if (!ctx || n < 0 || !d_out || (n > 0 && !d_in)) return BNPK_ERR_ARG;
  void* scratch = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, bnpk_scan_scratch_bytes(n), &scratch, (hipStream_t)stream));
  bnpk_timer t(ctx, "exclusive_scan_i64", (hipStream_t)stream);
  return bnpk_scan_launch(ctx, d_in, n, 1, d_out, true, (int64_t*)scratch, (hipStream_t)stream);
}

int bnpk_row_offsets(bnpk_ctx* ctx, const int64_t* d_lens, int64_t n, int window, int64_t* d_offsets,
                     void* stream) {
  if (!ctx || n < 0 || !d_offsets || (n > 0 && !d_lens) || window < 1) return BNPK_ERR_ARG;
  void* scratch = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, bnpk_scan_scratch_bytes(n), &scratch, (hipStream_t)stream));
  bnpk_timer t(ctx, "row_offsets", (hipStream_t)stream);
  return bnpk_scan_launch(ctx, d_lens, n, window, d_offsets, true, (int64_t*)scratch, (hipStream_t)stream);
}

int bnpk_fill_i64(bnpk_ctx* ctx, int64_t* d_ptr, int64_t n, int64_t value, void* stream) {
  if (!ctx || n < 0 || (n > 0 && !d_ptr)) return BNPK_ERR_ARG;
  if (n == 0) return BNPK_OK;
  int64_t blocks = ceil_div(n, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_ptr, n, value);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

}  // extern "C"
