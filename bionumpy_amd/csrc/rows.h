// Tile -> first-row table for the output-flat ragged kernels.
// table[t] = the row that contains flat position t*tile (last r with offsets[r] <= t*tile).  Built by one
// lane per row (each tile start lies in exactly one non-empty row), so the consuming kernels read two
// table entries instead of running a 26-step binary search over all row offsets per workgroup.
#pragma once
#include "common.h"

#ifdef __HIPCC__
namespace {

__global__ void tile_rows_kernel(const int64_t* __restrict__ off, int64_t n_rows, int64_t tile,
                                 int64_t* __restrict__ table) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; r < n_rows; r += stride) {
    int64_t s = off[r], e = off[r + 1];
    if (e <= s) continue;
    for (int64_t t = (s + tile - 1) / tile; t * tile < e; ++t) table[t] = r;
  }
}

// rows [lo, hi] that can contain the positions of tile `t` (hi is an upper bound)
__device__ __forceinline__ void tile_row_range(const int64_t* __restrict__ table, int64_t t, int64_t n_tiles,
                                               int64_t n_rows, int64_t& lo, int64_t& hi) {
  lo = table[t];
  hi = (t + 1 < n_tiles) ? table[t + 1] : n_rows - 1;
}

}  // namespace

// scratch bytes for `n_tiles` entries, and the launch (offsets has n_rows+1 entries, total = offsets[n_rows])
static inline size_t tile_rows_bytes(int64_t n_tiles) { return (size_t)(n_tiles + 1) * sizeof(int64_t); }

static inline int build_tile_rows(bnpk_ctx* ctx, const int64_t* d_off, int64_t n_rows, int64_t tile,
                                  int64_t* d_table, hipStream_t s) {
  hipLaunchKernelGGL(tile_rows_kernel, dim3(grid_for(ceil_div(n_rows, 256))), dim3(256), 0, s, d_off, n_rows, tile,
                     d_table);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}
#endif
