// Per-row reductions over ragged uint8 data (quality scores): np.sum / np.mean / np.min / np.max(ragged, axis=-1)
// as used by the read filters of the reference (scripts/small_example.py:36-46: np.mean(chunk.quality, axis=1) > 30,
// np.min(chunk.quality, axis=1) > 10).  Rows are short (a read), so eight lanes share a row: each takes aligned
// dwords of the row's byte range, masks the bytes outside it, and the partial sums / minima / maxima meet in three
// shuffle steps.
#include "common.h"

namespace {

constexpr int RR_GROUP = 8;                          // lanes per row
constexpr int RR_ROWS_PER_BLOCK = BNPK_BLOCK / RR_GROUP;

__global__ __launch_bounds__(BNPK_BLOCK) void row_reduce_u8_kernel(const uint8_t* __restrict__ data,
                                                                   const int64_t* __restrict__ off, int64_t n_rows,
                                                                   int64_t* __restrict__ sums, uint8_t* __restrict__ mins,
                                                                   uint8_t* __restrict__ maxs) {
  const int g = threadIdx.x & (RR_GROUP - 1);
  int64_t row = (int64_t)blockIdx.x * RR_ROWS_PER_BLOCK + (threadIdx.x / RR_GROUP);
  const int64_t stride = (int64_t)gridDim.x * RR_ROWS_PER_BLOCK;
  const uint32_t* words = reinterpret_cast<const uint32_t*>(data);      // (the buffer is 16-byte aligned)
  for (; row < n_rows; row += stride) {                                  // (the eight lanes of a group stay together)
    const int64_t s = off[row], e = off[row + 1];
    unsigned long long sum = 0;
    unsigned mn = 255u, mx = 0u;
    for (int64_t d = (s >> 2) + g; d * 4 < e; d += RR_GROUP) {
      const uint32_t x = words[d];
      const int64_t b0 = d * 4;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const unsigned v = (x >> (8 * b)) & 0xffu;
        const bool in = b0 + b >= s && b0 + b < e;
        sum += in ? v : 0u;
        mn = in ? min(mn, v) : mn;
        mx = in ? max(mx, v) : mx;
      }
    }
#pragma unroll
    for (int m = 1; m < RR_GROUP; m <<= 1) {
      sum += __shfl_xor(sum, m, 64);
      mn = min(mn, (unsigned)__shfl_xor((int)mn, m, 64));
      mx = max(mx, (unsigned)__shfl_xor((int)mx, m, 64));
    }
    if (g == 0) {
      if (sums) sums[row] = (int64_t)sum;
      if (mins) mins[row] = (uint8_t)mn;
      if (maxs) maxs[row] = (uint8_t)mx;
    }
  }
}

}  // namespace

extern "C" {

int bnpk_row_reduce_u8(bnpk_ctx* ctx, const uint8_t* d_data, const int64_t* d_offsets, int64_t n_rows, int64_t* d_sums,
                       uint8_t* d_mins, uint8_t* d_maxs, void* stream) {
  if (!ctx || n_rows < 0 || (n_rows > 0 && !d_offsets)) return BNPK_ERR_ARG;
  if (n_rows == 0 || (!d_sums && !d_mins && !d_maxs)) return BNPK_OK;
  if (((uintptr_t)d_data & 3) != 0) return BNPK_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  bnpk_timer t(ctx, "row_reduce_u8", s);
  hipLaunchKernelGGL(row_reduce_u8_kernel, dim3(grid_for(ceil_div(n_rows, RR_ROWS_PER_BLOCK))), dim3(BNPK_BLOCK), 0, s,
                     d_data, d_offsets, n_rows, d_sums, d_mins, d_maxs);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

}  // extern "C"
