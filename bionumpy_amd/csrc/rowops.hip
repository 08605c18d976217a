// Per-row reductions over ragged uint8 data (quality scores): np.sum / np.mean / np.min / np.max(ragged, axis=-1)
// as used by the read filters of the reference (scripts/small_example.py:36-46: np.mean(chunk.quality, axis=1) > 30,
// np.min(chunk.quality, axis=1) > 10).  Rows are short (a read), so a few lanes share a row: each takes aligned
// pieces of the row's byte range, masks the bytes outside it, and the partial sums / minima / maxima meet in two or three
// shuffle steps.
#include "common.h"
#include "rows.h"

namespace {

constexpr int RR_GROUP = 8;                          // lanes per row (8-byte elements)
constexpr int RR8_GROUP = 4;                         // lanes per row of bytes, sixteen bytes each and step
constexpr int RR_ROWS_PER_BLOCK = BNPK_BLOCK / RR_GROUP, RR8_ROWS_PER_BLOCK = BNPK_BLOCK / RR8_GROUP;

// Sixteen bytes per lane and step, four lanes per row.  The sum of four bytes is ONE V_DOT4_U32_U8 against 1,1,1,1; minima and maxima are kept as
// two pairs of 16-bit fields (the even and the odd bytes of a dword: V_PK_MIN_U16 / V_PK_MAX_U16 take two bytes each);
// only the first and the last piece of a row mask the bytes outside it.  (Byte by byte — extract, two 64-bit range
// compares, select, add, min, max — this kernel spent ~190 vector instructions per sixteen bytes, twice what a pass over
// text may cost before it, not HBM, sets the pace: 3.3 ms for the 7.5 GB of quality values of 50 M reads.)
typedef unsigned short rr_u16x2 __attribute__((ext_vector_type(2)));

// starts: the rows need not lie back to back — row r is data[starts[r] .. + off[r + 1] - off[r]) (a field of a text chunk
// that nobody gathered); subtract: a constant taken off every byte first, with the wrap-around of uint8 arithmetic (the
// offset of a DigitEncoding: bionumpy/encodings/__init__.py:15-16).
__device__ __forceinline__ uint64_t rr_subtract(uint64_t x, uint64_t sub) {   // per-byte wrap-around subtraction, no borrows
  constexpr uint64_t H = 0x8080808080808080ull;
  return ((x | H) - (sub & ~H)) ^ ((x ^ ~sub) & H);
}

__global__ __launch_bounds__(BNPK_BLOCK) void row_reduce_u8_kernel(const uint8_t* __restrict__ data, int64_t data_size,
                                                                   const int64_t* __restrict__ starts,
                                                                   const int64_t* __restrict__ off, int64_t n_rows, int subtract,
                                                                   int64_t* __restrict__ sums, uint8_t* __restrict__ mins,
                                                                   uint8_t* __restrict__ maxs) {
  const int g = threadIdx.x & (RR8_GROUP - 1);
  int64_t row = (int64_t)blockIdx.x * RR8_ROWS_PER_BLOCK + (threadIdx.x / RR8_GROUP);
  const int64_t stride = (int64_t)gridDim.x * RR8_ROWS_PER_BLOCK;
  // (a buffer that does not start on a 16-byte boundary — a view into a larger one — is indexed from the boundary in front of it)
  const int64_t skew = (int64_t)(reinterpret_cast<uintptr_t>(data) & 15);
  data -= skew;
  const int64_t total = (data_size >= 0 ? data_size : off[n_rows]) + skew;
  const bool extremes = mins != nullptr || maxs != nullptr;              // (uniform)
  const uint64_t sub = (uint64_t)(subtract & 0xff) * 0x0101010101010101ull;
  for (; row < n_rows; row += stride) {                                  // (the lanes of a group stay together)
    const int64_t s = (starts ? starts[row] : off[row]) + skew, e = s + (off[row + 1] - off[row]);
    unsigned long long sum = 0;
    unsigned mn0 = 0x00ff00ffu, mn1 = 0x00ff00ffu, mx0 = 0u, mx1 = 0u;   // even / odd bytes as 16-bit fields
    for (int64_t b0 = ((s >> 4) + g) << 4; b0 < e; b0 += 16 * RR8_GROUP) {
      uint64_t v[2];
      if (b0 + 16 <= total) {
        const uint4 x = *reinterpret_cast<const uint4*>(data + b0);
        v[0] = (uint64_t)x.x | ((uint64_t)x.y << 32);
        v[1] = (uint64_t)x.z | ((uint64_t)x.w << 32);
      } else {                                                           // the last, partial piece of the buffer
        v[0] = v[1] = 0;
        for (int j = 0; b0 + j < total; ++j) v[j >> 3] |= (uint64_t)data[b0 + j] << (8 * (j & 7));
      }
      if (subtract) { v[0] = rr_subtract(v[0], sub); v[1] = rr_subtract(v[1], sub); }      // (uniform)
      const bool inside = b0 >= s && b0 + 16 <= e;                       // all sixteen bytes belong to the row
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        uint64_t in = ~0ull;                                             // 0xff in every byte that belongs to the row
        if (!inside) {
          const int64_t w0 = b0 + 8 * h;
          if (w0 + 8 <= s || w0 >= e) in = 0;
          else {
            if (w0 < s) in &= ~0ull << (8 * (int)(s - w0));
            if (w0 + 8 > e) in &= ~0ull >> (8 * (int)(w0 + 8 - e));
          }
        }
        const uint64_t vs = v[h] & in;
        sum += __builtin_amdgcn_udot4((uint32_t)vs, 0x01010101u, __builtin_amdgcn_udot4((uint32_t)(vs >> 32), 0x01010101u, 0u, false), false);
        if (extremes) {
          const uint64_t lo = v[h] | ~in;                                // bytes outside the row: 0xff for the minimum, 0 (vs) for the maximum
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const uint32_t a = (uint32_t)(lo >> (32 * q)), b = (uint32_t)(vs >> (32 * q));
            mn0 = __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(rr_u16x2, mn0), __builtin_bit_cast(rr_u16x2, a & 0x00ff00ffu)));
            mn1 = __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(rr_u16x2, mn1), __builtin_bit_cast(rr_u16x2, (a >> 8) & 0x00ff00ffu)));
            mx0 = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(rr_u16x2, mx0), __builtin_bit_cast(rr_u16x2, b & 0x00ff00ffu)));
            mx1 = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(rr_u16x2, mx1), __builtin_bit_cast(rr_u16x2, (b >> 8) & 0x00ff00ffu)));
          }
        }
      }
    }
    unsigned mn = min(min(mn0 & 0xffffu, mn0 >> 16), min(mn1 & 0xffffu, mn1 >> 16));
    unsigned mx = max(max(mx0 & 0xffffu, mx0 >> 16), max(mx1 & 0xffffu, mx1 >> 16));
#pragma unroll
    for (int m = 1; m < RR8_GROUP; m <<= 1) {
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

// The same for 8-byte elements (k-mer hashes: Minimizers.__call__ takes kmer_hashes.raw().min(axis=-1),
// bionumpy/sequence/minimizers.py:15-17; motif scores: float64).  One group of eight lanes per row.
template <typename T>
__global__ __launch_bounds__(BNPK_BLOCK) void row_reduce_wide_kernel(const T* __restrict__ data, const int64_t* __restrict__ off,
                                                                     int64_t n_rows, T* __restrict__ sums, T* __restrict__ mins,
                                                                     T* __restrict__ maxs) {
  const int g = threadIdx.x & (RR_GROUP - 1);
  int64_t row = (int64_t)blockIdx.x * RR_ROWS_PER_BLOCK + (threadIdx.x / RR_GROUP);
  const int64_t stride = (int64_t)gridDim.x * RR_ROWS_PER_BLOCK;
  for (; row < n_rows; row += stride) {
    const int64_t s = off[row], e = off[row + 1];
    T sum = T(0), mn = T(0), mx = T(0);
    bool any = false;
    for (int64_t i = s + g; i < e; i += RR_GROUP) {
      const T v = data[i];
      sum += v;
      // (numpy's min / max propagate NaN: v != v)
      mn = !any ? v : ((v < mn || v != v) ? v : mn);
      mx = !any ? v : ((v > mx || v != v) ? v : mx);
      any = true;
    }
#pragma unroll
    for (int m = 1; m < RR_GROUP; m <<= 1) {
      const T os = __shfl_xor(sum, m, 64), on = __shfl_xor(mn, m, 64), ox = __shfl_xor(mx, m, 64);
      const bool oa = __shfl_xor((int)any, m, 64) != 0;
      sum += os;
      if (oa) {
        const bool keep_n = any && mn != mn, keep_x = any && mx != mx;           // this side already holds a NaN
        mn = !any ? on : (keep_n ? mn : ((on < mn || on != on) ? on : mn));
        mx = !any ? ox : (keep_x ? mx : ((ox > mx || ox != ox) ? ox : mx));
        any = true;
      }
    }
    if (g == 0) {
      if (sums) sums[row] = sum;
      if (mins) mins[row] = mn;
      if (maxs) maxs[row] = mx;
    }
  }
}

}  // namespace

extern "C" {

int bnpk_row_reduce_wide(bnpk_ctx* ctx, const void* d_data, int is_f64, const int64_t* d_offsets, int64_t n_rows, void* d_sums,
                         void* d_mins, void* d_maxs, void* stream) {
  if (!ctx || n_rows < 0 || (n_rows > 0 && !d_offsets)) return BNPK_ERR_ARG;
  if (n_rows == 0 || (!d_sums && !d_mins && !d_maxs)) return BNPK_OK;
  hipStream_t s = (hipStream_t)stream;
  bnpk_timer t(ctx, "row_reduce_wide", s);
  const dim3 grid(grid_for(ceil_div(n_rows, RR_ROWS_PER_BLOCK)));
  if (is_f64)
    hipLaunchKernelGGL(row_reduce_wide_kernel<double>, grid, dim3(BNPK_BLOCK), 0, s, (const double*)d_data, d_offsets, n_rows,
                       (double*)d_sums, (double*)d_mins, (double*)d_maxs);
  else
    hipLaunchKernelGGL(row_reduce_wide_kernel<int64_t>, grid, dim3(BNPK_BLOCK), 0, s, (const int64_t*)d_data, d_offsets, n_rows,
                       (int64_t*)d_sums, (int64_t*)d_mins, (int64_t*)d_maxs);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_row_reduce_u8_view(bnpk_ctx* ctx, const uint8_t* d_data, int64_t data_size, const int64_t* d_starts,
                            const int64_t* d_offsets, int64_t n_rows, int subtract, int64_t* d_sums, uint8_t* d_mins,
                            uint8_t* d_maxs, void* stream) {
  if (!ctx || n_rows < 0 || (n_rows > 0 && !d_offsets) || subtract < 0 || subtract > 255) return BNPK_ERR_ARG;
  if (d_starts && data_size < 0) return BNPK_ERR_ARG;
  if (n_rows == 0 || (!d_sums && !d_mins && !d_maxs)) return BNPK_OK;
  hipStream_t s = (hipStream_t)stream;
  bnpk_timer t(ctx, "row_reduce_u8", s);
  hipLaunchKernelGGL(row_reduce_u8_kernel, dim3(grid_for(ceil_div(n_rows, RR8_ROWS_PER_BLOCK))), dim3(BNPK_BLOCK), 0, s,
                     d_data, data_size, d_starts, d_offsets, n_rows, subtract, d_sums, d_mins, d_maxs);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_row_reduce_u8(bnpk_ctx* ctx, const uint8_t* d_data, const int64_t* d_offsets, int64_t n_rows, int64_t* d_sums,
                       uint8_t* d_mins, uint8_t* d_maxs, void* stream) {
  return bnpk_row_reduce_u8_view(ctx, d_data, -1, nullptr, d_offsets, n_rows, 0, d_sums, d_mins, d_maxs, stream);
}

}  // extern "C"

// ---- element-wise helpers of the read filters ------------------------------------------------------------------------
// np.mean(chunk.quality, axis=1) > 30, mask[::3] = False, mask1 & mask2 (scripts/small_example.py:36-46): the per-row
// values of a 50 M-read chunk are 400 MB — they stay in HBM behind bionumpy_amd/device_vector.py and these kernels
// compute on them; only what the caller looks at crosses PCIe.
namespace {

__global__ void vec_ratio_rows_kernel(const int64_t* __restrict__ sums, const int64_t* __restrict__ off, int64_t n,
                                      double* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = (double)sums[i] / (double)(off[i + 1] - off[i]);     // 0 / 0 = nan, as numpy
}

template <typename T>
__device__ __forceinline__ bool vec_cmp(T x, T y, int op) {
  return op == 0 ? x < y : op == 1 ? x <= y : op == 2 ? x > y : op == 3 ? x >= y : op == 4 ? x == y : x != y;
}

template <typename T>
__global__ void vec_compare_kernel(const T* __restrict__ x, int64_t n, int op, T y, uint8_t* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = vec_cmp(x[i], y, op) ? 1 : 0;
}

__global__ void mask_logic_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, int64_t n, int op,
                                  uint8_t* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const bool x = a[i] != 0, y = b ? b[i] != 0 : false;
    out[i] = (op == 0 ? (x && y) : op == 1 ? (x || y) : op == 2 ? (x != y) : !x) ? 1 : 0;
  }
}

// whole entries `rows` of a one-line-per-field buffer: starts[i] = byte behind the newline in front of entry rows[i] (0 for
// entry 0), lens[i] = through the newline of its last line
__global__ void entry_table_kernel(const int64_t* __restrict__ nl, int lpe, const int64_t* __restrict__ rows, int64_t m,
                                   int64_t* __restrict__ starts, int64_t* __restrict__ lens) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < m; i += stride) {
    const int64_t first = rows[i] * lpe;
    const int64_t s = first > 0 ? nl[first - 1] + 1 : 0;
    starts[i] = s;
    lens[i] = nl[first + lpe - 1] + 1 - s;
  }
}

struct jl_offsets { const int64_t* off[4]; int extra[4]; };    // per line: row offsets of its field (nullptr: one byte), prefix + newline

// bytes of entry r when its lines are written out: sum over the lines of (field row length | 1) + prefix + 1
__global__ void join_line_lens_kernel(jl_offsets lines, int n_lines, int64_t n_rows, int64_t* __restrict__ lens) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; r < n_rows; r += stride) {
    int64_t total = 0;
    for (int i = 0; i < n_lines; ++i) total += (lines.off[i] ? lines.off[i][r + 1] - lines.off[i][r] : 1) + lines.extra[i];
    lens[r] = total;
  }
}

__global__ void take_i64_kernel(const int64_t* __restrict__ arr, const int64_t* __restrict__ idx, int64_t m,
                                int64_t* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < m; i += stride) out[i] = arr[idx[i]];
}

__global__ void mask_fill_kernel(uint8_t* __restrict__ mask, int64_t start, int64_t step, int64_t count, uint8_t value) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < count; i += stride) mask[start + i * step] = value;
}

}  // namespace

extern "C" {

int bnpk_vec_ratio_rows(bnpk_ctx* ctx, const int64_t* d_sums, const int64_t* d_offsets, int64_t n, double* d_out, void* stream) {
  if (!ctx || n < 0 || (n > 0 && (!d_sums || !d_offsets || !d_out))) return BNPK_ERR_ARG;
  if (n == 0) return BNPK_OK;
  hipStream_t s = (hipStream_t)stream;
  bnpk_timer t(ctx, "vec_ratio_rows", s);
  hipLaunchKernelGGL(vec_ratio_rows_kernel, dim3(grid_for(ceil_div(n, 256))), dim3(256), 0, s, d_sums, d_offsets, n, d_out);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_vec_compare(bnpk_ctx* ctx, const void* d_x, int64_t n, int dtype, int op, double scalar_f64, int64_t scalar_i64,
                     uint8_t* d_out, void* stream) {
  if (!ctx || n < 0 || dtype < 0 || dtype > 2 || op < 0 || op > 5 || (n > 0 && (!d_x || !d_out))) return BNPK_ERR_ARG;
  if (n == 0) return BNPK_OK;
  hipStream_t s = (hipStream_t)stream;
  bnpk_timer t(ctx, "vec_compare", s);
  const dim3 grid(grid_for(ceil_div(n, 256))), block(256);
  if (dtype == 0) hipLaunchKernelGGL((vec_compare_kernel<double>), grid, block, 0, s, (const double*)d_x, n, op, scalar_f64, d_out);
  else if (dtype == 1) hipLaunchKernelGGL((vec_compare_kernel<int64_t>), grid, block, 0, s, (const int64_t*)d_x, n, op, scalar_i64, d_out);
  else {
    if (scalar_i64 < 0 || scalar_i64 > 255) return BNPK_ERR_ARG;     // (the caller folds comparisons that no uint8 can decide)
    hipLaunchKernelGGL((vec_compare_kernel<uint8_t>), grid, block, 0, s, (const uint8_t*)d_x, n, op, (uint8_t)scalar_i64, d_out);
  }
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_mask_logic(bnpk_ctx* ctx, const uint8_t* d_a, const uint8_t* d_b, int64_t n, int op, uint8_t* d_out, void* stream) {
  if (!ctx || n < 0 || op < 0 || op > 3 || (n > 0 && (!d_a || !d_out || (op != 3 && !d_b)))) return BNPK_ERR_ARG;
  if (n == 0) return BNPK_OK;
  hipStream_t s = (hipStream_t)stream;
  bnpk_timer t(ctx, "mask_logic", s);
  hipLaunchKernelGGL(mask_logic_kernel, dim3(grid_for(ceil_div(n, 256))), dim3(256), 0, s, d_a, op == 3 ? nullptr : d_b, n, op, d_out);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_entry_table(bnpk_ctx* ctx, const int64_t* d_newlines, int lines_per_entry, const int64_t* d_rows, int64_t m,
                     int64_t* d_starts, int64_t* d_lens, void* stream) {
  if (!ctx || m < 0 || lines_per_entry < 1 || (m > 0 && (!d_newlines || !d_rows || !d_starts || !d_lens))) return BNPK_ERR_ARG;
  if (m == 0) return BNPK_OK;
  hipStream_t s = (hipStream_t)stream;
  bnpk_timer t(ctx, "entry_table", s);
  hipLaunchKernelGGL(entry_table_kernel, dim3(grid_for(ceil_div(m, 256))), dim3(256), 0, s, d_newlines, lines_per_entry, d_rows, m,
                     d_starts, d_lens);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_join_line_lens(bnpk_ctx* ctx, int64_t n_rows, int n_lines, const int64_t* const* d_field_offsets, const int* prefix,
                        int64_t* d_lens, void* stream) {
  if (!ctx || n_rows < 0 || n_lines < 1 || n_lines > 4 || !d_field_offsets || !prefix || (n_rows > 0 && !d_lens)) return BNPK_ERR_ARG;
  if (n_rows == 0) return BNPK_OK;
  jl_offsets lines;
  for (int i = 0; i < 4; ++i) {
    lines.off[i] = i < n_lines ? d_field_offsets[i] : nullptr;
    lines.extra[i] = i < n_lines ? prefix[i] + 1 : 0;
  }
  hipStream_t s = (hipStream_t)stream;
  bnpk_timer t(ctx, "join_line_lens", s);
  hipLaunchKernelGGL(join_line_lens_kernel, dim3(grid_for(ceil_div(n_rows, 256))), dim3(256), 0, s, lines, n_lines, n_rows, d_lens);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_take_i64(bnpk_ctx* ctx, const int64_t* d_arr, const int64_t* d_idx, int64_t m, int64_t* d_out, void* stream) {
  if (!ctx || m < 0 || (m > 0 && (!d_arr || !d_idx || !d_out))) return BNPK_ERR_ARG;
  if (m == 0) return BNPK_OK;
  hipStream_t s = (hipStream_t)stream;
  bnpk_timer t(ctx, "take_i64", s);
  hipLaunchKernelGGL(take_i64_kernel, dim3(grid_for(ceil_div(m, 256))), dim3(256), 0, s, d_arr, d_idx, m, d_out);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_mask_fill(bnpk_ctx* ctx, uint8_t* d_mask, int64_t n, int64_t start, int64_t step, int64_t count, int value, void* stream) {
  if (!ctx || n < 0 || start < 0 || step < 1 || count < 0 || (count > 0 && (!d_mask || start + (count - 1) * step >= n)))
    return BNPK_ERR_ARG;
  if (count == 0) return BNPK_OK;
  hipStream_t s = (hipStream_t)stream;
  bnpk_timer t(ctx, "mask_fill", s);
  hipLaunchKernelGGL(mask_fill_kernel, dim3(grid_for(ceil_div(count, 256))), dim3(256), 0, s, d_mask, start, step, count,
                     (uint8_t)(value ? 1 : 0));
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

}  // extern "C"

// ---- join_fields: text of records from their fields (bionumpy/io/one_line_buffer.py:119-134) ----------------------
// Entry r = its lines one after the other; line i = `prefix` header bytes, the field's row r (plus `add`: quality
// scores are written as score + 33), a newline.  A line without a field is the constant byte `fill` ('+').
namespace {

constexpr int JL_MAX_LINES = 4;
constexpr int64_t JL_TILE = 16384;                      // output bytes a workgroup puts together in LDS

struct jl_line {
  const uint8_t* data;         // flat bytes of the field (nullptr: a one-byte constant line)
  const int64_t* off;          // its row offsets
  const int64_t* starts;       // nullptr: row r lies at off[r]; else at starts[r] (a view of a larger buffer, never gathered)
  int64_t size;                // bytes of `data` (what a 16-byte read must stay inside)
  int add;                     // added to every byte of the field
  int prefix;                  // header bytes in front of the field (0 or 1)
  uint8_t fill;                // the constant byte of a line without a field
};
struct jl_lines { jl_line l[JL_MAX_LINES]; };

__device__ __forceinline__ int64_t jl_row_of(const int64_t* __restrict__ off, int64_t lo, int64_t hi, int64_t p) {
  while (lo < hi) {
    const int64_t mid = lo + ((hi - lo + 1) >> 1);
    if (off[mid] <= p) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// The text of a tile of the output is put together in LDS and leaves as aligned 16-byte stores.  The unit of work is a
// 16-byte WORD OF A FIELD: a lane loads it where it lies (one unaligned 16-byte load) and stores it where it belongs in
// the tile (one unaligned 16-byte LDS store — gfx950 takes them, at a quarter of the aligned rate, which is still 16 bytes
// per clock and CU: scripts/exp/ubench/lds_unaligned.hip); the last word of a field goes in pieces of 8, 4, 2, 1 bytes.
// Which field a word belongs to is found without a search: one thread per entry writes a descriptor per line (source,
// place in the tile, bytes left, number of words that overlap the tile) and the constant bytes (header byte, fill byte,
// newlines); a scan over the word counts numbers the words; every descriptor marks its first word in an array over the
// words, and a running maximum carries the mark to the words behind it.  Then all 256 lanes copy words, independently.
// Rounds 1-2 worked the other way round — every lane owned sixteen OUTPUT bytes, found its entry by a binary search and
// walked the lines to see where its bytes came from: ~1200 lane cycles per sixteen bytes, 31 ms for 50 M FASTQ records,
// whatever was done to the loads; half a wavefront per entry (the first version of this kernel) was no faster: the
// entries of a group went through their loads one after the other.
constexpr int JL_RECS = 64;                             // entries described per pass (a FASTQ tile holds ~52)
constexpr int JL_DESC = JL_RECS * JL_MAX_LINES;
constexpr int JL_WORDS = 1536;                          // words numbered per pass (a FASTQ tile: ~1150)
constexpr int JL_PER_LANE = JL_WORDS / BNPK_BLOCK;

__global__ __launch_bounds__(BNPK_BLOCK) void join_lines_kernel(jl_lines lines, int n_lines, uint8_t header,
                                                                const int64_t* __restrict__ entry_off, int64_t n_rows,
                                                                int64_t total, const int64_t* __restrict__ tile_rows,
                                                                uint8_t* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[JL_TILE];
  __shared__ int64_t d_src[JL_DESC];                           // first byte of the field's first word inside the tile
  __shared__ int d_q[JL_DESC];                                 // where that word goes (may be up to 15 bytes before the tile)
  __shared__ int d_left[JL_DESC];                              // bytes of the field from that word on (capped)
  __shared__ int d_first[JL_DESC + 1];                         // words before the descriptor's first one
  __shared__ unsigned short owner[JL_WORDS];
  __shared__ int s_scan[BNPK_BLOCK / 64 + 1];
  const int64_t blk0 = (int64_t)blockIdx.x * JL_TILE;
  if (blk0 >= total) return;
  const int tile_n = (int)min(JL_TILE, total - blk0);
  const int64_t lo = tile_rows[blockIdx.x];
  const int64_t hi = (blk0 + JL_TILE < total) ? tile_rows[blockIdx.x + 1] : n_rows - 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int64_t field_end[JL_MAX_LINES];                             // sixteen bytes can be read at `at` iff at + 16 <= field_end
#pragma unroll
  for (int l = 0; l < JL_MAX_LINES; ++l)
    field_end[l] = (l < n_lines && lines.l[l].data) ? (lines.l[l].starts ? lines.l[l].size : lines.l[l].off[n_rows]) : 0;
  const unsigned tile_base = (unsigned)(size_t)tile;           // (LDS addresses are 32-bit)
  auto put = [&](int64_t q, uint8_t v) { if (q >= 0 && q < tile_n) tile[q] = v; };
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

  for (int64_t r0 = lo; r0 <= hi; r0 += JL_RECS) {
    const int n_rec = (int)min((int64_t)JL_RECS, hi - r0 + 1);
    const int n_desc = n_rec * n_lines;
    // ---- one thread per entry: its constant bytes, a descriptor per line
    int words_mine = 0;                                        // (of the descriptors this thread writes: summed below)
    if (tid < n_rec) {
      const int64_t r = r0 + tid;
      int64_t o = entry_off[r] - blk0;                         // where the entry starts, relative to the tile (may lie before it)
      for (int i = 0; i < n_lines; ++i) {
        const jl_line& L = lines.l[i];
        const int d = tid * n_lines + i;
        if (L.prefix) { put(o, header); ++o; }
        int64_t flen = 1;
        int nw = 0;
        if (!L.data) {
          put(o, L.fill);
        } else {
          const int64_t fs = L.starts ? L.starts[r] : L.off[r];
          flen = L.off[r + 1] - L.off[r];
          // the 16-byte words of the field that overlap the tile
          const int64_t w_lo = o < 0 ? (-o) >> 4 : 0;
          const int64_t w_hi = min((flen + 15) >> 4, ((int64_t)tile_n - o + 15) >> 4);
          if (w_hi > w_lo) {
            nw = (int)(w_hi - w_lo);
            d_src[d] = fs + 16 * w_lo;
            d_q[d] = (int)(o + 16 * w_lo);
            d_left[d] = (int)min(flen - 16 * w_lo, (int64_t)1 << 30);
          }
        }
        d_first[d] = nw;
        words_mine += nw;
        o += flen;
        put(o, 10);
        ++o;
      }
    }
    __syncthreads();
    // ---- number the words: exclusive scan of the counts (a thread scans the lines of its entry)
    int before;
    {
      const int inc = (int)wave_inclusive_scan((unsigned)words_mine);
      if (lane == 63) s_scan[wave] = inc;
      __syncthreads();
      int base = 0, all = 0;
      for (int w = 0; w < BNPK_BLOCK / 64; ++w) { if (w < wave) base += s_scan[w]; all += s_scan[w]; }
      before = base + inc - words_mine;
      if (tid == 0) s_scan[BNPK_BLOCK / 64] = all;
    }
    if (tid < n_rec) {
      for (int i = 0; i < n_lines; ++i) {
        const int d = tid * n_lines + i, nw = d_first[d];
        d_first[d] = before;
        before += nw;
      }
    }
    __syncthreads();
    const int n_words = s_scan[BNPK_BLOCK / 64];
    if (tid == 0) d_first[n_desc] = n_words;
    // ---- the words, JL_WORDS at a time
    for (int base = 0; base < n_words; base += JL_WORDS) {
      for (int t = tid; t < JL_WORDS; t += BNPK_BLOCK) owner[t] = 0;
      __syncthreads();
      for (int d = tid; d < n_desc; d += BNPK_BLOCK) {           // a descriptor marks its first word of this pass
        const int f = d_first[d], e = d_first[d + 1];
        if (e > f && e > base && f < base + JL_WORDS) owner[max(f, base) - base] = (unsigned short)(d + 1);
      }
      __syncthreads();
      // running maximum over the marks: a lane takes JL_PER_LANE consecutive words
      unsigned mark[JL_PER_LANE], top = 0;
#pragma unroll
      for (int j = 0; j < JL_PER_LANE; ++j) { top = max(top, (unsigned)owner[tid * JL_PER_LANE + j]); mark[j] = top; }
      unsigned run = top;                                      // inclusive maximum over the lanes of the wave ..
#pragma unroll
      for (int sh = 1; sh < 64; sh <<= 1) {
        const unsigned other = (unsigned)__shfl_up((int)run, sh, 64);
        if (lane >= sh) run = max(run, other);
      }
      if (lane == 63) s_scan[wave] = (int)run;
      __syncthreads();
      unsigned carry = (unsigned)__shfl_up((int)run, 1, 64);
      if (lane == 0) carry = 0;
      for (int w = 0; w < wave; ++w) carry = max(carry, (unsigned)s_scan[w]);   // .. and over the waves before
      __syncthreads();
#pragma unroll
      for (int j = 0; j < JL_PER_LANE; ++j) owner[tid * JL_PER_LANE + j] = (unsigned short)max(mark[j], carry);
      __syncthreads();
      // every lane copies words: t = tid, tid + 256, ... (consecutive lanes, consecutive words of a field: coalesced)
      const int n_here = min(JL_WORDS, n_words - base);
      for (int t0 = 0; t0 < n_here; t0 += BNPK_BLOCK * 2) {
        int dsc[2], m[2];
        int64_t q[2];
        uint64_t x[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {                          // two independent words per lane in flight
          const int t = t0 + u * BNPK_BLOCK + tid;
          dsc[u] = -1;
          if (t < n_here) {
            const int d = (int)owner[t] - 1;
            const int w = base + t - d_first[d];
            const int64_t at = d_src[d] + 16 * (int64_t)w;
            dsc[u] = d % n_lines;
            m[u] = min(16, d_left[d] - 16 * w);
            q[u] = (int64_t)d_q[d] + 16 * (int64_t)w;
            const jl_line& L = lines.l[0];
            (void)L;
            const uint8_t* src = nullptr;
            int64_t end = 0;
#pragma unroll
            for (int l = 0; l < JL_MAX_LINES; ++l)
              if (l == dsc[u]) { src = lines.l[l].data; end = field_end[l]; }
            x[u][0] = x[u][1] = 0;
            if (at + 16 <= end) {
              __builtin_memcpy(x[u], src + at, 16);
            } else {
              uint64_t x0 = 0, x1 = 0;                         // (no dynamic index: the words stay in registers)
              for (int b = 0; b < m[u]; ++b) {
                const uint64_t v = (uint64_t)src[at + b] << (8 * (b & 7));
                if (b < 8) x0 |= v; else x1 |= v;
              }
              x[u][0] = x0;
              x[u][1] = x1;
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (dsc[u] < 0) continue;
          int add = 0;
#pragma unroll
          for (int l = 0; l < JL_MAX_LINES; ++l)
            if (l == dsc[u]) add = lines.l[l].add;
          if (add) {                                           // per-byte wrap-around addition without carries between bytes
            const uint64_t a = (uint64_t)(add & 0xff) * 0x0101010101010101ull;
#pragma unroll
            for (int h = 0; h < 2; ++h)
              x[u][h] = ((x[u][h] & 0x7f7f7f7f7f7f7f7full) + (a & 0x7f7f7f7f7f7f7f7full)) ^ ((x[u][h] ^ a) & 0x8080808080808080ull);
          }
          if (q[u] >= 0 && q[u] + 16 <= tile_n) {
            const unsigned addr = tile_base + (unsigned)q[u];
            if (m[u] == 16) {
              const u32x4 v = {(unsigned)x[u][0], (unsigned)(x[u][0] >> 32), (unsigned)x[u][1], (unsigned)(x[u][1] >> 32)};
              asm volatile("ds_write_b128 %0, %1" : : "v"(addr), "v"(v) : "memory");
            } else {                                           // in pieces of 8, 4, 2, 1 bytes (unaligned, like the whole)
              uint64_t rest = x[u][0];
              unsigned at = addr;
              if (m[u] & 8) { asm volatile("ds_write_b64 %0, %1" : : "v"(at), "v"(rest) : "memory"); rest = x[u][1]; at += 8; }
              if (m[u] & 4) { asm volatile("ds_write_b32 %0, %1" : : "v"(at), "v"((unsigned)rest) : "memory"); rest >>= 32; at += 4; }
              if (m[u] & 2) { asm volatile("ds_write_b16 %0, %1" : : "v"(at), "v"((unsigned)rest) : "memory"); rest >>= 16; at += 2; }
              if (m[u] & 1) { asm volatile("ds_write_b8 %0, %1" : : "v"(at), "v"((unsigned)rest) : "memory"); }
            }
          } else {                                             // what sticks out of the tile: byte by byte
            const uint64_t x0 = x[u][0], x1 = x[u][1];
            for (int b = 0; b < m[u]; ++b) put(q[u] + b, (uint8_t)((b < 8 ? x0 : x1) >> (8 * (b & 7))));
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (the asm stores are the compiler's blind spot)
      __syncthreads();
    }
    __syncthreads();
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = tid * 16; t < tile_n; t += BNPK_BLOCK * 16) {
    if (t + 16 <= tile_n) {
      const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(tile + t);
      typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
      u64x2 pair;
      pair.x = v.x;
      pair.y = v.y;
      __builtin_nontemporal_store(pair, reinterpret_cast<u64x2*>(out + blk0 + t));
    } else {
      for (int b = t; b < tile_n; ++b) out[blk0 + b] = tile[b];
    }
  }
}

}  // namespace

extern "C" int bnpk_join_lines(bnpk_ctx* ctx, int64_t n_rows, int n_lines, const uint8_t* const* d_field_data,
                               const int64_t* const* d_field_offsets, const int64_t* const* d_field_starts,
                               const int64_t* field_sizes, const int* add, const int* prefix,
                               const uint8_t* fill, uint8_t header, const int64_t* d_entry_offsets, int64_t total,
                               uint8_t* d_out, void* stream) {
  if (!ctx || n_rows < 0 || n_lines < 1 || n_lines > JL_MAX_LINES || total < 0 || !d_field_data || !d_field_offsets || !add ||
      !prefix || !fill)
    return BNPK_ERR_ARG;
  if (n_rows == 0 || total == 0) return BNPK_OK;
  if (!d_entry_offsets || !d_out) return BNPK_ERR_ARG;
  jl_lines lines;
  memset(&lines, 0, sizeof(lines));
  for (int i = 0; i < n_lines; ++i) {
    if (d_field_data[i] && !d_field_offsets[i]) return BNPK_ERR_ARG;
    if (prefix[i] < 0 || prefix[i] > 1) return BNPK_ERR_ARG;
    const int64_t* starts = d_field_starts ? d_field_starts[i] : nullptr;
    if (starts && (!field_sizes || field_sizes[i] < 0)) return BNPK_ERR_ARG;
    lines.l[i] = jl_line{d_field_data[i], d_field_offsets[i], starts, starts ? field_sizes[i] : 0, add[i], prefix[i], fill[i]};
  }
  hipStream_t s = (hipStream_t)stream;
  const int64_t n_tiles = ceil_div(total, JL_TILE);
  if (n_tiles > BNPK_MAX_BLOCKS) return BNPK_ERR_RANGE;
  void* table = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, tile_rows_bytes(n_tiles), &table, (hipStream_t)stream));
  bnpk_timer t(ctx, "join_lines", s);
  BNPK_CHECK(build_tile_rows(ctx, d_entry_offsets, n_rows, JL_TILE, (int64_t*)table, s));
  hipLaunchKernelGGL(join_lines_kernel, dim3((unsigned)n_tiles), dim3(BNPK_BLOCK), 0, s, lines, n_lines, header,
                     d_entry_offsets, n_rows, total, (const int64_t*)table, d_out);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

// ---- per-column sums of ragged uint8 data: np.sum / np.mean(ragged, axis=0) --------------------------------------
// (scripts/small_example.py:20-22,49-52: mean match / base quality per read position).  Column c of row r is element
// off[r] + c; sums[c] adds those bytes over the rows that are long enough, counts[c] counts them.  A lane owns 8
// consecutive flat bytes, finds its row through the tile->row table and adds into an LDS-private table (columns
// < CS_LDS_COLS; longer rows go to global atomics directly), flushed with 64-bit atomics.
namespace {

constexpr int CS_BYTES_PER_LANE = 8;
constexpr int64_t CS_TILE = (int64_t)BNPK_BLOCK * CS_BYTES_PER_LANE;
constexpr int CS_LDS_COLS = 4096;

__global__ __launch_bounds__(BNPK_BLOCK) void col_sums_u8_kernel(const uint8_t* __restrict__ data,
                                                                 const int64_t* __restrict__ off, int64_t n_rows,
                                                                 int64_t total, const int64_t* __restrict__ tile_rows,
                                                                 int64_t n_cols, unsigned long long* __restrict__ sums,
                                                                 unsigned long long* __restrict__ counts) {
  __shared__ unsigned lsum[CS_LDS_COLS];
  __shared__ unsigned lcnt[CS_LDS_COLS];
  const int lds_cols = (int)min((int64_t)CS_LDS_COLS, n_cols);
  for (int c = threadIdx.x; c < lds_cols; c += BNPK_BLOCK) { lsum[c] = 0; lcnt[c] = 0; }
  __syncthreads();
  const int64_t p0 = ((int64_t)blockIdx.x * BNPK_BLOCK + threadIdx.x) * CS_BYTES_PER_LANE;
  if (p0 < total) {
    const int64_t lo = tile_rows[blockIdx.x];
    const int64_t hi = ((int64_t)(blockIdx.x + 1) * CS_TILE < total) ? tile_rows[blockIdx.x + 1] : n_rows - 1;
    int64_t r = jl_row_of(off, lo, hi, p0);
    int64_t s = off[r], e = off[r + 1];
    const int64_t p1 = min(p0 + CS_BYTES_PER_LANE, total);
    for (int64_t p = p0; p < p1; ++p) {
      while (e <= p) { ++r; s = e; e = off[r + 1]; }
      const int64_t c = p - s;
      const unsigned v = data[p];
      if (c < lds_cols) { atomicAdd(&lsum[c], v); atomicAdd(&lcnt[c], 1u); }
      else { atomicAdd(&sums[c], (unsigned long long)v); atomicAdd(&counts[c], 1ull); }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < lds_cols; c += BNPK_BLOCK) {
    if (lcnt[c]) { atomicAdd(&sums[c], (unsigned long long)lsum[c]); atomicAdd(&counts[c], (unsigned long long)lcnt[c]); }
  }
}

}  // namespace

extern "C" int bnpk_col_sums_u8(bnpk_ctx* ctx, const uint8_t* d_data, const int64_t* d_offsets, int64_t n_rows, int64_t total,
                                int64_t n_cols, int64_t* d_sums, int64_t* d_counts, void* stream) {
  if (!ctx || n_rows < 0 || total < 0 || n_cols < 0 || (n_cols > 0 && (!d_sums || !d_counts))) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (n_cols > 0) {
    BNPK_HIP(ctx, hipMemsetAsync(d_sums, 0, (size_t)n_cols * 8, s));
    BNPK_HIP(ctx, hipMemsetAsync(d_counts, 0, (size_t)n_cols * 8, s));
  }
  if (n_rows == 0 || total == 0 || n_cols == 0) return BNPK_OK;
  if (!d_data || !d_offsets) return BNPK_ERR_ARG;
  const int64_t n_tiles = ceil_div(total, CS_TILE);
  if (n_tiles > BNPK_MAX_BLOCKS) return BNPK_ERR_RANGE;
  void* table = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, tile_rows_bytes(n_tiles), &table, (hipStream_t)stream));
  bnpk_timer t(ctx, "col_sums_u8", s);
  BNPK_CHECK(build_tile_rows(ctx, d_offsets, n_rows, CS_TILE, (int64_t*)table, s));
  hipLaunchKernelGGL(col_sums_u8_kernel, dim3((unsigned)n_tiles), dim3(BNPK_BLOCK), 0, s, d_data, d_offsets, n_rows, total,
                     (const int64_t*)table, n_cols, reinterpret_cast<unsigned long long*>(d_sums),
                     reinterpret_cast<unsigned long long*>(d_counts));
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}
