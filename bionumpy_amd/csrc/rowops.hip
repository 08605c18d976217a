// Per-row reductions over ragged uint8 data (quality scores): np.sum / np.mean / np.min / np.max(ragged, axis=-1)
// as used by the read filters of the reference (scripts/small_example.py:36-46: np.mean(chunk.quality, axis=1) > 30,
// np.min(chunk.quality, axis=1) > 10).  Rows are short (a read), so eight lanes share a row: each takes aligned
// dwords of the row's byte range, masks the bytes outside it, and the partial sums / minima / maxima meet in three
// shuffle steps.
#include "common.h"
#include "rows.h"

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
constexpr int JL_BYTES_PER_LANE = 16;
constexpr int JL_CHUNKS = 4;                            // chunks of 16 bytes per lane: the staging is paid once per 16 KiB
constexpr int64_t JL_TILE = (int64_t)BNPK_BLOCK * JL_BYTES_PER_LANE * JL_CHUNKS;

struct jl_line {
  const uint8_t* data;         // flat bytes of the field (nullptr: a one-byte constant line)
  const int64_t* off;          // its row offsets
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

// The entries a workgroup's 4 KiB of output come from (a dozen for FASTQ records) are staged in LDS first — their offsets
// and, per line, the offsets of their field rows — with coalesced loads: without that every lane walks a chain of up to
// ten dependent global loads (binary search, entry bounds, two offsets per line) before it touches a byte, and the kernel
// runs at the rate of that latency.  A lane whose sixteen bytes lie inside one field (17 in 20 for 150-base reads) takes
// them with ONE unaligned 16-byte load; the others assemble theirs from spans of up to eight bytes.
constexpr int JL_ROWS = 256;

__global__ __launch_bounds__(BNPK_BLOCK) void join_lines_kernel(jl_lines lines, int n_lines, uint8_t header,
                                                                const int64_t* __restrict__ entry_off, int64_t n_rows,
                                                                int64_t total, const int64_t* __restrict__ tile_rows,
                                                                uint8_t* __restrict__ out) {
  __shared__ int64_t s_ent[JL_ROWS + 1];
  __shared__ int64_t s_off[JL_MAX_LINES][JL_ROWS + 1];
  const int64_t blk0 = (int64_t)blockIdx.x * JL_TILE;
  if (blk0 >= total) return;
  const int64_t lo = tile_rows[blockIdx.x];
  const int64_t hi = (blk0 + JL_TILE < total) ? tile_rows[blockIdx.x + 1] : n_rows - 1;
  const bool staged = hi - lo + 1 <= JL_ROWS;
  if (staged) {
    const int n_stage = (int)(hi - lo + 1);
    for (int i = threadIdx.x; i <= n_stage; i += BNPK_BLOCK) {
      s_ent[i] = entry_off[lo + i];
      for (int l = 0; l < n_lines; ++l)
        if (lines.l[l].data) s_off[l][i] = lines.l[l].off[lo + i];
    }
    __syncthreads();
  }
  auto EO = [&](int64_t r) -> int64_t { return staged ? s_ent[r - lo] : entry_off[r]; };
  auto FO = [&](int l, int64_t r) -> int64_t { return staged ? s_off[l][r - lo] : lines.l[l].off[r]; };
#pragma unroll 1
  for (int chunk = 0; chunk < JL_CHUNKS; ++chunk) {
  const int64_t p0 = blk0 + ((int64_t)chunk * BNPK_BLOCK + threadIdx.x) * JL_BYTES_PER_LANE;
  if (p0 >= total) return;
  int64_t r;
  {
    int64_t a = lo, b = hi;                                    // last row with EO(row) <= p0
    while (a < b) {
      const int64_t mid = a + ((b - a + 1) >> 1);
      if (EO(mid) <= p0) a = mid; else b = mid - 1;
    }
    r = a;
  }
  const int64_t p1 = min(p0 + JL_BYTES_PER_LANE, total);
  int64_t e0 = EO(r), e1 = EO(r + 1);
  int64_t p = p0;
  uint64_t word[3] = {0, 0, 0};                                // the lane's sixteen output bytes, stored once
  // m (1..8) bytes, the low bytes of x, go to the lane's next output positions
  auto emit = [&](uint64_t x, int m) {
    if (m < 8) x &= (1ull << (8 * m)) - 1ull;
    const int j = (int)(p - p0), sh = 8 * (j & 7);
    word[j >> 3] |= x << sh;
    if (sh) word[(j >> 3) + 1] |= x >> (64 - sh);
    p += m;
  };
  auto add_bytes = [&](uint64_t x, int add) -> uint64_t {     // per-byte wrap-around addition without carries between bytes
    const uint64_t a = (uint64_t)(add & 0xff) * 0x0101010101010101ull;
    return ((x & 0x7f7f7f7f7f7f7f7full) + (a & 0x7f7f7f7f7f7f7f7full)) ^ ((x ^ a) & 0x8080808080808080ull);
  };
  // A lane whose sixteen bytes cross line boundaries (four lanes in ten for FASTQ records) first walks the lines WITHOUT
  // touching the fields: constants (header byte, newline, fill byte) go straight into the word, and what has to come from
  // a field is noted as a piece {source, place, length, add}.  Then the loads of all pieces are issued together — 16
  // unaligned bytes each — and shifted into place.  The walk used to wait for every span's load before it looked at the
  // next line: two to four dependent round trips that the other lanes of the wavefront waited for.
  struct piece_t { const uint8_t* src; int j, m, add; bool wide; };
  constexpr int MAX_PIECES = 3;
  piece_t pieces[MAX_PIECES];
  int n_pieces = 0;
  auto place = [&](uint64_t lo, uint64_t hi, int j, int m) {   // bytes 0 .. m-1 of (lo, hi) to output positions j .. j+m-1
    if (m < 8) { lo &= (1ull << (8 * m)) - 1ull; hi = 0; }
    else if (m < 16) hi &= (1ull << (8 * (m - 8))) - 1ull;
    const int sh = 8 * (j & 7);
    if (j < 8) {
      word[0] |= lo << sh;
      word[1] |= (sh ? lo >> (64 - sh) : 0ull) | (hi << sh);
    } else {
      word[1] |= lo << sh;
    }
  };
  bool overflow = false;
  while (p < p1 && !overflow) {
    while (e1 <= p) { ++r; e0 = e1; e1 = EO(r + 1); }
    int64_t t = p - e0;                                        // offset inside entry r
    for (int i = 0; i < n_lines && p < p1; ++i) {
      const jl_line& L = lines.l[i];
      const int64_t fs = L.data ? FO(i, r) : 0;
      const int64_t flen = L.data ? FO(i, r + 1) - fs : 1;
      const int64_t line_len = L.prefix + flen + 1;
      if (t >= line_len) { t -= line_len; continue; }
      if (t < L.prefix) { emit(header, 1); ++t; }
      if (p < p1 && t < L.prefix + flen) {
        const int m = (int)min((int64_t)L.prefix + flen - t, p1 - p);
        if (!L.data) {
          emit(L.fill, 1);                                     // (a line without a field is one constant byte)
        } else {
          if (n_pieces == MAX_PIECES) { overflow = true; break; }
          const int64_t at = fs + (t - L.prefix);
          const piece_t pc = {L.data + at, (int)(p - p0), m, L.add, at + 16 <= L.off[n_rows]};   // wide: 16 bytes can be read there
#pragma unroll
          for (int q = 0; q < MAX_PIECES; ++q)                 // (static indices: the pieces stay in registers)
            if (q == n_pieces) pieces[q] = pc;
          ++n_pieces;
          p += m;
        }
        t += m;
      }
      if (p < p1 && t == L.prefix + flen) emit(10, 1);
      t = 0;                                                   // the next line starts at its first byte
    }
  }
  uint64_t got[MAX_PIECES][2];
#pragma unroll
  for (int q = 0; q < MAX_PIECES; ++q) {
    got[q][0] = got[q][1] = 0;
    if (q < n_pieces) {
      if (pieces[q].wide) {
        __builtin_memcpy(got[q], pieces[q].src, 16);
      } else {
        for (int b = 0; b < pieces[q].m; ++b) got[q][b >> 3] |= (uint64_t)pieces[q].src[b] << (8 * (b & 7));
      }
    }
  }
#pragma unroll
  for (int q = 0; q < MAX_PIECES; ++q) {
    if (q < n_pieces) {
      uint64_t lo = got[q][0], hi = got[q][1];
      if (pieces[q].add) { lo = add_bytes(lo, pieces[q].add); hi = add_bytes(hi, pieces[q].add); }
      place(lo, hi, pieces[q].j, pieces[q].m);
    }
  }
  // (more than MAX_PIECES fields in sixteen bytes — fields of a byte or two: the rest of the lane's bytes the slow way)
  while (p < p1) {
    while (e1 <= p) { ++r; e0 = e1; e1 = EO(r + 1); }
    int64_t t = p - e0;
    for (int i = 0; i < n_lines && p < p1; ++i) {
      const jl_line& L = lines.l[i];
      const int64_t fs = L.data ? FO(i, r) : 0;
      const int64_t flen = L.data ? FO(i, r + 1) - fs : 1;
      const int64_t line_len = L.prefix + flen + 1;
      if (t >= line_len) { t -= line_len; continue; }
      if (t < L.prefix) { emit(header, 1); ++t; }
      while (p < p1 && t < L.prefix + flen) {
        const int m = (int)min(min((int64_t)L.prefix + flen - t, p1 - p), (int64_t)8);
        uint64_t x;
        if (!L.data) {
          x = L.fill;
        } else {
          const uint8_t* src = L.data + fs + (t - L.prefix);
          if (fs + (t - L.prefix) + 8 <= L.off[n_rows]) {
            __builtin_memcpy(&x, src, 8);
          } else {                                             // the last bytes of the field's buffer
            x = 0;
            for (int q = 0; q < m; ++q) x |= (uint64_t)src[q] << (8 * q);
          }
          if (L.add) x = add_bytes(x, L.add);
        }
        emit(x, m);
        t += m;
      }
      if (p < p1 && t == L.prefix + flen) emit(10, 1);
      t = 0;
    }
  }
  if (p1 - p0 == JL_BYTES_PER_LANE) {                          // (p0 is a multiple of 16, the buffer 16-byte aligned)
    *reinterpret_cast<uint4*>(out + p0) = make_uint4((uint32_t)word[0], (uint32_t)(word[0] >> 32), (uint32_t)word[1], (uint32_t)(word[1] >> 32));
  } else {
    for (int j = 0; j < (int)(p1 - p0); ++j) out[p0 + j] = (uint8_t)(word[j >> 3] >> (8 * (j & 7)));
  }
  }
}

}  // namespace

extern "C" int bnpk_join_lines(bnpk_ctx* ctx, int64_t n_rows, int n_lines, const uint8_t* const* d_field_data,
                               const int64_t* const* d_field_offsets, const int* add, const int* prefix,
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
    lines.l[i] = jl_line{d_field_data[i], d_field_offsets[i], add[i], prefix[i], fill[i]};
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
