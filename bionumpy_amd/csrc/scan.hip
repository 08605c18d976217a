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

// One pass (round 5): a tile of 4096 items is read once, in 16-byte pieces that neighbouring lanes take from neighbouring
// addresses, scanned in that striped order (a wave scan per row of 512 items, sixteen row-by-wave partials combined through
// LDS), and gets the sum of the tiles before it by decoupled look-back: the tile's total is published in a status word, a
// wavefront reads the 64 words in front of it and adds up to the nearest one that already holds an inclusive prefix.  Tiles
// are numbered by a ticket taken when the workgroup STARTS, so every tile in front of a running one is running or done.
// (The three-kernel form read the input twice and wrote eight scattered words per lane: 0.6 ms per 50 M rows, this: ~0.25.)
constexpr unsigned long long SC_AGG = 1ull << 62, SC_INC = 2ull << 62, SC_VAL = (1ull << 62) - 1ull;
constexpr int SC_ROWS = 8, SC_TILE = SC_ROWS * 2 * BNPK_BLOCK;          // rows of 2 * BNPK_BLOCK items; 4096 per tile: the look-back chain advances ~50 tiles per microsecond whatever their size

__global__ __launch_bounds__(BNPK_BLOCK) void scan_onepass_kernel(const int64_t* __restrict__ in, int64_t n, int window,
                                                                  unsigned long long* __restrict__ status,
                                                                  unsigned* __restrict__ ticket, int64_t* __restrict__ out,
                                                                  int write_total) {
  __shared__ int64_t part[SC_ROWS * (BNPK_BLOCK / 64)];
  __shared__ int64_t before_sh;
  __shared__ unsigned tile_sh;
  if (threadIdx.x == 0) tile_sh = atomicAdd(ticket, 1u);
  __syncthreads();
  const int64_t tile = tile_sh;
  const int64_t base = tile * SC_TILE;
  const int lane = lane_id(), wave = wave_id();
  int64_t a[SC_ROWS], b[SC_ROWS], inc[SC_ROWS];
#pragma unroll
  for (int j = 0; j < SC_ROWS; ++j) {
    const int64_t i = base + (int64_t)j * (2 * BNPK_BLOCK) + 2 * threadIdx.x;
    if (i + 1 < n) {
      const longlong2 v = *reinterpret_cast<const longlong2*>(in + i);     // (i is even and the arrays are 16-byte aligned)
      a[j] = xform(v.x, window);
      b[j] = xform(v.y, window);
    } else {
      a[j] = i < n ? xform(in[i], window) : 0;
      b[j] = 0;
    }
    int64_t s = a[j] + b[j];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int64_t o = __shfl_up(s, d, 64);
      if (lane >= d) s += o;
    }
    inc[j] = s;
    if (lane == 63) part[j * (BNPK_BLOCK / 64) + wave] = s;
  }
  __syncthreads();
  int64_t mine = 0, total = 0;                              // the partials in front of this wavefront's rows, and all of them
  int64_t row_before[SC_ROWS];
#pragma unroll
  for (int j = 0; j < SC_ROWS; ++j) {
#pragma unroll
    for (int w = 0; w < BNPK_BLOCK / 64; ++w) {
      if (w == wave) row_before[j] = total;
      total += part[j * (BNPK_BLOCK / 64) + w];
    }
  }
  (void)mine;
  if (wave == 0) {
    if (lane == 0) __hip_atomic_store(status + tile, SC_AGG | (unsigned long long)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long sum = 0;
    int64_t hi = tile;
    while (hi > 0) {
      const int64_t i = hi - 1 - lane;
      unsigned long long v, need;
      int first_inc;
      while (true) {
        v = i >= 0 ? __hip_atomic_load(status + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : SC_INC;
        const unsigned long long incm = __ballot((v >> 62) == 2ull), none = __ballot((v >> 62) == 0ull);
        first_inc = incm ? __builtin_ctzll(incm) : 64;
        need = first_inc >= 63 ? ~0ull : ((1ull << (first_inc + 1)) - 1ull);
        if ((none & need) == 0ull) break;
        __builtin_amdgcn_s_sleep(1);
      }
      unsigned long long m = ((need >> lane) & 1ull) ? (v & SC_VAL) : 0ull;
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) m += __shfl_xor(m, d, 64);
      sum += m;
      if (first_inc < 64) break;
      hi -= 64;
    }
    if (lane == 0) {
      __hip_atomic_store(status + tile, SC_INC | (sum + (unsigned long long)total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      before_sh = (int64_t)sum;
    }
  }
  __syncthreads();
  const int64_t before = before_sh;
#pragma unroll
  for (int j = 0; j < SC_ROWS; ++j) {
    const int64_t i = base + (int64_t)j * (2 * BNPK_BLOCK) + 2 * threadIdx.x;
    const int64_t x = before + row_before[j] + inc[j] - (a[j] + b[j]);
    if (i + 1 < n) {
      longlong2 v;
      v.x = x;
      v.y = x + a[j];
      *reinterpret_cast<longlong2*>(out + i) = v;
    } else if (i < n) {
      out[i] = x;
    }
  }
  if (write_total && base + SC_TILE >= n && threadIdx.x == 0) out[n] = before + total;
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
  const bool aligned = (((uintptr_t)d_in | (uintptr_t)d_out) & 15) == 0;
  if (nb > 1 && aligned) {                                 // one pass: status words + the ticket live in the scratch
    const int64_t nt = ceil_div(n, (int64_t)SC_TILE);
    BNPK_HIP(ctx, hipMemsetAsync(d_scratch, 0, (size_t)(nt + 1) * 8, stream));
    hipLaunchKernelGGL(scan_onepass_kernel, dim3((unsigned)nt), dim3(BNPK_BLOCK), 0, stream, d_in, n, window,
                       reinterpret_cast<unsigned long long*>(d_scratch), reinterpret_cast<unsigned*>(d_scratch + nt), d_out,
                       write_total ? 1 : 0);
    BNPK_HIP(ctx, hipGetLastError());
    return BNPK_OK;
  }
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
