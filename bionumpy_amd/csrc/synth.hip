// Synthetic FASTQ generator (bench / test input) written straight into HBM.
// Every byte is a pure function of (seed, read id, position); bionumpy_amd/synth.py holds the numpy twin.
#include <algorithm>

#include "common.h"

namespace {

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x) {   // splitmix64 finaliser
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

constexpr uint64_t GENOME_SALT = 0x67656E6F6D65ull;   // "genome"

struct base_source {
  uint64_t seed;
  int mode;
  int64_t genome_len;
  int read_len;
  // cached 32-base block
  uint64_t blk_key = ~0ull, blk = 0;
  int64_t cur_read = -1;
  uint64_t read_key = 0;
  int64_t read_start = 0;

  __device__ __forceinline__ uint32_t base(int64_t read, int pos) {
    if (read != cur_read) {
      cur_read = read;
      read_key = mix64(seed + (uint64_t)read);
      if (mode == 1) read_start = (int64_t)(read_key % (uint64_t)(genome_len - read_len + 1));
      blk_key = ~0ull;
    }
    uint64_t stream, idx;
    if (mode == 0) { stream = read_key; idx = (uint64_t)pos; }
    else { stream = mix64(seed ^ GENOME_SALT); idx = (uint64_t)(read_start + pos); }
    uint64_t key = idx >> 5;
    if (key != blk_key) { blk_key = key; blk = mix64(stream + key); }
    return (uint32_t)(blk >> (2 * (idx & 31))) & 3u;
  }
};

__global__ __launch_bounds__(BNPK_BLOCK) void synth_fastq_kernel(uint8_t* __restrict__ out, int64_t first_read,
                                                                 int64_t n_reads, int read_len, uint64_t seed,
                                                                 int mode, int64_t genome_len) {
  const int64_t rec = 2 * (int64_t)read_len + 16;
  const int64_t total = n_reads * rec;
  int64_t p0 = ((int64_t)blockIdx.x * BNPK_BLOCK + threadIdx.x) * 16;
  if (p0 >= total) return;
  base_source src{seed, mode, genome_len, read_len};
  const uint32_t alphabet = 0x54474341u;   // 'A','C','G','T'
  uint64_t lo = 0, hi = 0;
  int64_t r = p0 / rec;
  int64_t off = p0 - r * rec;
  int cnt = (int)min((int64_t)16, total - p0);
  for (int j = 0; j < cnt; ++j) {
    uint32_t b;
    if (off == 0) b = '@';
    else if (off <= 10) {                        // 10-digit zero padded read id
      int64_t id = first_read + r;
      int64_t div = 1;
      for (int d = 0; d < 10 - (int)off; ++d) div *= 10;
      b = '0' + (uint32_t)((id / div) % 10);
    } else if (off == 11) b = '\n';
    else if (off < 12 + read_len) b = (alphabet >> (8 * src.base(first_read + r, (int)(off - 12)))) & 0xff;
    else if (off == 12 + read_len) b = '\n';
    else if (off == 13 + read_len) b = '+';
    else if (off == 14 + read_len) b = '\n';
    else if (off < rec - 1) b = 'I';
    else b = '\n';
    if (j < 8) lo |= (uint64_t)b << (8 * j); else hi |= (uint64_t)b << (8 * (j - 8));
    if (++off == rec) { off = 0; ++r; }
  }
  if (cnt == 16) {
    *reinterpret_cast<uint4*>(out + p0) = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
  } else {
    for (int j = 0; j < cnt; ++j) out[p0 + j] = (uint8_t)(j < 8 ? lo >> (8 * j) : hi >> (8 * (j - 8)));
  }
}

// the rate this chip streams at: dst[i] = src[i], 16 bytes per lane, non-temporal — what bench.py prints next to the 8 TB/s
// spec peak ("measure achievable with a device memcpy and state it")
__global__ __launch_bounds__(256) void copy_peak_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n16) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  typedef unsigned v4 __attribute__((ext_vector_type(4)));
  const v4* s4 = reinterpret_cast<const v4*>(src);
  v4* d4 = reinterpret_cast<v4*>(dst);
  for (; i < n16; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(&s4[i]), &d4[i]);
}

// the same copy in the forms a streaming kernel can take (bnpk_copy_rates): plain (temporal) accesses in a grid-stride loop,
// one 16-byte element per thread with no loop at all (the float4 copy MI355X_MICROARCH.md quotes at 6.29 TB/s), and four
// independent elements per thread and iteration
__global__ __launch_bounds__(256) void copy_plain_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n16) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n16; i += stride) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void copy_oneshot_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n16) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n16) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void copy_unrolled_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n16) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
  }
  for (; i < n16; i += stride) dst[i] = src[i];
}

}  // namespace

extern "C" {

int64_t bnpk_synth_record_bytes(int read_len) { return 2 * (int64_t)read_len + 16; }

int bnpk_synth_fastq(bnpk_ctx* ctx, uint8_t* d_out, int64_t first_read, int64_t n_reads, int read_len,
                     uint64_t seed, int mode, int64_t genome_len, void* stream) {
  if (!ctx || n_reads < 0 || read_len < 1 || first_read < 0 || (mode != 0 && mode != 1)) return BNPK_ERR_ARG;
  if (mode == 1 && genome_len < read_len) return BNPK_ERR_ARG;
  if (first_read + n_reads > 9999999999LL) return BNPK_ERR_RANGE;
  if (n_reads == 0) return BNPK_OK;
  if (!d_out) return BNPK_ERR_ARG;
  if (((uintptr_t)d_out & 15) != 0) return BNPK_ERR_ALIGN;
  int64_t total = n_reads * bnpk_synth_record_bytes(read_len);
  int64_t blocks = ceil_div(ceil_div(total, 16), BNPK_BLOCK);
  if (blocks > BNPK_MAX_BLOCKS) return BNPK_ERR_RANGE;
  hipStream_t s = (hipStream_t)stream;
  bnpk_timer t(ctx, "synth_fastq", s);
  hipLaunchKernelGGL(synth_fastq_kernel, dim3((unsigned)blocks), dim3(BNPK_BLOCK), 0, s, d_out, first_read, n_reads,
                     read_len, seed, mode, genome_len);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_copy_peak(bnpk_ctx* ctx, const void* d_src, void* d_dst, int64_t bytes, int reps, double* h_gb_per_s, void* stream) {
  if (!ctx || !d_src || !d_dst || bytes < 16 || reps < 1 || !h_gb_per_s) return BNPK_ERR_ARG;
  if (((uintptr_t)d_src & 15) || ((uintptr_t)d_dst & 15)) return BNPK_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  const int64_t n16 = bytes / 16;
  const unsigned grid = (unsigned)std::min<int64_t>((n16 + 255) / 256, (int64_t)ctx->compute_units * 64);
  hipEvent_t e0, e1;
  BNPK_HIP(ctx, hipEventCreate(&e0));
  BNPK_HIP(ctx, hipEventCreate(&e1));
  hipLaunchKernelGGL(copy_peak_kernel, dim3(grid), dim3(256), 0, s, (const uint4*)d_src, (uint4*)d_dst, n16);   // warm-up
  BNPK_HIP(ctx, hipEventRecord(e0, s));
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL(copy_peak_kernel, dim3(grid), dim3(256), 0, s, (const uint4*)d_src, (uint4*)d_dst, n16);
  BNPK_HIP(ctx, hipEventRecord(e1, s));
  BNPK_HIP(ctx, hipEventSynchronize(e1));
  float ms = 0.f;
  BNPK_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *h_gb_per_s = ms > 0 ? 2.0 * (double)(n16 * 16) * reps / (ms * 1e-3) / 1e9 : 0.0;   // bytes read + bytes written
  return BNPK_OK;
}

int bnpk_copy_rates(bnpk_ctx* ctx, const void* d_src, void* d_dst, int64_t bytes, int reps, double* h_gb_per_s4, void* stream) {
  if (!ctx || !d_src || !d_dst || bytes < 16 || reps < 1 || !h_gb_per_s4) return BNPK_ERR_ARG;
  if (((uintptr_t)d_src & 15) || ((uintptr_t)d_dst & 15)) return BNPK_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  const int64_t n16 = bytes / 16;
  if ((n16 + 255) / 256 > BNPK_MAX_BLOCKS) return BNPK_ERR_RANGE;
  const unsigned loop_grid = (unsigned)std::min<int64_t>((n16 + 255) / 256, (int64_t)ctx->compute_units * 64);
  const unsigned full_grid = (unsigned)((n16 + 255) / 256);
  const uint4* src = (const uint4*)d_src;
  uint4* dst = (uint4*)d_dst;
  hipEvent_t e0, e1;
  BNPK_HIP(ctx, hipEventCreate(&e0));
  BNPK_HIP(ctx, hipEventCreate(&e1));
  for (int v = 0; v < 4; ++v) {
    for (int r = -1; r < reps; ++r) {                   // (r = -1: warm-up)
      if (r == 0) BNPK_HIP(ctx, hipEventRecord(e0, s));
      if (v == 0) hipLaunchKernelGGL(copy_peak_kernel, dim3(loop_grid), dim3(256), 0, s, src, dst, n16);
      if (v == 1) hipLaunchKernelGGL(copy_plain_kernel, dim3(loop_grid), dim3(256), 0, s, src, dst, n16);
      if (v == 2) hipLaunchKernelGGL(copy_oneshot_kernel, dim3(full_grid), dim3(256), 0, s, src, dst, n16);
      if (v == 3) hipLaunchKernelGGL(copy_unrolled_kernel, dim3(loop_grid), dim3(256), 0, s, src, dst, n16);
    }
    BNPK_HIP(ctx, hipEventRecord(e1, s));
    BNPK_HIP(ctx, hipEventSynchronize(e1));
    float ms = 0.f;
    BNPK_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    h_gb_per_s4[v] = ms > 0 ? 2.0 * (double)(n16 * 16) * reps / (ms * 1e-3) / 1e9 : 0.0;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return BNPK_OK;
}

}  // extern "C"