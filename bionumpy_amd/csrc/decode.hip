// FASTQ / FASTA chunk decode on gfx950: newline census + ordered compaction (A2), entry validation (A3),
// field table (A4/A5).  All of it is byte/integer work bound by the HBM read of the raw chunk; every
// wavefront streams contiguous 1 KiB pieces (16 B per lane) and ranks its matches with lane shuffles.
#include "common.h"
#include "scan.h"

namespace {

constexpr int VEC = 16;                                  // bytes per lane per load
constexpr int WAVE_BYTES = BNPK_WAVE * VEC;              // 1 KiB per wavefront-instruction
constexpr int ITERS = 4;                                 // loads per lane per tile
constexpr int WAVES = BNPK_BLOCK / BNPK_WAVE;
constexpr int TILE_BYTES = WAVES * ITERS * WAVE_BYTES;   // 16 KiB per workgroup

// high bit of every byte of w that equals the byte replicated in `rep` (exact, no borrow artefacts)
__device__ __forceinline__ uint32_t match_bytes(uint32_t w, uint32_t rep) {
  uint32_t x = w ^ rep;
  uint32_t t = (x & 0x7f7f7f7fu) + 0x7f7f7f7fu;
  return ~(t | x | 0x7f7f7f7fu);
}

// 16-bit mask (bit j = byte pos+j matches) for the 16 bytes at `pos`; bytes >= n never match
__device__ __forceinline__ uint32_t match16(const uint8_t* __restrict__ buf, int64_t pos, int64_t n,
                                            uint32_t rep) {
  if (pos >= n) return 0;
  uint32_t m = 0;
  if (pos + VEC <= n) {
    uint4 v = *reinterpret_cast<const uint4*>(buf + pos);
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint32_t h = match_bytes(w[q], rep);
      // gather the 4 high bits (bits 7,15,23,31) into 4 consecutive bits
      uint32_t b = ((h >> 7) & 1u) | ((h >> 14) & 2u) | ((h >> 21) & 4u) | ((h >> 28) & 8u);
      m |= b << (4 * q);
    }
  } else {
    uint8_t value = (uint8_t)(rep & 0xff);
    for (int j = 0; j < VEC && pos + j < n; ++j)
      if (buf[pos + j] == value) m |= 1u << j;
  }
  return m;
}

__device__ __forceinline__ int64_t lane_pos(int64_t tile_base, int it) {
  return tile_base + (int64_t)wave_id() * (ITERS * WAVE_BYTES) + (int64_t)it * WAVE_BYTES + lane_id() * VEC;
}

__global__ __launch_bounds__(BNPK_BLOCK) void byte_census_kernel(const uint8_t* __restrict__ buf, int64_t n,
                                                                 uint32_t rep, int64_t* __restrict__ tile_counts) {
  __shared__ int smem[WAVES];
  int64_t tile_base = (int64_t)blockIdx.x * TILE_BYTES;
  int c = 0;
#pragma unroll
  for (int it = 0; it < ITERS; ++it) c += __popc(match16(buf, lane_pos(tile_base, it), n, rep));
  c = wave_reduce_sum(c);
  if (lane_id() == 0) smem[wave_id()] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < WAVES; ++w) t += smem[w];
    tile_counts[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(BNPK_BLOCK) void byte_positions_kernel(const uint8_t* __restrict__ buf, int64_t n,
                                                                    uint32_t rep,
                                                                    const int64_t* __restrict__ tile_offsets,
                                                                    int64_t limit, int64_t* __restrict__ out) {
  __shared__ int smem[WAVES];
  int64_t tile_base = (int64_t)blockIdx.x * TILE_BYTES;
  int64_t first = tile_offsets[blockIdx.x];
  if (first >= limit || tile_offsets[blockIdx.x + 1] == first) return;   // uniform per block
  uint32_t m[ITERS];
  int c = 0;
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    m[it] = match16(buf, lane_pos(tile_base, it), n, rep);
    c += __popc(m[it]);
  }
  int wave_total = wave_reduce_sum(c);
  if (lane_id() == 0) smem[wave_id()] = wave_total;
  __syncthreads();
  int64_t rank = first;
  for (int w = 0; w < wave_id(); ++w) rank += smem[w];
  // matches are ordered (wave, iteration, lane, byte): rank the lanes of each iteration with a wave scan
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    int pc = __popc(m[it]);
    int inc = wave_inclusive_scan(pc);
    int64_t r = rank + inc - pc;
    int64_t pos = lane_pos(tile_base, it);
    uint32_t mm = m[it];
    while (mm) {
      int j = __ffs(mm) - 1;
      mm &= mm - 1;
      if (r < limit) out[r] = pos + j;
      ++r;
    }
    rank += __shfl(inc, 63, 64);
  }
}

// one thread per entry (reference: strided gathers data[new_lines[..]+1])
__global__ void validate_entries_kernel(const uint8_t* __restrict__ buf, const int64_t* __restrict__ nl,
                                        int64_t n_entries, int lpe, uint8_t header, int check_plus,
                                        unsigned long long* __restrict__ err) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; r < n_entries; r += stride) {
    int64_t hpos = (r == 0) ? 0 : nl[r * lpe - 1] + 1;
    if (buf[hpos] != header) atomicMin(&err[0], (unsigned long long)r);
    if (check_plus) {
      int64_t ppos = nl[r * lpe + 1] + 1;
      if (buf[ppos] != '+') atomicMin(&err[1], (unsigned long long)r);
    }
    if (r < lpe && nl[0] >= 1) {      // _modify_for_carriage_return looks at the first lpe header lines only
      int64_t e = nl[r * lpe];
      if (e >= 1 && buf[e - 1] == '\r') err[2] = 1ull;
    }
  }
}

__global__ void init_err_kernel(int64_t* err) {
  err[0] = BNPK_NONE;
  err[1] = BNPK_NONE;
  err[2] = 0;
}

__global__ void field_table_kernel(const uint8_t* __restrict__ buf, const int64_t* __restrict__ nl,
                                   int64_t n_entries, int lpe, int field, int line_offset, int strip_cr,
                                   int64_t* __restrict__ starts, int64_t* __restrict__ lens) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; r < n_entries; r += stride) {
    int64_t line = r * lpe + field;
    int64_t s = (line == 0 ? 0 : nl[line - 1] + 1) + line_offset;
    int64_t e = nl[line];
    if (strip_cr && e >= 1 && buf[e - 1] == '\r') e -= 1;
    starts[r] = s;
    lens[r] = e - s;
  }
}

}  // namespace

extern "C" {

int64_t bnpk_scan_tiles(int64_t n_bytes) { return n_bytes <= 0 ? 0 : ceil_div(n_bytes, TILE_BYTES); }

int bnpk_byte_census(bnpk_ctx* ctx, const uint8_t* d_buf, int64_t n, uint8_t value, int64_t* d_tile_offsets,
                     void* stream) {
  if (!ctx || n < 0 || !d_tile_offsets || (n > 0 && !d_buf)) return BNPK_ERR_ARG;
  if (((uintptr_t)d_buf & 15) != 0) return BNPK_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  int64_t tiles = bnpk_scan_tiles(n);
  if (tiles > BNPK_MAX_BLOCKS) return BNPK_ERR_RANGE;
  void* scratch = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, bnpk_scan_scratch_bytes(tiles), &scratch, (hipStream_t)stream));
  uint32_t rep = 0x01010101u * value;
  {
    bnpk_timer t(ctx, "byte_census", s);
    if (tiles > 0)
      hipLaunchKernelGGL(byte_census_kernel, dim3((unsigned)tiles), dim3(BNPK_BLOCK), 0, s, d_buf, n, rep,
                         d_tile_offsets);
  }
  BNPK_HIP(ctx, hipGetLastError());
  bnpk_timer t2(ctx, "tile_offsets_scan", s);
  return bnpk_scan_launch(ctx, d_tile_offsets, tiles, 1, d_tile_offsets, true, (int64_t*)scratch, s);
}

int bnpk_byte_positions(bnpk_ctx* ctx, const uint8_t* d_buf, int64_t n, uint8_t value,
                        const int64_t* d_tile_offsets, int64_t limit, int64_t* d_pos, void* stream) {
  if (!ctx || n < 0 || !d_tile_offsets || (n > 0 && !d_buf) || limit < 0) return BNPK_ERR_ARG;
  if (((uintptr_t)d_buf & 15) != 0) return BNPK_ERR_ALIGN;
  int64_t tiles = bnpk_scan_tiles(n);
  if (tiles > BNPK_MAX_BLOCKS) return BNPK_ERR_RANGE;
  if (tiles == 0 || limit == 0) return BNPK_OK;
  if (!d_pos) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  bnpk_timer t(ctx, "byte_positions", s);
  hipLaunchKernelGGL(byte_positions_kernel, dim3((unsigned)tiles), dim3(BNPK_BLOCK), 0, s, d_buf, n,
                     0x01010101u * value, d_tile_offsets, limit, d_pos);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_validate_entries(bnpk_ctx* ctx, const uint8_t* d_buf, const int64_t* d_newlines, int64_t n_lines,
                          int lines_per_entry, uint8_t header, int check_plus, int64_t* d_err3, void* stream) {
  if (!ctx || !d_err3 || lines_per_entry < 1 || n_lines < 0 || n_lines % lines_per_entry != 0)
    return BNPK_ERR_ARG;
  if (check_plus && lines_per_entry < 3) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  bnpk_timer t(ctx, "validate_entries", s);
  hipLaunchKernelGGL(init_err_kernel, dim3(1), dim3(1), 0, s, d_err3);
  int64_t n_entries = n_lines / lines_per_entry;
  if (n_entries > 0) {
    if (!d_buf || !d_newlines) return BNPK_ERR_ARG;
    hipLaunchKernelGGL(validate_entries_kernel, dim3(grid_for(ceil_div(n_entries, 256))), dim3(256), 0, s, d_buf,
                       d_newlines, n_entries, lines_per_entry, header, check_plus,
                       reinterpret_cast<unsigned long long*>(d_err3));
  }
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_field_table(bnpk_ctx* ctx, const uint8_t* d_buf, const int64_t* d_newlines, int64_t n_entries,
                     int lines_per_entry, int field, int line_offset, int strip_cr, int64_t* d_starts,
                     int64_t* d_lens, void* stream) {
  if (!ctx || n_entries < 0 || lines_per_entry < 1 || field < 0 || field >= lines_per_entry) return BNPK_ERR_ARG;
  if (n_entries == 0) return BNPK_OK;
  if (!d_buf || !d_newlines || !d_starts || !d_lens) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  bnpk_timer t(ctx, "field_table", s);
  hipLaunchKernelGGL(field_table_kernel, dim3(grid_for(ceil_div(n_entries, 256))), dim3(256), 0, s, d_buf,
                     d_newlines, n_entries, lines_per_entry, field, line_offset, strip_cr, d_starts, d_lens);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

}  // extern "C"
