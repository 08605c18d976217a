// FASTQ / FASTA chunk decode on gfx950: newline census + ordered compaction (A2), entry validation (A3),
// field table (A4/A5).  All of it is byte/integer work bound by the HBM read of the raw chunk; every
// wavefront streams contiguous 1 KiB pieces (16 B per lane) and ranks its matches with lane shuffles.
#include <algorithm>

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

__device__ __forceinline__ uint32_t nomatch16_of(uint4 v, uint32_t rep) { return nomatch16(v.x, v.y, v.z, v.w, rep); }

// 16-bit mask (bit j = byte pos+j matches) for the 16 bytes at `pos`; bytes >= n never match
__device__ __forceinline__ uint32_t match16(const uint8_t* __restrict__ buf, int64_t pos, int64_t n,
                                            uint32_t rep) {
  if (pos >= n) return 0;
  uint32_t m = 0;
  if (pos + VEC <= n) {
    m = ~nomatch16_of(*reinterpret_cast<const uint4*>(buf + pos), rep) & 0xffffu;
  } else {
    uint8_t value = (uint8_t)(rep & 0xff);
    for (int j = 0; j < VEC && pos + j < n; ++j)
      if (buf[pos + j] == value) m |= 1u << j;
  }
  return m;
}

// bit j of the result = byte j of v matches
__device__ __forceinline__ uint32_t match_vec(uint4 v, uint32_t rep) { return ~nomatch16(v.x, v.y, v.z, v.w, rep) & 0xffffu; }

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

// VALIDATE: the newline table of a FASTQ / FASTA chunk and the checks of OneLineBuffer._validate / FastQBuffer._validate
// (bionumpy/io/one_line_buffer.py:156-173, fastq_buffer.py:39-45) in the same pass: the thread that finds the newline in
// front of a line looks at the line's first byte — it is in the tile it holds, or the byte behind it — instead of a second
// kernel that gathers two table entries and two text bytes per entry from all over memory (2.4 ms per 50 M reads: 0.05 of
// the HBM peak).  err[0] / err[1]: first entry whose header / third line starts wrongly (atomicMin); err[2]: bit 0 = a header
// line of the first lpe entries ends in '\r', bit 1 = the first line is empty (then the reference leaves '\r' alone:
// _modify_for_carriage_return, one_line_buffer.py:176-182).
template <bool VALIDATE>
__global__ __launch_bounds__(BNPK_BLOCK) void byte_positions_kernel(const uint8_t* __restrict__ buf, int64_t n,
                                                                    uint32_t rep,
                                                                    const int64_t* __restrict__ tile_offsets,
                                                                    int64_t limit, int64_t* __restrict__ out, int lpe,
                                                                    uint8_t header, int check_plus,
                                                                    unsigned long long* __restrict__ err) {
  __shared__ int smem[WAVES];
  int64_t tile_base = (int64_t)blockIdx.x * TILE_BYTES;
  int64_t first = tile_offsets[blockIdx.x];
  if (VALIDATE && blockIdx.x == 0 && threadIdx.x == 0 && limit >= lpe && buf[0] != header) atomicMin(&err[0], 0ull);
  if (first >= limit || tile_offsets[blockIdx.x + 1] == first) return;   // uniform per block
  uint32_t m[ITERS];
  uint4 text[VALIDATE ? ITERS : 1];                          // VALIDATE: the sixteen bytes themselves stay in registers: the byte that
  uint32_t behind[VALIDATE ? ITERS : 1];                     // starts the next line is one of them, or the next lane's first
  int c = 0;
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int64_t pos = lane_pos(tile_base, it);
    if (VALIDATE && pos + VEC <= n) {
      text[it] = *reinterpret_cast<const uint4*>(buf + pos);
      m[it] = match_vec(text[it], rep);
      behind[it] = text[it].x & 0xffu;
    } else {
      m[it] = match16(buf, pos, n, rep);
      if (VALIDATE) {
        text[it] = make_uint4(0u, 0u, 0u, 0u);
        behind[it] = 0x100u;                                 // (no vector here: whoever asks reads memory)
      }
    }
    c += __popc(m[it]);
  }
  if (VALIDATE) {
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const uint32_t next = __shfl_down(behind[it], 1, 64);
      behind[it] = lane_id() == 63 ? 0x100u : next;            // (lane 63: the first byte of another wavefront's chunk)
    }
  }
  int wave_total = wave_reduce_sum(c);
  if (lane_id() == 0) smem[wave_id()] = wave_total;
  __syncthreads();
  int rank = 0;                                                        // (ranks and positions relative to the tile: 32 bits)
  for (int w = 0; w < wave_id(); ++w) rank += smem[w];
  const int lim = (int)min(limit - first, (int64_t)1 << 30);
  int64_t* __restrict__ out_t = out + first;
  const int first_phase = VALIDATE ? (int)(first % lpe) : 0;           // (line index modulo lines per entry, in 32 bits from here)
  const bool pow2 = (lpe & (lpe - 1)) == 0;                            // 4 (FASTQ), 2 (FASTA): a mask instead of a division per newline
  // matches are ordered (wave, iteration, lane, byte): rank the lanes of each iteration with a wave scan
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    int pc = __popc(m[it]);
    int inc = wave_inclusive_scan(pc);
    int r = rank + inc - pc;
    const int64_t pos = lane_pos(tile_base, it);
    uint32_t mm = m[it];
    while (mm) {
      int j = __ffs(mm) - 1;
      mm &= mm - 1;
      if (r < lim) {
        const int64_t p = pos + j;
        out_t[r] = p;
        if (VALIDATE) {
          const unsigned ph = (unsigned)r + (unsigned)first_phase;                          // (< 16384 + lpe)
          const int phase = pow2 ? (int)(ph & (unsigned)(lpe - 1)) : (int)(ph % (unsigned)lpe);  // of the line this newline ends
          if (r + 1 < lim) {                                 // the line behind it belongs to a complete entry
            const int next = phase + 1 == lpe ? 0 : phase + 1;
            if (next == 0 || (check_plus && next == 2)) {
              uint32_t b;
              if (pos + VEC > n) {
                b = 0x100u;
              } else if (j < 15) {
                const int q = (j + 1) >> 2;
                const uint32_t w = q == 0 ? text[it].x : q == 1 ? text[it].y : q == 2 ? text[it].z : text[it].w;
                b = (w >> (8 * ((j + 1) & 3))) & 0xffu;
              } else {
                b = behind[it];
              }
              if (b > 0xffu) b = buf[p + 1];
              if (next == 0 && b != header) atomicMin(&err[0], (unsigned long long)((first + r + 1) / lpe));
              if (next == 2 && b != '+') atomicMin(&err[1], (unsigned long long)((first + r + 1) / lpe));
            }
          }
          if (phase == 0 && first + r < (int64_t)lpe * lpe) {   // the header line of one of the first lpe entries
            if (first + r == 0 && p == 0) atomicOr(&err[2], 2ull);
            if (p >= 1 && buf[p - 1] == '\r') atomicOr(&err[2], 1ull);
          }
        }
      }
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


// ---- the chunks of a reader with small windows, cut out of one big batch -------------------------------------------------
// out: [0] rows written, [1] status (1 = a window without a complete entry grew past max_chunk), [2] the window end of the last
// chunk cut (what the reference's reader would hold as its left-over is [end of that chunk, this)), [3] entries consumed;
// rows from word 4 on: {entries up to and including the chunk, end byte of the chunk, has_cr of the chunk, window end}.
// One lane walks the windows (they depend on each other); the newline table is searched by all 64 lanes at once
// (a 64-ary search: three or four rounds of one load per lane instead of twenty dependent loads).
__device__ __forceinline__ int64_t wc_lower_bound(const int64_t* __restrict__ nl, int64_t n, int64_t value) {
  // number of newlines at a position < value
  int64_t lo = 0, hi = n;                                // the answer lies in [lo, hi]
  while (hi - lo > 64) {
    const int64_t step = (hi - lo + 63) / 64;
    const int64_t at = lo + (int64_t)(lane_id() + 1) * step - 1;          // lane l probes the end of its piece
    const bool below = at < hi && nl[at] < value;
    const unsigned long long m = __ballot(below);
    const int pieces = __popcll(m);                      // leading pieces that lie entirely below `value`
    const int64_t new_lo = lo + (int64_t)pieces * step;
    const int64_t new_hi = min(hi, new_lo + step);
    lo = min(new_lo, hi);
    hi = new_hi < lo ? lo : new_hi;
  }
  const int64_t at = lo + lane_id();
  const bool below = at < hi && nl[at] < value;
  return lo + __popcll(__ballot(below));
}

__global__ __launch_bounds__(64) void window_cuts_kernel(const uint8_t* __restrict__ buf, const int64_t* __restrict__ nl,
                                                         int64_t n_lines, int lpe, int64_t window, int64_t avail, int finished,
                                                         int64_t first_held, int64_t max_chunk, int max_cuts,
                                                         int64_t* __restrict__ out) {
  int64_t j = 0, s = 0, held = first_held, last_end = first_held, rows = 0, status = 0;
  const int64_t n_entries = n_lines / lpe;
  while (rows < max_cuts && j < n_entries) {
    // the reference reads min_chunk_size - held new bytes (min_chunk_size if it holds that much already): parser.py:117-120
    int64_t w_end = s + (held >= window ? held + window : window);
    int64_t j_end = j;
    bool stop = false;
    while (true) {
      if (w_end > avail) {
        if (!finished) { stop = true; break; }           // the window reaches past what is here: the next batch's business
        w_end = avail;
      }
      if (max_chunk > 0 && w_end - s > max_chunk) { status = 1; stop = true; break; }
      j_end = wc_lower_bound(nl, n_lines, w_end) / lpe;
      if (j_end > j || w_end >= avail) break;
      w_end += window;                                   // no complete entry yet: min_chunk_size more (parser.py:128-131)
    }
    if (stop || j_end <= j) break;
    const int64_t e = nl[j_end * lpe - 1] + 1;
    // _modify_for_carriage_return (one_line_buffer.py:176-182) on this chunk: the first lpe header lines
    int has_cr = 0;
    if (nl[j * lpe] - s >= 1) {
      for (int r = 0; r < lpe && j + r < j_end; ++r) {
        const int64_t le = nl[(j + r) * lpe];
        if (le - s >= 1 && buf[le - 1] == '\r') has_cr = 1;
      }
    }
    if (lane_id() == 0) {
      out[4 + 4 * rows] = j_end;
      out[5 + 4 * rows] = e;
      out[6 + 4 * rows] = has_cr;
      out[7 + 4 * rows] = w_end;
    }
    ++rows;
    held = w_end - e;
    last_end = w_end;
    s = e;
    j = j_end;
  }
  if (lane_id() == 0) {
    out[0] = rows;
    out[1] = status;
    out[2] = last_end;
    out[3] = j;
  }
}

// out[i] = nl[i] - (start byte of the chunk that holds line i), for the lines of the chunks listed in `cuts` (as above)
__global__ __launch_bounds__(BNPK_BLOCK) void rebase_lines_kernel(const int64_t* __restrict__ nl, const int64_t* __restrict__ cuts,
                                                                  int lpe, int64_t* __restrict__ out) {
  __shared__ int64_t ends[256], starts[256];
  const int rows = (int)cuts[0];
  for (int i = threadIdx.x; i < rows; i += blockDim.x) {
    ends[i] = cuts[4 + 4 * i] * lpe;                     // lines up to and including chunk i
    starts[i] = i == 0 ? 0 : cuts[5 + 4 * (i - 1)];
  }
  __syncthreads();
  const int64_t n = rows ? ends[rows - 1] : 0;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    int lo = 0, hi = rows - 1;                           // first chunk whose lines end behind i
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (ends[mid] > i) hi = mid; else lo = mid + 1;
    }
    out[i] = nl[i] - starts[lo];
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
  hipLaunchKernelGGL(byte_positions_kernel<false>, dim3((unsigned)tiles), dim3(BNPK_BLOCK), 0, s, d_buf, n,
                     0x01010101u * value, d_tile_offsets, limit, d_pos, 1, (uint8_t)0, 0, (unsigned long long*)nullptr);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_line_positions(bnpk_ctx* ctx, const uint8_t* d_buf, int64_t n, const int64_t* d_tile_offsets, int64_t n_lines,
                        int lines_per_entry, uint8_t header, int check_plus, int64_t* d_pos, int64_t* d_err3, void* stream) {
  if (!ctx || n < 0 || !d_tile_offsets || (n > 0 && !d_buf) || n_lines < 0 || !d_err3 || lines_per_entry < 1 ||
      n_lines % lines_per_entry != 0 || (check_plus && lines_per_entry < 3))
    return BNPK_ERR_ARG;
  if (((uintptr_t)d_buf & 15) != 0) return BNPK_ERR_ALIGN;
  int64_t tiles = bnpk_scan_tiles(n);
  if (tiles > BNPK_MAX_BLOCKS) return BNPK_ERR_RANGE;
  hipStream_t s = (hipStream_t)stream;
  bnpk_timer t(ctx, "line_positions", s);
  hipLaunchKernelGGL(init_err_kernel, dim3(1), dim3(1), 0, s, d_err3);
  if (tiles > 0 && n_lines > 0) {
    if (!d_pos) return BNPK_ERR_ARG;
    hipLaunchKernelGGL(byte_positions_kernel<true>, dim3((unsigned)tiles), dim3(BNPK_BLOCK), 0, s, d_buf, n, 0x01010101u * (uint32_t)'\n',
                       d_tile_offsets, n_lines, d_pos, lines_per_entry, header, check_plus, reinterpret_cast<unsigned long long*>(d_err3));
  }
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

int64_t bnpk_window_cuts_words(int max_cuts) { return 4 + 4 * (int64_t)std::max(max_cuts, 0); }

int bnpk_window_cuts(bnpk_ctx* ctx, const uint8_t* d_buf, const int64_t* d_newlines, int64_t n_lines, int lines_per_entry,
                     int64_t window, int64_t avail, int finished, int64_t first_held, int64_t max_chunk, int max_cuts,
                     int64_t* d_out, void* stream) {
  if (!ctx || !d_out || lines_per_entry < 1 || n_lines < 0 || n_lines % lines_per_entry != 0 || window < 1 || avail < 0 ||
      first_held < 0 || max_cuts < 1 || max_cuts > 256)
    return BNPK_ERR_ARG;
  if (n_lines > 0 && (!d_buf || !d_newlines)) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  bnpk_timer t(ctx, "window_cuts", s);
  hipLaunchKernelGGL(window_cuts_kernel, dim3(1), dim3(64), 0, s, d_buf, d_newlines, n_lines, lines_per_entry, window, avail,
                     finished, first_held, max_chunk, max_cuts, d_out);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_rebase_lines(bnpk_ctx* ctx, const int64_t* d_newlines, int64_t n_lines, const int64_t* d_cuts, int lines_per_entry,
                      int64_t* d_out, void* stream) {
  if (!ctx || n_lines < 0 || lines_per_entry < 1) return BNPK_ERR_ARG;
  if (n_lines == 0) return BNPK_OK;
  if (!d_newlines || !d_cuts || !d_out) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  bnpk_timer t(ctx, "rebase_lines", s);
  hipLaunchKernelGGL(rebase_lines_kernel, dim3(grid_for(std::min<int64_t>(ceil_div(n_lines, BNPK_BLOCK), 2048))), dim3(BNPK_BLOCK), 0, s,
                     d_newlines, d_cuts, lines_per_entry, d_out);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

}  // extern "C"
