// A13: multi-line FASTA on the device (bionumpy/io/multiline_buffer.py:33-106).
//
// Reading: from the ordered newline positions of a chunk (bnpk_byte_positions) the reference derives, with numpy
// expressions per LINE, where the chunk has to be cut (the last newline followed by '>'), which lines are headers,
// the header views, the sequence-line views and the length of every record's sequence (the sum of its lines).  Here
// one reduction finds the cut, one element-wise kernel classifies the lines, three exclusive scans number the
// records / the sequence lines / the sequence bytes, and one scatter kernel writes the tables — nothing per line
// happens on the host.
// Writing: from_data (multiline_buffer.py:68-86) as an output-flat kernel: every output byte finds its record (binary
// search over the record offsets) and is '>', a name byte, a sequence byte or the newline that ends a line of
// n_characters_per_line = 80 letters.
#include <algorithm>

#include "common.h"
#include "scan.h"

namespace {

constexpr uint8_t NEWLINE = 10, CR = 13;

// out[0] = 1 + largest i with buf[nl[i] + 1] == marker (0 if none), out[1] = number of such i
__global__ __launch_bounds__(BNPK_BLOCK) void ml_cut_kernel(const uint8_t* __restrict__ buf, const int64_t* __restrict__ nl,
                                                            int64_t n_nl, uint8_t marker, unsigned long long* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  long long last = -1, count = 0;
  for (; i < n_nl; i += stride)
    if (buf[nl[i] + 1] == marker) { last = i; ++count; }
  last = wave_reduce_max(last);
  count = wave_reduce_sum(count);
  if (lane_id() == 0) {
    if (last >= 0) atomicMax(&out[0], (unsigned long long)(last + 1));
    if (count) atomicAdd(&out[1], (unsigned long long)count);
  }
}

// line i of the cut chunk (i <= n_nl; line n_nl ends at size - 1): start, length without the newline (and without a
// carriage return before it if strip_cr), header flag; the three scan inputs
__global__ __launch_bounds__(BNPK_BLOCK) void ml_lines_kernel(const uint8_t* __restrict__ buf, int64_t size,
                                                              const int64_t* __restrict__ nl, int64_t n_nl, uint8_t marker,
                                                              int strip_cr, int64_t* __restrict__ is_header,
                                                              int64_t* __restrict__ is_seq, int64_t* __restrict__ seq_bytes) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i <= n_nl; i += stride) {
    const int64_t start = i ? nl[i - 1] + 1 : 0;
    int64_t end = i < n_nl ? nl[i] : size - 1;
    if (strip_cr && end > 0 && buf[end - 1] == CR) --end;
    const bool header = i == 0 || buf[start] == marker;
    is_header[i] = header ? 1 : 0;
    is_seq[i] = header ? 0 : 1;
    seq_bytes[i] = header ? 0 : end - start;
  }
}

__global__ __launch_bounds__(BNPK_BLOCK) void ml_scatter_kernel(const uint8_t* __restrict__ buf, int64_t size,
                                                                const int64_t* __restrict__ nl, int64_t n_nl, uint8_t marker,
                                                                int strip_cr, const int64_t* __restrict__ rec_of,
                                                                const int64_t* __restrict__ seq_of,
                                                                const int64_t* __restrict__ bytes_before,
                                                                int64_t* __restrict__ header_starts,
                                                                int64_t* __restrict__ header_lens,
                                                                int64_t* __restrict__ rec_bytes_before,
                                                                int64_t* __restrict__ seq_starts,
                                                                int64_t* __restrict__ seq_lens) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i <= n_nl; i += stride) {
    const int64_t start = i ? nl[i - 1] + 1 : 0;
    int64_t end = i < n_nl ? nl[i] : size - 1;
    if (strip_cr && end > 0 && buf[end - 1] == CR) --end;
    if (i == 0 || buf[start] == marker) {
      const int64_t r = rec_of[i];
      header_starts[r] = start + 1;
      header_lens[r] = end - start - 1;
      rec_bytes_before[r] = bytes_before[i];
    } else {
      const int64_t j = seq_of[i];
      seq_starts[j] = start;
      seq_lens[j] = end - start;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) rec_bytes_before[rec_of[n_nl + 1]] = bytes_before[n_nl + 1];
}

// rec_lens[r] = rec_bytes_before[r + 1] - rec_bytes_before[r]
__global__ void ml_diff_kernel(const int64_t* __restrict__ csum, int64_t n, int64_t* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = csum[i + 1] - csum[i];
}

// from_data: record r = '>' name '\n' then its sequence in lines of `width` letters, each ended by '\n'
__global__ __launch_bounds__(BNPK_BLOCK) void ml_wrap_kernel(const uint8_t* __restrict__ names, const int64_t* __restrict__ name_off,
                                                             const uint8_t* __restrict__ seq, const int64_t* __restrict__ seq_off,
                                                             int64_t n_rec, const int64_t* __restrict__ out_off, int64_t total,
                                                             int width, uint8_t marker, uint8_t* __restrict__ out) {
  int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; o < total; o += stride) {
    const int64_t r = find_row(out_off, 0, n_rec - 1, o);
    const int64_t p = o - out_off[r];
    const int64_t name_len = name_off[r + 1] - name_off[r];
    uint8_t c;
    if (p == 0) c = marker;
    else if (p <= name_len) c = names[name_off[r] + p - 1];
    else if (p == name_len + 1) c = NEWLINE;
    else {
      const int64_t q = p - (name_len + 2), line = q / (width + 1), col = q - line * (width + 1);
      const int64_t at = line * width + col, seq_len = seq_off[r + 1] - seq_off[r];
      c = (col < width && at < seq_len) ? seq[seq_off[r] + at] : NEWLINE;
    }
    out[o] = c;
  }
}

// out_off[r] = bytes of the records before r: name + 2, sequence, one newline per line
__global__ void ml_wrap_sizes_kernel(const int64_t* __restrict__ name_off, const int64_t* __restrict__ seq_off, int64_t n_rec,
                                     int width, int64_t* __restrict__ sizes) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; r < n_rec; r += stride) {
    const int64_t s = seq_off[r + 1] - seq_off[r];
    sizes[r] = (name_off[r + 1] - name_off[r]) + 2 + s + (s + width - 1) / width;
  }
}

}  // namespace

extern "C" {

int bnpk_multiline_cut(bnpk_ctx* ctx, const uint8_t* d_buf, const int64_t* d_newlines, int64_t n_newlines, uint8_t marker,
                       int64_t* h_last_entry_newline, int64_t* h_n_entry_newlines, void* stream) {
  if (!ctx || n_newlines < 0 || !h_last_entry_newline || !h_n_entry_newlines) return BNPK_ERR_ARG;
  *h_last_entry_newline = -1;
  *h_n_entry_newlines = 0;
  if (n_newlines == 0) return BNPK_OK;
  if (!d_buf || !d_newlines) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  void* scratch = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, 16, &scratch, (hipStream_t)stream));
  BNPK_HIP(ctx, hipMemsetAsync(scratch, 0, 16, s));
  {
    bnpk_timer t(ctx, "multiline_cut", s);
    hipLaunchKernelGGL(ml_cut_kernel, dim3(grid_for(std::min<int64_t>(ceil_div(n_newlines, 256), 4096))), dim3(256), 0, s,
                       d_buf, d_newlines, n_newlines, marker, (unsigned long long*)scratch);
    BNPK_HIP(ctx, hipGetLastError());
  }
  long long host[2];
  BNPK_HIP(ctx, hipMemcpyAsync(host, scratch, 16, hipMemcpyDeviceToHost, s));
  BNPK_HIP(ctx, hipStreamSynchronize(s));
  *h_last_entry_newline = host[0] - 1;
  *h_n_entry_newlines = host[1];
  return BNPK_OK;
}

int bnpk_multiline_table(bnpk_ctx* ctx, const uint8_t* d_buf, int64_t size, const int64_t* d_newlines, int64_t n_newlines,
                         uint8_t marker, int strip_cr, int64_t* d_header_starts, int64_t* d_header_lens,
                         int64_t* d_record_lens, int64_t* d_seq_line_starts, int64_t* d_seq_line_lens, int64_t* h_totals3,
                         void* stream) {
  if (!ctx || size < 1 || n_newlines < 0 || !d_buf || (n_newlines > 0 && !d_newlines) || !d_header_starts ||
      !d_header_lens || !d_record_lens || !d_seq_line_starts || !d_seq_line_lens || !h_totals3)
    return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int64_t n_lines = n_newlines + 1;
  // scratch: three scan arrays of n_lines + 1, the per-record cumulative byte counts (<= n_lines + 1), scan partials
  const size_t arr = (size_t)(n_lines + 1) * 8;
  void* scratch = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, 4 * arr + bnpk_scan_scratch_bytes(n_lines) + 64, &scratch, (hipStream_t)stream));
  int64_t* rec_of = (int64_t*)scratch;
  int64_t* seq_of = rec_of + (n_lines + 1);
  int64_t* bytes_before = seq_of + (n_lines + 1);
  int64_t* rec_csum = bytes_before + (n_lines + 1);
  int64_t* partials = rec_csum + (n_lines + 1);
  const unsigned grid = grid_for(std::min<int64_t>(ceil_div(n_lines, 256), 8192));
  bnpk_timer t(ctx, "multiline_table", s);
  hipLaunchKernelGGL(ml_lines_kernel, dim3(grid), dim3(256), 0, s, d_buf, size, d_newlines, n_newlines, marker, strip_cr,
                     rec_of, seq_of, bytes_before);
  BNPK_CHECK(bnpk_scan_launch(ctx, rec_of, n_lines, 1, rec_of, true, partials, s));
  BNPK_CHECK(bnpk_scan_launch(ctx, seq_of, n_lines, 1, seq_of, true, partials, s));
  BNPK_CHECK(bnpk_scan_launch(ctx, bytes_before, n_lines, 1, bytes_before, true, partials, s));
  hipLaunchKernelGGL(ml_scatter_kernel, dim3(grid), dim3(256), 0, s, d_buf, size, d_newlines, n_newlines, marker, strip_cr,
                     (const int64_t*)rec_of, (const int64_t*)seq_of, (const int64_t*)bytes_before, d_header_starts,
                     d_header_lens, rec_csum, d_seq_line_starts, d_seq_line_lens);
  BNPK_HIP(ctx, hipGetLastError());
  int64_t totals[3];
  BNPK_HIP(ctx, hipMemcpyAsync(&totals[0], rec_of + n_lines, 8, hipMemcpyDeviceToHost, s));
  BNPK_HIP(ctx, hipMemcpyAsync(&totals[1], seq_of + n_lines, 8, hipMemcpyDeviceToHost, s));
  BNPK_HIP(ctx, hipMemcpyAsync(&totals[2], bytes_before + n_lines, 8, hipMemcpyDeviceToHost, s));
  BNPK_HIP(ctx, hipStreamSynchronize(s));
  hipLaunchKernelGGL(ml_diff_kernel, dim3(grid_for(std::min<int64_t>(ceil_div(totals[0], 256), 4096))), dim3(256), 0, s,
                     (const int64_t*)rec_csum, totals[0], d_record_lens);
  BNPK_HIP(ctx, hipGetLastError());
  h_totals3[0] = totals[0];
  h_totals3[1] = totals[1];
  h_totals3[2] = totals[2];
  return BNPK_OK;
}

int bnpk_multiline_wrap(bnpk_ctx* ctx, const uint8_t* d_names, const int64_t* d_name_offsets, const uint8_t* d_seq,
                        const int64_t* d_seq_offsets, int64_t n_records, int width, uint8_t marker, int64_t* d_out_offsets,
                        uint8_t* d_out, int64_t out_capacity, int64_t* h_total, void* stream) {
  if (!ctx || n_records < 0 || width < 1 || !h_total) return BNPK_ERR_ARG;
  *h_total = 0;
  if (n_records == 0) return BNPK_OK;
  if (!d_name_offsets || !d_seq_offsets || !d_out_offsets) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  void* scratch = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, bnpk_scan_scratch_bytes(n_records) + 64, &scratch, (hipStream_t)stream));
  bnpk_timer t(ctx, "multiline_wrap", s);
  hipLaunchKernelGGL(ml_wrap_sizes_kernel, dim3(grid_for(std::min<int64_t>(ceil_div(n_records, 256), 4096))), dim3(256), 0, s,
                     d_name_offsets, d_seq_offsets, n_records, width, d_out_offsets);
  BNPK_CHECK(bnpk_scan_launch(ctx, d_out_offsets, n_records, 1, d_out_offsets, true, (int64_t*)scratch, s));
  int64_t total = 0;
  BNPK_HIP(ctx, hipMemcpyAsync(&total, d_out_offsets + n_records, 8, hipMemcpyDeviceToHost, s));
  BNPK_HIP(ctx, hipStreamSynchronize(s));
  *h_total = total;
  if (!d_out) return BNPK_OK;                            // (first call: sizes only)
  if (out_capacity < total || !d_names || !d_seq) return BNPK_ERR_ARG;
  hipLaunchKernelGGL(ml_wrap_kernel, dim3(grid_for(std::min<int64_t>(ceil_div(total, 256), (int64_t)ctx->compute_units * 32))),
                     dim3(256), 0, s, d_names, d_name_offsets, d_seq, d_seq_offsets, n_records, (const int64_t*)d_out_offsets,
                     total, width, marker, d_out);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

}  // extern "C"
