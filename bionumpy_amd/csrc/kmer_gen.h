// Device helpers that turn a flat output index of the ragged k-mer array into a k-mer value read from the
// packed 2-bit reads (shared by kmers.hip and partition.hip).
#pragma once
#include "common.h"

#ifdef __HIPCC__
namespace {

struct row_cursor {
  int64_t row, out_end, in_pos;
};

__device__ __forceinline__ row_cursor seek_row(const int64_t* __restrict__ in_off,
                                               const int64_t* __restrict__ out_off, int64_t rlo, int64_t rhi,
                                               int64_t o) {
  row_cursor c;
  c.row = find_row(out_off, rlo, rhi, o);
  c.out_end = out_off[c.row + 1];
  c.in_pos = in_off[c.row] + (o - out_off[c.row]);
  return c;
}

// advance the cursor from output o-1 to output o
__device__ __forceinline__ void next_output(row_cursor& c, const int64_t* __restrict__ in_off,
                                            const int64_t* __restrict__ out_off, int64_t o) {
  if (o < c.out_end) { c.in_pos += 1; return; }
  do {
    ++c.row;
    c.out_end = out_off[c.row + 1];
  } while (o >= c.out_end);
  c.in_pos = in_off[c.row];
}

struct word_window {
  int64_t wi = -2;
  uint64_t lo = 0, hi = 0;
};

// the 64 bits starting at base `pos` of the packed stream
__device__ __forceinline__ uint64_t bits_at(const uint64_t* __restrict__ W, int64_t pos, word_window& ww) {
  int64_t wi = pos >> 5;
  if (wi != ww.wi) {
    if (wi == ww.wi + 1) { ww.lo = ww.hi; ww.hi = W[wi + 1]; }
    else { ww.lo = W[wi]; ww.hi = W[wi + 1]; }
    ww.wi = wi;
  }
  int sh = 2 * (int)(pos & 31);
  return sh ? (ww.lo >> sh) | (ww.hi << (64 - sh)) : ww.lo;
}

}  // namespace
#endif
