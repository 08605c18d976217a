// Ragged gather fused with ASCII -> 2-bit DNA encode (A6 + A7) for gfx950.
// Output-flat mapping: one lane produces one packed uint64 (32 bases) so that a wavefront stores
// 512 contiguous bytes of packed words (and 2 KiB of 1-byte codes); the lane walks the source rows
// it overlaps, finding its first row by a binary search narrowed to the rows the workgroup touches.
#include <algorithm>

#include "common.h"
#include "rows.h"

namespace {

constexpr int BASES_PER_WORD = 32;

__device__ __forceinline__ uint32_t dna_code(uint32_t b, bool* ok) {
  uint32_t u = b & 0xDFu;                      // fold lower case onto upper case (exact for A C G T)
  *ok = (u == 'A') | (u == 'C') | (u == 'G') | (u == 'T');
  return ((u >> 1) & 3u) ^ ((u >> 2) & 1u);    // A C G T -> 0 1 2 3
}

constexpr uint64_t REP01 = 0x0101010101010101ull;
constexpr uint64_t REP7F = 0x7f7f7f7f7f7f7f7full;
constexpr uint64_t REP80 = 0x8080808080808080ull;

// 0x80 in every byte of x that equals the byte replicated in rep (exact SWAR compare)
__device__ __forceinline__ uint64_t eq_bytes(uint64_t x, uint64_t rep) {
  uint64_t z = x ^ rep;
  uint64_t t = (z & REP7F) + REP7F;
  return ~(t | z | REP7F);
}

// up to 8 source bytes -> byte codes (0..3, invalid -> 0) + mask of invalid bytes (0x80 per bad byte)
__device__ __forceinline__ uint64_t dna_codes8(uint64_t x, uint64_t lanes, uint64_t* bad) {
  uint64_t u = x & (0xDFull * REP01);                       // fold lower case onto upper case
  uint64_t valid = eq_bytes(u, 'A' * REP01) | eq_bytes(u, 'C' * REP01) | eq_bytes(u, 'G' * REP01) |
                   eq_bytes(u, 'T' * REP01);
  *bad = ~valid & REP80 & lanes;
  uint64_t c = ((u >> 1) & (3ull * REP01)) ^ ((u >> 2) & REP01);   // A C G T -> 0 1 2 3
  return c & ((valid >> 7) * 0xFFull) & lanes;
}

// eight byte codes (2 significant bits each) -> 16 packed bits
__device__ __forceinline__ uint64_t compress_codes8(uint64_t c) {
  uint64_t t = (c | (c >> 6)) & 0x000F000F000F000Full;
  t = (t | (t >> 12)) & 0x000000FF000000FFull;
  return (t | (t >> 24)) & 0xFFFFull;
}

// m (1..8) bytes at src as a little-endian uint64; one unaligned 8-byte load when the row still has 8 bytes
__device__ __forceinline__ uint64_t load_upto8(const uint8_t* __restrict__ src, int seg, int* m) {
  uint64_t x;
  if (seg >= 8) {
    __builtin_memcpy(&x, src, 8);
    *m = 8;
  } else {
    x = 0;
    for (int q = 0; q < seg; ++q) x |= (uint64_t)src[q] << (8 * q);
    *m = seg;
  }
  return x;
}

// 32 source bytes -> the packed word of their codes, a bit per invalid byte, and (optionally) the byte codes
__device__ __forceinline__ void encode32(const uint64_t (&x)[4], uint64_t& word, unsigned& bad32, uint64_t (&cw)[4]) {
  word = 0;
  bad32 = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    uint64_t badm;
    const uint64_t c = dna_codes8(x[q], ~0ull, &badm);
    cw[q] = c;
    word |= compress_codes8(c) << (16 * q);
    bad32 |= (unsigned)((((badm >> 7) * 0x0102040810204080ull) >> 56) & 0xffull) << (8 * q);
  }
}

// The same 32 bytes as eight dwords, a few instructions each: V_PERM_B32 looks the letter of every code up again and the
// XOR with the (case-folded) text is the validity test; a dot product packs four codes into a byte (as fastq.hip's
// fq_codes4).  encode32 above costs ~380 vector instructions and was, with one lane in five running it twice more for
// a row boundary, what bounded this kernel (6.1 ms per 50 M reads at 4 cycles an instruction), not memory.
__device__ __forceinline__ void encode32_perm(const uint32_t (&x)[8], uint64_t& word, unsigned& bad32, uint32_t (&cw)[8]) {
  uint32_t z[8], zany = 0, lo = 0, hi = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const uint32_t u = x[q] & 0xDFDFDFDFu;                     // fold lower case onto upper case (exact for A C G T)
    const uint32_t c = ((u >> 1) & 0x03030303u) ^ ((u >> 2) & 0x01010101u);
    z[q] = u ^ __builtin_amdgcn_perm(0u, 0x54474341u, c);      // 'A' 'C' 'G' 'T' selected by the code
    zany |= z[q];
    cw[q] = c;
    const uint32_t b = __builtin_amdgcn_udot4(c, 0x40100401u, 0u, false);
    if (q < 4) lo |= b << (8 * q); else hi |= b << (8 * (q - 4));
  }
  bad32 = 0;
  if (zany) {                                                  // some byte is no base: its code is 0, its bit is set
    lo = hi = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      uint32_t keep = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if ((z[q] >> (8 * j)) & 0xffu) bad32 |= 1u << (4 * q + j); else keep |= 0xffu << (8 * j);
      }
      cw[q] &= keep;
      const uint32_t b = __builtin_amdgcn_udot4(cw[q], 0x40100401u, 0u, false);
      if (q < 4) lo |= b << (8 * q); else hi |= b << (8 * (q - 4));
    }
  }
  word = (uint64_t)lo | ((uint64_t)hi << 32);
}

constexpr int GE_WPL = 2;                                      // packed words per lane: a tile is 256 * GE_WPL words
constexpr int GE_ROWS = 1024;                                  // rows of a tile (16384 bases) staged in LDS

// The rows a workgroup's 16384 bases come from (~110 reads) are staged in LDS — offsets relative to the tile, starts —
// with one coalesced load each; a lane finds the row of its 32 bases there.  Four lanes in five have all 32 inside one
// row: two unaligned 16-byte loads, issued together.  Most of the others straddle ONE row boundary, k bases before it
// and a next row of at least 32 - k: they load the 32 bytes at their position in the first row as well (the bytes past
// the row's end are whatever follows it in the buffer) and the 32 bytes that begin k bytes BEFORE the next row, and
// take byte i from the first load for i < k and from the second otherwise — one select per dword; every lane then
// encodes its 32 bytes once (encode32_perm).  Only lanes over short rows, or whose two loads would leave the buffer,
// walk row segments with 8-byte loads as rounds 1-2 did for every lane — after a binary search over the offsets in
// GLOBAL memory: 12.7 ms per 50 M reads, a chain of ten dependent loads per lane, 0.09 of the HBM peak.
template <bool WRITE_CODES, bool WRITE_PACKED>
__global__ __launch_bounds__(BNPK_BLOCK) void gather_encode_kernel(
    const uint8_t* __restrict__ buf, int64_t buf_size, const int64_t* __restrict__ starts, const int64_t* __restrict__ offsets,
    int64_t n_rows, int64_t total, const int64_t* __restrict__ tile_rows, int64_t n_tiles,
    uint8_t* __restrict__ codes, uint64_t* __restrict__ packed, unsigned* __restrict__ ends32,
    unsigned long long* __restrict__ err) {
  __shared__ int rel[GE_ROWS + 1];                             // offsets[rr0 + i] - first base of the tile (clamped below)
  __shared__ int64_t st[GE_ROWS];
  __shared__ int64_t first_off;
  int64_t rr[2];
  const int64_t n_words = (total + BASES_PER_WORD - 1) / BASES_PER_WORD;
  const int64_t n_alloc = total / BASES_PER_WORD + 2;          // words of `packed`: the pad word(s) are read by the k-mer kernel
  const int64_t w0 = (int64_t)blockIdx.x * (BNPK_BLOCK * GE_WPL);
  if (w0 >= n_words) {
    for (int64_t w = w0 + threadIdx.x; w < n_alloc; w += BNPK_BLOCK) {
      if (WRITE_PACKED) packed[w] = 0;
      if (ends32) ends32[w] = 0;
    }
    return;
  }
  tile_row_range(tile_rows, blockIdx.x, n_tiles, n_rows, rr[0], rr[1]);
  const int64_t blk_first = w0 * BASES_PER_WORD;
  // (32-bit halves: see gather_rows_kernel for the compiler bug a 64-bit uniform select runs into here)
  const uint64_t row_span = (uint64_t)(rr[1] - rr[0]);
  unsigned span_lo = (unsigned)row_span, span_hi = (unsigned)(row_span >> 32);
  asm volatile("" : "+s"(span_lo), "+s"(span_hi));
  const bool staged = span_hi == 0u && span_lo < (unsigned)GE_ROWS;
  const int n_stage = staged ? (int)span_lo + 1 : 0;
  if (staged) {
    for (int i = threadIdx.x; i <= n_stage; i += BNPK_BLOCK) {
      const int64_t d = offsets[rr[0] + i] - blk_first;
      rel[i] = (int)max(d, (int64_t)-(1 << 30));
      if (i < n_stage) st[i] = starts[rr[0] + i];
      if (i == 0) first_off = offsets[rr[0]];
    }
    __syncthreads();
  }
  unsigned long long bad = (unsigned long long)BNPK_NONE;
#pragma unroll
  for (int it = 0; it < GE_WPL; ++it) {
  const int64_t w = w0 + it * BNPK_BLOCK + threadIdx.x;
  if (w >= n_words) {
    if (WRITE_PACKED && w < n_alloc) packed[w] = 0;
    if (ends32 && w < n_alloc) ends32[w] = 0;
    continue;
  }
  unsigned endbits = 0;                                      // bit i: base i of this word is the last of its row
  int64_t pos = w * BASES_PER_WORD;
  const int64_t end = min(pos + BASES_PER_WORD, total);
  uint64_t word = 0;
  uint64_t cw[5] = {0, 0, 0, 0, 0};
  bool done = false;
  int64_t row;
  if (staged) {
    const int at = (int)(pos - blk_first);
    int lo = 0, hi = n_stage - 1;                              // last i with rel[i] <= at (skips empty rows)
    while (lo < hi) {
      const int mid = lo + ((hi - lo + 1) >> 1);
      if (rel[mid] <= at) lo = mid; else hi = mid - 1;
    }
    row = rr[0] + lo;
    const int row_end = rel[lo + 1];
    const int64_t start_r = lo == 0 ? first_off - blk_first : (int64_t)rel[lo];      // (tile-relative; may lie far before the tile)
    if (end - pos == BASES_PER_WORD) {
      uint32_t x[8], c8[8];
      const int k = row_end - at;                              // bases of the row from here on (>= 1)
      const int64_t from = st[lo] + ((int64_t)at - start_r);
      if (k >= BASES_PER_WORD) {                               // all 32 bases inside the row
        __builtin_memcpy(x, buf + from, 16);
        __builtin_memcpy(x + 4, buf + from + 16, 16);
        endbits = k == BASES_PER_WORD ? 1u << 31 : 0u;
        done = true;
      } else if (lo + 2 <= n_stage && rel[lo + 2] - row_end >= BASES_PER_WORD - k && from + BASES_PER_WORD <= buf_size &&
                 st[lo + 1] >= k && st[lo + 1] - k + BASES_PER_WORD <= buf_size) {   // one boundary, k bases before it
        uint32_t y[8];
        const uint8_t* head = buf + (st[lo + 1] - k);
        __builtin_memcpy(x, buf + from, 16);
        __builtin_memcpy(x + 4, buf + from + 16, 16);
        __builtin_memcpy(y, head, 16);
        __builtin_memcpy(y + 4, head + 16, 16);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int nb = min(max(k - 4 * q, 0), 4);            // bytes of this dword that come from the first row
          const uint32_t m = nb == 4 ? ~0u : (1u << (8 * nb)) - 1u;
          x[q] = (x[q] & m) | (y[q] & ~m);
        }
        endbits = (1u << (k - 1)) | (rel[lo + 2] - row_end == BASES_PER_WORD - k ? 1u << 31 : 0u);
        done = true;
      }
      if (done) {
        unsigned bad32;
        encode32_perm(x, word, bad32, c8);
        if (bad32) bad = min(bad, (unsigned long long)(pos + (__ffs((int)bad32) - 1)));
        if (WRITE_CODES) {
#pragma unroll
          for (int q = 0; q < 4; ++q) cw[q] = (uint64_t)c8[2 * q] | ((uint64_t)c8[2 * q + 1] << 32);
        }
      }
    }
  } else {
    row = find_row(offsets, rr[0], rr[1], pos);
  }
  if (!done) {
    int64_t row_end = offsets[row + 1];
    const uint8_t* src = buf + starts[row] + (pos - offsets[row]);
    int j = 0;                                  // bases produced so far
    while (pos < end) {
      while (pos >= row_end) {                  // next non-empty row
        ++row;
        row_end = offsets[row + 1];
        src = buf + starts[row];
      }
      int seg = (int)min(row_end - pos, end - pos);
      while (seg > 0) {
        int m;
        uint64_t x = load_upto8(src, seg, &m);
        uint64_t lanes = (m == 8) ? ~0ull : ((1ull << (8 * m)) - 1ull);
        uint64_t badm;
        uint64_t c = dna_codes8(x, lanes, &badm);
        if (badm) {
          unsigned long long at = (unsigned long long)(pos + ((__ffsll((long long)badm) - 1) >> 3));
          if (at < bad) bad = at;
        }
        if (WRITE_CODES) {
          int sh = 8 * (j & 7);
          cw[j >> 3] |= c << sh;
          if (sh) cw[(j >> 3) + 1] |= c >> (64 - sh);
        }
        word |= compress_codes8(c) << (2 * j);
        j += m; src += m; pos += m; seg -= m;
      }
      if (pos == row_end) endbits |= 1u << (j - 1);
    }
  }
  if (ends32) ends32[w] = endbits;
  if (WRITE_PACKED) packed[w] = word;
  if (WRITE_CODES) {
    int64_t p0 = w * BASES_PER_WORD;
    if (p0 + BASES_PER_WORD <= total) {
      uint4* dst = reinterpret_cast<uint4*>(codes + p0);
      dst[0] = make_uint4((uint32_t)cw[0], (uint32_t)(cw[0] >> 32), (uint32_t)cw[1], (uint32_t)(cw[1] >> 32));
      dst[1] = make_uint4((uint32_t)cw[2], (uint32_t)(cw[2] >> 32), (uint32_t)cw[3], (uint32_t)(cw[3] >> 32));
    } else {
      for (int q = 0; p0 + q < total; ++q) codes[p0 + q] = (uint8_t)(cw[q >> 3] >> (8 * (q & 7)));
    }
  }
  }
  if (bad != (unsigned long long)BNPK_NONE) atomicMin(err, bad);
}

// Rows [r0, r1) of a compact packed DNA ragged array as a compact packed array of their own: the bases
// [first, first + n_bases) of the packed stream shifted down to bit 0 (zero behind them) and the rows' offsets minus `first`.
// (What a chunk of a batch that was encoded as a whole takes: the rows are contiguous in the batch's stream.)
__global__ __launch_bounds__(BNPK_BLOCK) void packed_rows_slice_kernel(const uint64_t* __restrict__ packed, int64_t n_words_in,
                                                                       const int64_t* __restrict__ offsets, int64_t r0, int64_t n_rows,
                                                                       int64_t first, int64_t n_bases, uint64_t* __restrict__ out,
                                                                       int64_t* __restrict__ out_offsets) {
  const int64_t n_words = n_bases / 32 + 2;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int sh = 2 * (int)(first & 31);
  const int64_t w0 = first >> 5;
  for (int64_t w = i; w < n_words; w += stride) {
    const int64_t a = w0 + w;
    uint64_t v = a < n_words_in ? packed[a] >> sh : 0ull;
    if (sh && a + 1 < n_words_in) v |= packed[a + 1] << (64 - sh);
    const int64_t left = n_bases - 32 * w;                   // bases of this word that belong to the slice
    if (left <= 0) v = 0;
    else if (left < 32) v &= (1ull << (2 * left)) - 1ull;
    out[w] = v;
  }
  for (int64_t r = i; r <= n_rows; r += stride) out_offsets[r] = offsets[r0 + r] - first;
}

// plain gather (optionally subtracting a constant from every byte): 4 x 16 output bytes per lane.
// The rows a workgroup's 16 KiB of output come from (usually ~50) are staged in LDS first — their offsets relative to
// the block and their starts — with one coalesced load each: without that every lane walks a chain of six to eight
// dependent global loads (binary search, row end, start, data) and the kernel runs at the rate of that latency.  What is
// left of the chain (tile table -> offsets -> data) is paid once per 16 KiB: a lane finds the rows of its four chunks in
// LDS, issues the four data loads together, and only then looks at them.  A chunk inside one row (19 in 20 for rows of
// ~300 bytes) is ONE unaligned 16-byte load; the others are assembled from row segments.  Blocks with more rows than the
// table holds (rows of a few bytes) keep the direct path.
constexpr int GR_ROWS = 1024;
#ifndef GR_CHUNKS_N
#define GR_CHUNKS_N 4
#endif
constexpr int GR_PER = 16, GR_CHUNKS = GR_CHUNKS_N;
constexpr int64_t GR_TILE = (int64_t)BNPK_BLOCK * GR_PER * GR_CHUNKS;

__device__ __forceinline__ uint64_t gr_subtract(uint64_t x, uint64_t sub) {   // per-byte wrap-around subtraction, no borrows
  return ((x | REP80) - (sub & ~REP80)) ^ ((x ^ ~sub) & REP80);
}

__global__ __launch_bounds__(BNPK_BLOCK) void gather_rows_kernel(
    const uint8_t* __restrict__ buf, const int64_t* __restrict__ starts, const int64_t* __restrict__ offsets,
    int64_t n_rows, int64_t total, int subtract, const int64_t* __restrict__ tile_rows,
    uint8_t* __restrict__ out) {
  __shared__ int rel[GR_ROWS + 1];                             // offsets[rr0 + i] - blk_first (clamped below)
  __shared__ int64_t st[GR_ROWS];                              // starts[rr0 + i]
  __shared__ int64_t first_off;                                // offsets[rr0] (rel[0] is clamped: the row may start gigabytes back)
  int64_t rr[2];
  const int64_t blk_first = (int64_t)blockIdx.x * GR_TILE;
  if (blk_first >= total) return;
  tile_row_range(tile_rows, blockIdx.x, gridDim.x, n_rows, rr[0], rr[1]);
  // rows rr0 .. rr0 + n_stage - 1, if they all fit.  In 32-bit halves on purpose: hipcc 7.2 turns a wave-uniform SELECT on
  // a 64-bit signed compare (`min(rr[1] - rr[0] + 1, GR_ROWS + 1)`, `diff < GR_ROWS ? diff + 1 : 0`) into V_CMP + S_CSELECT
  // and loses the copy of VCC into SCC when the same compare also feeds a branch — the select then reads the carry of
  // whatever scalar add came last (n_stage was 1025, or 0, whatever the tile held).  32-bit compares are S_CMPs.
  const uint64_t row_span = (uint64_t)(rr[1] - rr[0]);
  unsigned span_lo = (unsigned)row_span, span_hi = (unsigned)(row_span >> 32);
  asm volatile("" : "+s"(span_lo), "+s"(span_hi));            // (or the optimiser fuses the halves into the 64-bit compare again)
  const bool staged = span_hi == 0u && span_lo < (unsigned)GR_ROWS;
  const int n_stage = staged ? (int)span_lo + 1 : 0;
  if (staged) {
    for (int i = threadIdx.x; i <= n_stage; i += BNPK_BLOCK) {
      const int64_t d = offsets[rr[0] + i] - blk_first;
      rel[i] = (int)max(d, (int64_t)-(1 << 30));
      if (i < n_stage) st[i] = starts[rr[0] + i];
      if (i == 0) first_off = offsets[rr[0]];
    }
    __syncthreads();
  }
  const uint64_t sub = (uint64_t)(subtract & 0xff) * REP01;
  auto store16 = [&](int64_t p0, const uint64_t (&v)[3]) {
    if (p0 + GR_PER <= total && (((uintptr_t)(out + p0)) & 15) == 0) {
      *reinterpret_cast<uint4*>(out + p0) = make_uint4((uint32_t)v[0], (uint32_t)(v[0] >> 32), (uint32_t)v[1], (uint32_t)(v[1] >> 32));
    } else {
      for (int q = 0; p0 + q < total && q < GR_PER; ++q) out[p0 + q] = (uint8_t)(v[q >> 3] >> (8 * (q & 7)));
    }
  };
  if (staged) {
#pragma unroll 1
    for (int c = 0; c < GR_CHUNKS; ++c) {
      const int at = (c * BNPK_BLOCK + (int)threadIdx.x) * GR_PER;
      const int64_t pos = blk_first + at;
      if (pos >= total) break;
      int lo = 0, hi = n_stage - 1;                            // last i with rel[i] <= at (skips empty rows)
      while (lo < hi) {
        const int mid = lo + ((hi - lo + 1) >> 1);
        if (rel[mid] <= at) lo = mid; else hi = mid - 1;
      }
      int row = lo, p = at, j = 0;
      const int e = (int)(min(pos + GR_PER, total) - blk_first);
      int row_end = rel[row + 1];
      const uint8_t* src = buf + st[row] + (row == 0 ? pos - first_off : (int64_t)(p - rel[row]));
      uint64_t v[3] = {0, 0, 0};
      if (row_end - p >= GR_PER && e - p == GR_PER) {
        // the lane's 16 bytes lie in ONE row (19 lanes in 20 for rows of ~300 bytes): one unaligned 16-byte load instead
        // of the byte-assembling walk over row segments below
        uint64_t a[2];
        __builtin_memcpy(a, src, 16);
        v[0] = gr_subtract(a[0], sub);
        v[1] = gr_subtract(a[1], sub);
        p = e;
      }
      if (p < e && e - p == GR_PER && row + 2 <= n_stage) {
        // ONE row boundary inside the chunk, k bytes before it, and both rows at least 16 bytes long (every boundary
        // chunk of FASTQ records): the 16 bytes that END row `row` and the 16 that START the next one, two independent
        // loads, shifted together — instead of a walk of dependent loads of eight, then single bytes, that the other 63
        // lanes of the wavefront wait for
        const int k = row_end - p;
        const int64_t start_r = row == 0 ? first_off - blk_first : (int64_t)rel[row];
        if (k < GR_PER && row_end - start_r >= GR_PER && rel[row + 2] - row_end >= GR_PER) {
          uint64_t a[2], b[2];
          __builtin_memcpy(a, buf + st[row] + (row_end - start_r) - GR_PER, 16);
          __builtin_memcpy(b, buf + st[row + 1], 16);
          // out = (A >> 8 (16 - k)) | (B << 8 k) over 128 bits, 0 < k < 16
          const int ra = 8 * (GR_PER - k), lb = 8 * k;
          uint64_t lo = ra >= 64 ? (a[1] >> (ra - 64)) : ((a[0] >> ra) | (a[1] << (64 - ra)));
          uint64_t hi = ra >= 64 ? 0ull : (a[1] >> ra);
          lo |= lb >= 64 ? 0ull : (b[0] << lb);
          hi |= lb >= 64 ? (b[0] << (lb - 64)) : ((b[1] << lb) | (b[0] >> (64 - lb)));
          v[0] = gr_subtract(lo, sub);
          v[1] = gr_subtract(hi, sub);
          p = e;
        }
      }
      while (p < e) {
        while (p >= row_end) {
          ++row;
          row_end = rel[row + 1];
          src = buf + st[row];
        }
        int seg = min(row_end - p, e - p);
        while (seg > 0) {
          int m;
          uint64_t x = gr_subtract(load_upto8(src, seg, &m), sub);
          if (m < 8) x &= (1ull << (8 * m)) - 1ull;
          const int sh = 8 * (j & 7);
          v[j >> 3] |= x << sh;
          if (sh) v[(j >> 3) + 1] |= x >> (64 - sh);
          j += m; src += m; p += m; seg -= m;
        }
      }
      store16(pos, v);
    }
    return;
  }
#pragma unroll 1
  for (int c = 0; c < GR_CHUNKS; ++c) {
    int64_t pos = blk_first + (int64_t)(c * BNPK_BLOCK + (int)threadIdx.x) * GR_PER;
    if (pos >= total) continue;
    const int64_t p0 = pos, end = min(pos + GR_PER, total);
    uint64_t v[3] = {0, 0, 0};
    int j = 0;
    int64_t row = find_row(offsets, rr[0], rr[1], pos);
    int64_t row_end = offsets[row + 1];
    const uint8_t* src = buf + starts[row] + (pos - offsets[row]);
    while (pos < end) {
      while (pos >= row_end) {
        ++row;
        row_end = offsets[row + 1];
        src = buf + starts[row];
      }
      int seg = (int)min(row_end - pos, end - pos);
      while (seg > 0) {
        int m;
        uint64_t x = gr_subtract(load_upto8(src, seg, &m), sub);
        if (m < 8) x &= (1ull << (8 * m)) - 1ull;
        const int sh = 8 * (j & 7);
        v[j >> 3] |= x << sh;
        if (sh) v[(j >> 3) + 1] |= x >> (64 - sh);
        j += m; src += m; pos += m; seg -= m;
      }
    }
    store16(p0, v);
  }
}

__device__ __forceinline__ void load32(const uint8_t* __restrict__ p, int64_t pos, int64_t n, uint64_t v[4]) {
  if (pos + 32 <= n && (((uintptr_t)(p + pos)) & 15) == 0) {
    const uint4* q = reinterpret_cast<const uint4*>(p + pos);
    uint4 a = q[0], b = q[1];
    v[0] = (uint64_t)a.x | ((uint64_t)a.y << 32);
    v[1] = (uint64_t)a.z | ((uint64_t)a.w << 32);
    v[2] = (uint64_t)b.x | ((uint64_t)b.y << 32);
    v[3] = (uint64_t)b.z | ((uint64_t)b.w << 32);
  } else {
    v[0] = v[1] = v[2] = v[3] = 0;
    for (int j = 0; j < 32 && pos + j < n; ++j) v[j >> 3] |= (uint64_t)p[pos + j] << (8 * (j & 7));
  }
}

__device__ __forceinline__ void store32(uint8_t* __restrict__ p, int64_t pos, int64_t n, const uint64_t v[4]) {
  if (pos + 32 <= n && (((uintptr_t)(p + pos)) & 15) == 0) {
    uint4* q = reinterpret_cast<uint4*>(p + pos);
    q[0] = make_uint4((uint32_t)v[0], (uint32_t)(v[0] >> 32), (uint32_t)v[1], (uint32_t)(v[1] >> 32));
    q[1] = make_uint4((uint32_t)v[2], (uint32_t)(v[2] >> 32), (uint32_t)v[3], (uint32_t)(v[3] >> 32));
  } else {
    for (int j = 0; j < 32 && pos + j < n; ++j) p[pos + j] = (uint8_t)(v[j >> 3] >> (8 * (j & 7)));
  }
}

// mode 0: ASCII -> codes (+packed, validated); mode 1: codes -> packed
template <int MODE>
__global__ __launch_bounds__(BNPK_BLOCK) void flat_encode_kernel(const uint8_t* __restrict__ in, int64_t n,
                                                                 uint8_t* __restrict__ codes,
                                                                 uint64_t* __restrict__ packed,
                                                                 unsigned long long* __restrict__ err) {
  int64_t n_words = (n + 31) / 32;
  int64_t w = (int64_t)blockIdx.x * BNPK_BLOCK + threadIdx.x;
  if (w > n_words) return;
  if (w == n_words) { if (packed) packed[w] = 0; return; }
  int64_t pos = w * 32;
  uint64_t v[4], cw[4] = {0, 0, 0, 0};
  load32(in, pos, n, v);
  uint64_t word = 0;
  unsigned long long bad = (unsigned long long)BNPK_NONE;
  int cnt = (int)min((int64_t)32, n - pos);
  for (int j = 0; j < cnt; ++j) {
    uint32_t b = (uint32_t)(v[j >> 3] >> (8 * (j & 7))) & 0xff;
    uint32_t c;
    if (MODE == 0) {
      bool ok;
      c = dna_code(b, &ok);
      if (!ok) { c = 0; if ((unsigned long long)(pos + j) < bad) bad = (unsigned long long)(pos + j); }
      cw[j >> 3] |= (uint64_t)c << (8 * (j & 7));
    } else {
      c = b & 3u;
    }
    word |= (uint64_t)c << (2 * j);
  }
  if (MODE == 0) {
    if (bad != (unsigned long long)BNPK_NONE) atomicMin(err, bad);
    if (codes) store32(codes, pos, n, cw);
  }
  if (packed) packed[w] = word;
}

__global__ __launch_bounds__(BNPK_BLOCK) void unpack_kernel(const uint64_t* __restrict__ packed, int64_t n,
                                                            int to_ascii, uint8_t* __restrict__ out) {
  int64_t w = (int64_t)blockIdx.x * BNPK_BLOCK + threadIdx.x;
  int64_t pos = w * 32;
  if (pos >= n) return;
  uint64_t word = packed[w];
  uint64_t v[4] = {0, 0, 0, 0};
  const uint32_t alphabet = 0x54474341u;   // 'A','C','G','T' little-endian
  for (int j = 0; j < 32; ++j) {
    uint32_t c = (uint32_t)(word >> (2 * j)) & 3u;
    if (to_ascii) c = (alphabet >> (8 * c)) & 0xff;
    v[j >> 3] |= (uint64_t)c << (8 * (j & 7));
  }
  store32(out, pos, n, v);
}

__global__ void take_bytes_kernel(const uint8_t* __restrict__ buf, const int64_t* __restrict__ pos, int64_t m,
                                  int64_t delta, uint8_t* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < m; i += stride) out[i] = buf[pos[i] + delta];
}


// A7 for any alphabet: out[i] = lut[in[i]] (AlphabetEncoding._encode, encodings/alphabet_encoding.py:37-46).  The
// 256-entry table sits in LDS; a lane translates 16 bytes (one aligned 16-byte load, four table look-ups per dword,
// one 16-byte store).  A byte whose table entry is 255 is invalid: the smallest such offset is kept in *err.
__global__ __launch_bounds__(BNPK_BLOCK) void lut_bytes_kernel(const uint8_t* __restrict__ in, int64_t n,
                                                               const uint8_t* __restrict__ lut_dev,
                                                               uint8_t* __restrict__ out,
                                                               unsigned long long* __restrict__ err) {
  __shared__ uint8_t lut[256];
  lut[threadIdx.x] = lut_dev[threadIdx.x];
  __syncthreads();
  const int64_t n16 = n >> 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned long long bad = (unsigned long long)BNPK_NONE;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
    const uint4 v = reinterpret_cast<const uint4*>(in)[i];
    uint32_t w[4] = {v.x, v.y, v.z, v.w}, r[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      r[q] = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const uint32_t c = lut[(w[q] >> (8 * b)) & 0xffu];
        if (c == 255u) bad = min(bad, (unsigned long long)(16 * i + 4 * q + b));
        r[q] |= c << (8 * b);
      }
    }
    reinterpret_cast<uint4*>(out)[i] = make_uint4(r[0], r[1], r[2], r[3]);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 15)) {                    // tail bytes
    const int64_t i = (n16 << 4) + threadIdx.x;
    const uint8_t c = lut[in[i]];
    if (c == 255) bad = min(bad, (unsigned long long)i);
    out[i] = c;
  }
  if (bad != (unsigned long long)BNPK_NONE) atomicMin(err, bad);
}

}  // namespace

extern "C" {

int bnpk_take_bytes(bnpk_ctx* ctx, const uint8_t* d_buf, const int64_t* d_pos, int64_t m, int64_t delta,
                    uint8_t* d_out, void* stream) {
  if (!ctx || m < 0) return BNPK_ERR_ARG;
  if (m == 0) return BNPK_OK;
  if (!d_buf || !d_pos || !d_out) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  bnpk_timer t(ctx, "take_bytes", s);
  hipLaunchKernelGGL(take_bytes_kernel, dim3(grid_for(ceil_div(m, 256))), dim3(256), 0, s, d_buf, d_pos, m, delta, d_out);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_packed_rows_slice(bnpk_ctx* ctx, const uint64_t* d_packed, int64_t n_bases_in, const int64_t* d_offsets, int64_t first_row,
                           int64_t n_rows, int64_t first_base, int64_t n_bases, uint64_t* d_out_packed, int64_t* d_out_offsets,
                           void* stream) {
  if (!ctx || n_bases_in < 0 || first_row < 0 || n_rows < 0 || first_base < 0 || n_bases < 0 || first_base + n_bases > n_bases_in ||
      !d_packed || !d_offsets || !d_out_packed || !d_out_offsets)
    return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  bnpk_timer t(ctx, "packed_rows_slice", s);
  const int64_t work = std::max<int64_t>(n_bases / 32 + 2, n_rows + 1);
  hipLaunchKernelGGL(packed_rows_slice_kernel, dim3(grid_for(std::min<int64_t>(ceil_div(work, BNPK_BLOCK), 4096))), dim3(BNPK_BLOCK), 0, s,
                     d_packed, n_bases_in / 32 + 2, d_offsets, first_row, n_rows, first_base, n_bases, d_out_packed, d_out_offsets);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_gather_encode_dna(bnpk_ctx* ctx, const uint8_t* d_buf, int64_t buf_size, const int64_t* d_starts,
                           const int64_t* d_offsets, int64_t n_rows, int64_t total, uint8_t* d_codes,
                           uint64_t* d_packed, uint64_t* d_row_ends, int64_t* d_err_offset, void* stream) {
  if (!ctx || n_rows < 0 || total < 0 || buf_size < 0 || !d_err_offset) return BNPK_ERR_ARG;
  if (!d_codes && !d_packed) return BNPK_ERR_ARG;
  if (total > 0 && (!d_buf || !d_starts || !d_offsets || n_rows == 0)) return BNPK_ERR_ARG;
  if (d_codes && ((uintptr_t)d_codes & 15)) return BNPK_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  int64_t n_words = (total + 31) / 32;
  constexpr int64_t tile_words = (int64_t)BNPK_BLOCK * GE_WPL;
  int64_t blocks = ceil_div(total / 32 + 2, tile_words);
  if (blocks > BNPK_MAX_BLOCKS) return BNPK_ERR_RANGE;
  auto* err = reinterpret_cast<unsigned long long*>(d_err_offset);
  // tiles of 512 packed words (16384 bases); the launch covers the pad word(s) too
  const int64_t n_tiles = ceil_div(n_words, tile_words);
  void* table = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, tile_rows_bytes(n_tiles), &table, (hipStream_t)stream));
  bnpk_timer t(ctx, "gather_encode_dna", s);
  unsigned* ends32 = reinterpret_cast<unsigned*>(d_row_ends);
  if (d_row_ends)                                            // (its last two words: the kernel covers total/32 + 2 halves of them)
    BNPK_HIP(ctx, hipMemsetAsync(d_row_ends + total / 64, 0, 16, s));
  if (total > 0)
    BNPK_CHECK(build_tile_rows(ctx, d_offsets, n_rows, tile_words * 32, (int64_t*)table, s));
  const int64_t* tr = (const int64_t*)table;
  dim3 g((unsigned)blocks), b(BNPK_BLOCK);
  if (d_codes && d_packed)
    hipLaunchKernelGGL((gather_encode_kernel<true, true>), g, b, 0, s, d_buf, buf_size, d_starts, d_offsets, n_rows,
                       total, tr, n_tiles, d_codes, d_packed, ends32, err);
  else if (d_packed)
    hipLaunchKernelGGL((gather_encode_kernel<false, true>), g, b, 0, s, d_buf, buf_size, d_starts, d_offsets, n_rows,
                       total, tr, n_tiles, d_codes, d_packed, ends32, err);
  else
    hipLaunchKernelGGL((gather_encode_kernel<true, false>), g, b, 0, s, d_buf, buf_size, d_starts, d_offsets, n_rows,
                       total, tr, n_tiles, d_codes, d_packed, ends32, err);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_gather_rows(bnpk_ctx* ctx, const uint8_t* d_buf, const int64_t* d_starts, const int64_t* d_offsets,
                     int64_t n_rows, int64_t total, int subtract, uint8_t* d_out, void* stream) {
  if (!ctx || n_rows < 0 || total < 0) return BNPK_ERR_ARG;
  if (total == 0) return BNPK_OK;
  if (!d_buf || !d_starts || !d_offsets || !d_out || n_rows == 0) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  int64_t blocks = ceil_div(total, GR_TILE);
  if (blocks > BNPK_MAX_BLOCKS) return BNPK_ERR_RANGE;
  void* table = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, tile_rows_bytes(blocks), &table, (hipStream_t)stream));
  bnpk_timer t(ctx, "gather_rows", s);
  BNPK_CHECK(build_tile_rows(ctx, d_offsets, n_rows, GR_TILE, (int64_t*)table, s));
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)blocks), dim3(BNPK_BLOCK), 0, s, d_buf, d_starts, d_offsets,
                     n_rows, total, subtract, (const int64_t*)table, d_out);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_encode_dna_flat(bnpk_ctx* ctx, const uint8_t* d_ascii, int64_t n, uint8_t* d_codes, uint64_t* d_packed,
                         int64_t* d_err_offset, void* stream) {
  if (!ctx || n < 0 || !d_err_offset || (!d_codes && !d_packed) || (n > 0 && !d_ascii)) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  int64_t blocks = ceil_div((n + 31) / 32 + 1, BNPK_BLOCK);
  if (blocks > BNPK_MAX_BLOCKS) return BNPK_ERR_RANGE;
  bnpk_timer t(ctx, "encode_dna_flat", s);
  hipLaunchKernelGGL((flat_encode_kernel<0>), dim3((unsigned)blocks), dim3(BNPK_BLOCK), 0, s, d_ascii, n, d_codes,
                     d_packed, reinterpret_cast<unsigned long long*>(d_err_offset));
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_pack_codes(bnpk_ctx* ctx, const uint8_t* d_codes, int64_t n, uint64_t* d_packed, void* stream) {
  if (!ctx || n < 0 || !d_packed || (n > 0 && !d_codes)) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  int64_t blocks = ceil_div((n + 31) / 32 + 1, BNPK_BLOCK);
  if (blocks > BNPK_MAX_BLOCKS) return BNPK_ERR_RANGE;
  bnpk_timer t(ctx, "pack_codes", s);
  hipLaunchKernelGGL((flat_encode_kernel<1>), dim3((unsigned)blocks), dim3(BNPK_BLOCK), 0, s, d_codes, n,
                     (uint8_t*)nullptr, d_packed, (unsigned long long*)nullptr);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_unpack_codes(bnpk_ctx* ctx, const uint64_t* d_packed, int64_t n, int to_ascii, uint8_t* d_out,
                      void* stream) {
  if (!ctx || n < 0) return BNPK_ERR_ARG;
  if (n == 0) return BNPK_OK;
  if (!d_packed || !d_out) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  int64_t blocks = ceil_div((n + 31) / 32, BNPK_BLOCK);
  if (blocks > BNPK_MAX_BLOCKS) return BNPK_ERR_RANGE;
  bnpk_timer t(ctx, "unpack_codes", s);
  hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)blocks), dim3(BNPK_BLOCK), 0, s, d_packed, n, to_ascii, d_out);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_lut_bytes(bnpk_ctx* ctx, const uint8_t* d_in, int64_t n, const uint8_t* h_lut256, uint8_t* d_out,
                   int64_t* d_err_offset, void* stream) {
  if (!ctx || n < 0 || !h_lut256 || !d_err_offset) return BNPK_ERR_ARG;
  if (n == 0) return BNPK_OK;
  if (!d_in || !d_out) return BNPK_ERR_ARG;
  if (((uintptr_t)d_in & 15) || ((uintptr_t)d_out & 15)) return BNPK_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  void* lut = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, 256, &lut, (hipStream_t)stream));
  BNPK_HIP(ctx, hipMemcpyAsync(lut, h_lut256, 256, hipMemcpyHostToDevice, s));
  BNPK_HIP(ctx, hipStreamSynchronize(s));              // (the table is the caller's host memory)
  bnpk_timer t(ctx, "lut_bytes", s);
  const int64_t blocks = std::min<int64_t>(std::max<int64_t>(ceil_div(n >> 4, BNPK_BLOCK), 1), (int64_t)ctx->compute_units * 16);
  hipLaunchKernelGGL(lut_bytes_kernel, dim3((unsigned)blocks), dim3(BNPK_BLOCK), 0, s, d_in, n, (const uint8_t*)lut, d_out,
                     reinterpret_cast<unsigned long long*>(d_err_offset));
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

}  // extern "C"
