// Reverse complement of ragged DNA (bionumpy/sequence/dna.py:36-65) and canonical k-mer hashes, gfx950.
//
//   get_reverse_complement(seq) = complement(seq)[..., ::-1]: every row is reversed and every base replaced by its
//   complement.  On the 2-bit form (A C G T = 0 1 2 3) the complement of a code is 3 - code = ~code & 3, so a run
//   of up to 32 bases is reverse-complemented with one reversal of the 2-bit groups of a 64-bit word and one NOT.
//   The ASCII form goes through the reference's 128-entry table (A<->T, C<->G, N->N, everything else -> 0).
//
//   canonical(h) = min(h, rc(h)) for a k-mer hash in the reference's layout (first base in the least significant
//   2 bits): rc(h) is the hash of the reverse complement k-mer = the reversed 2-bit groups of h, complemented.
#include "common.h"
#include "rows.h"

namespace {

// (experiment knobs of scripts/exp/rc_repro.py — the library is built with the defaults)
#ifndef BNPK_RCP_UNROLL
#define BNPK_RCP_UNROLL 4                            // 1: the loop over a lane's words stays rolled; 4 (with the VGPR floor): unrolled (see rc_packed_kernel)
#endif
#ifndef BNPK_RCP_LDS_PAD
#define BNPK_RCP_LDS_PAD 0                           // extra LDS bytes per workgroup (limits the workgroups per CU)
#endif
#ifndef BNPK_RCP_NO_LDS
#define BNPK_RCP_NO_LDS 0                            // 1: the row offsets are always read from global memory
#endif
#ifndef BNPK_RCP_FENCE
#define BNPK_RCP_FENCE 0                             // 1: a scheduling + memory fence behind every word of a lane
#endif
#define BNPK_PRAGMA_(x) _Pragma(#x)
#define BNPK_PRAGMA_UNROLL(n) BNPK_PRAGMA_(unroll n)

constexpr int RCP_WPL = 4;                           // output words (32 bases each) per lane of rc_packed
constexpr int RC_TILE_WORDS = BNPK_BLOCK * RCP_WPL;
constexpr int64_t RC_TILE_BASES = (int64_t)RC_TILE_WORDS * 32;
constexpr int RC_BYTES_PER_LANE = 16;
constexpr int64_t RC_TILE_BYTES = (int64_t)BNPK_BLOCK * RC_BYTES_PER_LANE;

// bases [pos, pos + n) of the packed stream, n <= 32, in the low 2n bits
__device__ __forceinline__ uint64_t packed_run(const uint64_t* __restrict__ w, int64_t pos, int n) {
  const int64_t i = pos >> 5;
  const int sh = 2 * (int)(pos & 31);
  uint64_t v = w[i] >> sh;
  if (sh && sh + 2 * n > 64) v |= w[i + 1] << (64 - sh);
  return n >= 32 ? v : (v & ((1ull << (2 * n)) - 1ull));
}

// last row r in [lo, hi] with off[r] <= p
__device__ __forceinline__ int64_t row_of(const int64_t* __restrict__ off, int64_t lo, int64_t hi, int64_t p) {
  while (lo < hi) {
    const int64_t mid = lo + ((hi - lo + 1) >> 1);
    if (off[mid] <= p) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__global__ __launch_bounds__(BNPK_BLOCK) void rc_packed_kernel(const uint64_t* __restrict__ in,
                                                               const int64_t* __restrict__ off, int64_t n_rows,
                                                               int64_t total, const int64_t* __restrict__ tile_rows,
                                                               int64_t n_tiles, uint64_t* __restrict__ out) {
  constexpr int RCP_LDS_ROWS = 510;                          // rows of a tile whose offsets are staged (reads of 64 bases and more)
  constexpr int RCP_SRC_WORDS = 1280;                        // packed words of the tile's rows that are staged (40 960 bases: the tile and a read on either side)
  __shared__ int64_t srow[RCP_LDS_ROWS + 2];
  __shared__ uint64_t ssrc[RCP_SRC_WORDS + 2];
  __shared__ int srow32[RCP_LDS_ROWS + 2];                   // the same offsets relative to the first staged word: 32-bit arithmetic from here on
#if !defined(BNPK_RCP_NO_FLOOR)
  BNPK_VGPR_FLOOR_32();                                      // the unrolled form of the loop below would get 24 VGPRs: see there
#endif
#if BNPK_RCP_LDS_PAD
  __shared__ int pad_words[BNPK_RCP_LDS_PAD / 4];
  if (total < 0) pad_words[threadIdx.x] = 1;                 // (never true: keeps the array)
#endif
  const int64_t w0 = (int64_t)blockIdx.x * RC_TILE_WORDS;
  if (w0 * 32 >= total) {                                    // (uniform) a tile of pad words only
    for (int64_t w = w0 + threadIdx.x; w < w0 + RC_TILE_WORDS; w += BNPK_BLOCK)
      if (w * 32 < total + 64) out[w] = 0;                   // the pad words of the packed layout
    return;
  }
  // (the table has an entry for every tile that starts inside the data; the pad words form tiles of their own)
  const int64_t lo = tile_rows[blockIdx.x];
  const int64_t hi = ((int64_t)(blockIdx.x + 1) * RC_TILE_BASES < total) ? tile_rows[blockIdx.x + 1] : n_rows - 1;
  // The offsets of the tile's rows (~220 reads of 150 bases) come into LDS with coalesced loads; a lane finds its rows and
  // the ends of the rows it crosses there, for its four words.  Searched in global memory and one word per lane, as this
  // kernel did, a lane sat through a chain of eight dependent loads before its first packed word: 3.4 ms per 50 M reads
  // for 3.8 GB of traffic.
  const bool staged = !BNPK_RCP_NO_LDS && hi - lo + 2 <= RCP_LDS_ROWS + 2;        // (uniform) offsets lo .. hi + 1
  if (staged) {
    for (int64_t i = threadIdx.x; i <= hi - lo + 1; i += BNPK_BLOCK) srow[i] = off[lo + i];
    __syncthreads();
  }
  auto offset_of = [&](int64_t row) { return (staged && row <= hi + 1) ? srow[row - lo] : off[row]; };
  // The bases the tile's words are made of are the tile's own rows, i.e. (for reads) the tile's own stretch of the packed
  // stream and a read on either side: they come into LDS with coalesced loads and the words are cut out of LDS (two reads per
  // piece) — gathered from global memory, two dependent 8-byte loads per piece at the mirror position, this kernel ran at 0.26
  // of the peak whatever was done to its loop.  Rows much longer than a tile (a chromosome) keep the global form.
  int64_t src_word0 = 0;
  bool src_staged = false;                                   // (uniform)
  const int64_t n_in_words = total / 32 + 2;                 // (the packed layout of bnpk_gather_encode_dna)
  if (staged) {
    const int64_t first = srow[0] >> 5, last = (srow[hi - lo + 1] + 31) >> 5;       // words [first, last) hold the rows lo .. hi
    if (last - first <= RCP_SRC_WORDS) {
      src_staged = true;
      src_word0 = first;
      for (int64_t i = threadIdx.x; i < last - first + 2; i += BNPK_BLOCK) ssrc[i] = first + i < n_in_words ? in[first + i] : 0ull;
      __syncthreads();
    }
  }
  // The allocation is pinned at 32 VGPRs above.  Round 4 saw the unrolled form of the loop below write garbled rows "from the 257th
  // workgroup on"; round 5 ran it down (NOTES.md "rc_packed: the cause"): the unrolled ISA is right (every load waited for, live
  // ranges read by hand, s_nop padding / forced waits / no LDS / fences change nothing) — what differs is that only the fully
  // unrolled kernel needed exactly 24 VGPRs, and with a 24-register allocation every workgroup that is not the first on its CU
  // computes garbage, nondeterministically (a CU mask of 128 / 64 / 32 / 8 CUs moves the first bad tile to 129 / 72 / 37 / 10).
  // The SAME code with the allocation bumped to 32 registers is right in every run; the build's ISA lint refuses a kernel that
  // comes out at 24.  With the floor in place the loop is unrolled again (BNPK_RCP_UNROLL).
  // reverse complement of the n <= 32 bases that END the source run [from, from + n): the low 2n bits
  auto rc_piece = [&](int64_t from, int n) -> uint64_t {
    uint64_t src;
    if (src_staged) {
      const int i = (int)((from >> 5) - src_word0), sh = 2 * (int)(from & 31);
      src = ssrc[i] >> sh;
      if (sh && sh + 2 * n > 64) src |= ssrc[i + 1] << (64 - sh);
      if (n < 32) src &= (1ull << (2 * n)) - 1ull;
    } else {
      src = packed_run(in, from, n);
    }
    const uint64_t rc = ~(reverse_2bit_groups(src) >> (64 - 2 * n));
    return n >= 32 ? rc : (rc & ((1ull << (2 * n)) - 1ull));
  };
  // any number of pieces (rows shorter than a word, empty rows): the walk
  auto walk = [&](int64_t r, int64_t p0, int64_t p1) -> uint64_t {
    uint64_t word = 0;
    int64_t p = p0;
    while (p < p1) {
      int64_t s = offset_of(r), e = offset_of(r + 1);
      while (e <= p) { ++r; s = e; e = offset_of(r + 1); }   // empty rows, and the step to the next row
      const int64_t stop = min(e, p1);
      // output positions [p, stop) of row [s, e) <- source positions s + e - 1 - p down to s + e - stop
      word |= rc_piece(s + e - stop, (int)(stop - p)) << (2 * (int)(p - p0));
      p = stop;
    }
    return word;
  };
  // A word of 32 output bases lies in one row or — reads are longer than a word — in the end of one and the start of the
  // next: both pieces are fetched without waiting for each other, for all of the lane's words at once (the walk above, a loop
  // with loads inside, was a chain of dependent round trips per word: 0.25 of the peak whether rolled or unrolled).  Words
  // that need a third piece take the walk.
  if (src_staged) {
    // everything the tile needs is in LDS and within 2^17 bases of the first staged word: positions as 32-bit offsets from it
    // (the 64-bit form below spends half of its instructions on carries; this kernel is bound by its instructions)
    const int64_t base = src_word0 << 5;
    const int nr = (int)(hi - lo);
    for (int i = threadIdx.x; i <= nr + 1; i += BNPK_BLOCK) srow32[i] = (int)(srow[i] - base);
    __syncthreads();
    auto piece32 = [&](int from, int n) -> uint64_t {
      const int i = from >> 5, sh = 2 * (from & 31);
      uint64_t src = ssrc[i] >> sh;
      if (sh && sh + 2 * n > 64) src |= ssrc[i + 1] << (64 - sh);
      const uint64_t rc = ~(reverse_2bit_groups(src) >> (64 - 2 * n));      // (bits of src above 2n fall out of the shift)
      return n >= 32 ? rc : (rc & ((1ull << (2 * n)) - 1ull));
    };
BNPK_PRAGMA_UNROLL(BNPK_RCP_UNROLL)
    for (int it = 0; it < RCP_WPL; ++it) {
      const int64_t w = w0 + it * BNPK_BLOCK + threadIdx.x;
      const int64_t p0 = w * 32;
      if (p0 >= total) {
        if (p0 < total + 64) out[w] = 0;
        continue;
      }
      const int q0 = (int)(p0 - base), q1 = (int)(min(p0 + 32, total) - base);
      int a = 0, b = nr;
      while (a < b) {
        const int mid = a + ((b - a + 1) >> 1);
        if (srow32[mid] <= q0) a = mid; else b = mid - 1;
      }
      const int s0 = srow32[a], e0 = srow32[a + 1], e1 = srow32[min(a + 2, nr + 1)];
      const int stop_a = min(e0, q1), n_a = stop_a - q0;
      const bool two = stop_a < q1;
      const int stop_b = min(e1, q1), n_b = stop_b - stop_a;
      uint64_t word;
      if (n_a <= 0 || (two && (n_b <= 0 || stop_b < q1))) {  // an empty row, or a third piece: the walk
        word = walk(lo + a, p0, min(p0 + 32, total));
      } else {
        word = piece32(s0 + e0 - stop_a, n_a);
        if (two) word |= piece32(e0 + e1 - stop_b, n_b) << (2 * n_a);
      }
      out[w] = word;
    }
    return;
  }
BNPK_PRAGMA_UNROLL(BNPK_RCP_UNROLL)
  for (int it = 0; it < RCP_WPL; ++it) {
    const int64_t w = w0 + it * BNPK_BLOCK + threadIdx.x;
    const int64_t p0 = w * 32;
    if (p0 >= total) {
      if (p0 < total + 64) out[w] = 0;                       // the pad words behind the last tile's data
      continue;
    }
    const int64_t p1 = min(p0 + 32, total);
    int64_t r;
    if (staged) {
      int a = 0, b = (int)(hi - lo);
      while (a < b) {
        const int mid = a + ((b - a + 1) >> 1);
        if (srow[mid] <= p0) a = mid; else b = mid - 1;
      }
      r = lo + a;
    } else {
      r = row_of(off, lo, hi, p0);
    }
    const int64_t s0 = offset_of(r), e0 = offset_of(r + 1);
    uint64_t word;
    if (e0 <= p0) {                                          // (cannot happen for p0 < total: the search returns the last row that starts at or before p0)
      word = walk(r, p0, p1);
    } else {
      const int64_t stop_a = min(e0, p1);
      const int n_a = (int)(stop_a - p0);
      const bool two = stop_a < p1;
      const int64_t e1 = two ? offset_of(min(r + 2, n_rows)) : e0;        // the row behind: [e0, e1)
      const int64_t stop_b = min(e1, p1);
      const int n_b = (int)(stop_b - stop_a);
      if (two && (n_b <= 0 || stop_b < p1)) {                // an empty row, or a third piece
        word = walk(r, p0, p1);
      } else {
        word = rc_piece(s0 + e0 - stop_a, n_a);
        if (two) word |= rc_piece(e0 + e1 - stop_b, n_b) << (2 * n_a);
      }
    }
    out[w] = word;
#if BNPK_RCP_FENCE
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
#endif
  }
}

// complement table of the reference (bionumpy/sequence/dna.py:10,29-33), as a function
__device__ __forceinline__ uint32_t ascii_complement(uint32_t b) {
  return b == 'A' ? 'T' : b == 'T' ? 'A' : b == 'C' ? 'G' : b == 'G' ? 'C' : b == 'N' ? 'N' : 0u;
}

// The table on four bytes in a few instructions: the letters A C G T N differ in their low three bits (1 3 7 4 6), so
// V_PERM_B32 with the selector x & 7 looks both the letter that OUGHT to stand there and its complement up in 8-byte
// tables; a byte that is not the letter its low bits promise becomes 0 (rare: fixed up behind a branch).  The SWAR form
// this replaces — five exact byte compares with a 64-bit multiply each — was ~110 instructions per eight bytes, run twice
// per wavefront (lanes inside a row, lanes over a row boundary): what bounded the kernel, 6.2 ms per 50 M reads.
__device__ __forceinline__ uint32_t ascii_complement4(uint32_t x) {
  const uint32_t idx = x & 0x07070707u;
  const uint32_t expect = __builtin_amdgcn_perm(0x474E0054u, 0x43004100u, idx);     // . A . C | T . N G
  uint32_t c = __builtin_amdgcn_perm(0x434E0041u, 0x47005400u, idx);                // . T . G | A . N C
  const uint32_t z = expect ^ x;
  if (z) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if ((z >> (8 * j)) & 0xffu) c &= ~(0xffu << (8 * j));
  }
  return c;
}

// 16 output bytes per lane, every lane the same way: out[p0 + i] = complement(in[s + e - 1 - p0 - i]) while the row lasts,
// i.e. the 16 bytes that END at in[s + k - 1] (k = e - p0 bytes of the row left), reversed.  A lane with k < 16 — one in
// ten for reads of 150 bases — loads the 16 bytes that end k bytes BEHIND the next row's end as well, reverses them too
// and takes byte i from the first for i < k: one select per dword, then ONE complement for everybody.  Lanes over rows
// shorter than that, or whose loads would leave the buffer, walk byte by byte as before.
// in_starts (optional): the input rows were never gathered — row r lies at in[in_starts[r] ..), in a buffer of in_size bytes;
// the output is compact either way (off).
__global__ __launch_bounds__(BNPK_BLOCK) void rc_bytes_kernel(const uint8_t* __restrict__ in, int64_t in_size,
                                                              const int64_t* __restrict__ in_starts,
                                                              const int64_t* __restrict__ off, int64_t n_rows,
                                                              int64_t total, const int64_t* __restrict__ tile_rows,
                                                              int64_t n_tiles, uint8_t* __restrict__ out) {
  // The rows of the tile (4 KB of output: ~27 reads of 150 bases) are few: their offsets are loaded ONCE, by one coalesced
  // load, into LDS, and every lane finds its row there — a binary search over global memory was five dependent loads per
  // lane in front of the first byte of data.  Tiles of many tiny rows keep the search over global memory.
  constexpr int RC_LDS_ROWS = 254;
  __shared__ int64_t srow[RC_LDS_ROWS + 2];
  __shared__ int64_t sstart[RC_LDS_ROWS + 2];
  const int64_t p0 = ((int64_t)blockIdx.x * BNPK_BLOCK + threadIdx.x) * RC_BYTES_PER_LANE;
  int64_t lo, hi;
  tile_row_range(tile_rows, blockIdx.x, n_tiles, n_rows, lo, hi);
  const bool staged = hi - lo + 2 <= RC_LDS_ROWS + 2;          // (uniform) offsets lo .. hi + 1
  if (staged) {
    if ((int64_t)threadIdx.x <= hi - lo + 1) {
      srow[threadIdx.x] = off[lo + threadIdx.x];
      if (in_starts && lo + threadIdx.x < n_rows) sstart[threadIdx.x] = in_starts[lo + threadIdx.x];
    }
    __syncthreads();
  }
  auto row_in = [&](int64_t row, int64_t row_off) {            // where row `row` (output offset row_off) begins in `in`
    return !in_starts ? row_off : (staged && row <= hi + 1) ? sstart[row - lo] : in_starts[row];
  };
  if (p0 >= total) return;
  int64_t r;
  if (staged) {
    int a = 0, b = (int)(hi - lo);
    while (a < b) {
      const int mid = a + ((b - a + 1) >> 1);
      if (srow[mid] <= p0) a = mid; else b = mid - 1;
    }
    r = lo + a;
  } else {
    r = row_of(off, lo, hi, p0);
  }
  const int64_t p1 = min(p0 + RC_BYTES_PER_LANE, total);
  int64_t s = staged ? srow[r - lo] : off[r], e = staged ? srow[r - lo + 1] : off[r + 1];
  uint32_t w[4] = {0, 0, 0, 0};                               // the lane's sixteen output bytes, stored once
  const int64_t k = e - p0;                                   // bytes of row r from p0 on (>= 1)
  int64_t is = row_in(r, s);                                  // (== s for back-to-back rows)
  const int64_t from = is + k - RC_BYTES_PER_LANE;            // the 16 bytes that end with the row's byte for p0
  bool done = false;
  if (p1 - p0 == RC_BYTES_PER_LANE && from >= 0 && from + RC_BYTES_PER_LANE <= in_size) {
    uint32_t a[4];
    if (k >= RC_BYTES_PER_LANE) {
      __builtin_memcpy(a, in + from, 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) w[q] = __builtin_amdgcn_perm(0u, a[3 - q], 0x00010203u);       // the 16 bytes reversed
      done = true;
    } else {
      const int64_t e2 = r + 2 <= n_rows ? ((staged && r + 2 <= hi + 1) ? srow[r + 2 - lo] : off[r + 2]) : e;
      const int64_t ie2 = e2 > e ? row_in(r + 1, e) + (e2 - e) : 0;     // where row r + 1 ends in `in`
      if (e2 - e >= RC_BYTES_PER_LANE - k && ie2 + k <= in_size && ie2 + k >= RC_BYTES_PER_LANE) {   // one boundary: the rest lies in row r + 1
        uint32_t b[4];
        __builtin_memcpy(a, in + from, 16);
        __builtin_memcpy(b, in + (ie2 + k - RC_BYTES_PER_LANE), 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t ra = __builtin_amdgcn_perm(0u, a[3 - q], 0x00010203u), rb = __builtin_amdgcn_perm(0u, b[3 - q], 0x00010203u);
          const int nb = min(max((int)k - 4 * q, 0), 4);      // bytes of this dword that come from row r
          const uint32_t m = nb == 4 ? ~0u : (1u << (8 * nb)) - 1u;
          w[q] = (ra & m) | (rb & ~m);
        }
        done = true;
      }
    }
  }
  if (done) {
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] = ascii_complement4(w[q]);
  } else {
    for (int64_t p = p0; p < p1; ++p) {
      while (e <= p) { ++r; s = e; e = off[r + 1]; is = in_starts ? in_starts[r] : s; }
      w[(p - p0) >> 2] |= ascii_complement(in[is + (e - 1 - p)]) << (8 * (int)((p - p0) & 3));
    }
  }
  if (p1 - p0 == RC_BYTES_PER_LANE) {                         // (p0 is a multiple of 16, the buffer 16-byte aligned)
    *reinterpret_cast<uint4*>(out + p0) = make_uint4(w[0], w[1], w[2], w[3]);
  } else {
    for (int j = 0; j < (int)(p1 - p0); ++j) out[p0 + j] = (uint8_t)(w[j >> 2] >> (8 * (j & 3)));
  }
}

__global__ __launch_bounds__(BNPK_BLOCK) void canonical_kernel(int64_t* __restrict__ h, int64_t n, int k) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const uint64_t mask = (1ull << (2 * k)) - 1ull;
  for (; i < n; i += stride) {
    const uint64_t x = (uint64_t)h[i];
    const uint64_t rc = ~(reverse_2bit_groups(x) >> (64 - 2 * k)) & mask;
    h[i] = (int64_t)min(x, rc);
  }
}

}  // namespace

extern "C" {

int bnpk_reverse_complement_packed(bnpk_ctx* ctx, const uint64_t* d_packed, const int64_t* d_offsets, int64_t n_rows,
                                   int64_t total, uint64_t* d_out, void* stream) {
  if (!ctx || n_rows < 0 || total < 0 || !d_out || !d_offsets || (total > 0 && !d_packed) || d_out == d_packed)
    return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int64_t n_words = total / 32 + 2;                    // the packed layout of bnpk_gather_encode_dna
  const int64_t n_tiles = ceil_div(n_words, RC_TILE_WORDS);
  if (n_tiles > BNPK_MAX_BLOCKS) return BNPK_ERR_RANGE;
  void* table = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, tile_rows_bytes(n_tiles), &table, (hipStream_t)stream));
  bnpk_timer t(ctx, "reverse_complement_packed", s);
  if (n_rows > 0 && total > 0) BNPK_CHECK(build_tile_rows(ctx, d_offsets, n_rows, RC_TILE_BASES, (int64_t*)table, s));
  hipLaunchKernelGGL(rc_packed_kernel, dim3((unsigned)n_tiles), dim3(BNPK_BLOCK), 0, s, d_packed, d_offsets, n_rows, total,
                     (const int64_t*)table, n_tiles, d_out);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_reverse_complement_bytes(bnpk_ctx* ctx, const uint8_t* d_bytes, const int64_t* d_offsets, int64_t n_rows,
                                  int64_t total, uint8_t* d_out, void* stream) {
  return bnpk_reverse_complement_rows(ctx, d_bytes, total, nullptr, d_offsets, n_rows, total, d_out, stream);
}

int bnpk_reverse_complement_rows(bnpk_ctx* ctx, const uint8_t* d_bytes, int64_t in_size, const int64_t* d_in_starts,
                                 const int64_t* d_offsets, int64_t n_rows, int64_t total, uint8_t* d_out, void* stream) {
  if (!ctx || n_rows < 0 || total < 0 || in_size < 0 || !d_offsets || (total > 0 && (!d_bytes || !d_out)) ||
      (total > 0 && d_out == d_bytes))
    return BNPK_ERR_ARG;
  if (total == 0 || n_rows == 0) return BNPK_OK;
  hipStream_t s = (hipStream_t)stream;
  const int64_t n_tiles = ceil_div(total, RC_TILE_BYTES);
  if (n_tiles > BNPK_MAX_BLOCKS) return BNPK_ERR_RANGE;
  void* table = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, tile_rows_bytes(n_tiles), &table, (hipStream_t)stream));
  bnpk_timer t(ctx, "reverse_complement_bytes", s);
  BNPK_CHECK(build_tile_rows(ctx, d_offsets, n_rows, RC_TILE_BYTES, (int64_t*)table, s));
  hipLaunchKernelGGL(rc_bytes_kernel, dim3((unsigned)n_tiles), dim3(BNPK_BLOCK), 0, s, d_bytes, in_size, d_in_starts, d_offsets,
                     n_rows, total, (const int64_t*)table, n_tiles, d_out);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_canonical_kmers(bnpk_ctx* ctx, int64_t* d_hashes, int64_t n, int k, void* stream) {
  if (!ctx || n < 0 || k < 1 || k > 31 || (n > 0 && !d_hashes)) return BNPK_ERR_ARG;
  if (n == 0) return BNPK_OK;
  hipStream_t s = (hipStream_t)stream;
  bnpk_timer t(ctx, "canonical_kmers", s);
  hipLaunchKernelGGL(canonical_kernel, dim3(grid_for(ceil_div(n, BNPK_BLOCK))), dim3(BNPK_BLOCK), 0, s, d_hashes, n, k);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

}  // extern "C"
