// Fused FASTQ / two-line-FASTA decode for the k-mer pipeline (A2-A7 in two reads of the text):
// newline scan, entry validation, sequence-line extraction and ASCII -> 2-bit packing without ever materialising
// the newline table, the field tables or the row offsets.
//
//   census  (reads the text once)  per 16 KiB tile: newline count + the number of payload bytes on the lines of
//           each phase (line index mod lines_per_entry, relative to the tile's first line); after a scan of the
//           newline counts every tile knows its absolute first line, picks "its" sequence-byte count and a second
//           scan gives every tile the flat base index its sequence bytes start at
//   encode  (reads the text once)  per tile: the sequence bytes 2-bit encoded into an LDS staging area aligned like the
//           global packed words, read ends marked in a bit mask, the first byte of header / '+' lines validated, both
//           written out (interior words with plain stores, the edge word shared with the neighbouring tile OR-ed)
//   starts  bit-parallel pass over the read-end mask: a k-mer starts at base i iff no read ends in [i, i+k-2]
//
// Census and encode each come as a general kernel (every lane classifies its own 16 bytes, whatever the line structure
// is) and a fast kernel (newline masks per byte, bookkeeping per line / per entry, dense lanes for the base runs,
// persistent workgroups with the next tile's text in flight) that takes all tiles but the odd ones and hands those to
// the general kernel through a list; see the comments at the kernels.  Same bits either way.
//
// Equivalent reference expressions: OneLineBuffer.from_raw_buffer + _validate (io/one_line_buffer.py:45-71,156-173,
// io/fastq_buffer.py:39-45), _get_buffer_extractor + get_field_by_number(1) (io/one_line_buffer.py:140-152,
// io/file_buffers.py:315-338), EncodedRaggedArray.ravel() + AlphabetEncoding._encode (encodings/alphabet_encoding.py:19-46),
// BitArray.pack (sequence/kmers.py:121) and the ragged trim [..., :-(k-1)] (sequence/kmers.py:100).
#include <algorithm>

#include "common.h"
#include "scan.h"

namespace {

constexpr int FQ_VEC = 16;                                   // bytes per lane per load
constexpr int FQ_WAVE_BYTES = BNPK_WAVE * FQ_VEC;            // 1 KiB per wavefront-instruction
constexpr int FQ_ITERS = 4;                                  // (2 = 8 KiB tiles was measured: 1.6x slower, the per-tile costs double)
constexpr int FQ_WAVES = BNPK_BLOCK / BNPK_WAVE;
constexpr int FQ_TILE = FQ_WAVES * FQ_ITERS * FQ_WAVE_BYTES; // 16 KiB per workgroup
constexpr int FQ_MAXLPE = 4;
constexpr int FQ_TREC = 1 + FQ_MAXLPE;                       // per-tile census record: first line, payload bytes per phase
constexpr uint8_t FQ_NL = 10, FQ_CR = 13;

// (nomatch16: common.h)
__device__ __forceinline__ uint32_t fq_nomatch16(uint64_t lo, uint64_t hi, uint32_t rep) {
  return nomatch16((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32), rep);
}
__device__ __forceinline__ uint32_t fq_mask16(uint64_t lo, uint64_t hi, uint32_t rep) {   // bit j = byte j matches
  return ~fq_nomatch16(lo, hi, rep) & 0xffffu;
}

// one 16-byte chunk of the tile: raw bytes, newline mask, mask of the bytes that are not payload (newlines and,
// when CRs are stripped, the CR right before a newline), number of valid bytes
struct fq_chunk {
  uint64_t lo, hi;                                           // bytes 0-7, 8-15 (no arrays: runtime byte indices must not
  uint32_t nl, skip, valid;                                  // push the chunk into scratch memory)
};

__device__ __forceinline__ int64_t fq_chunk_pos(int64_t tile_base, int it) {
  return tile_base + (int64_t)wave_id() * (FQ_ITERS * FQ_WAVE_BYTES) + (int64_t)it * FQ_WAVE_BYTES + lane_id() * FQ_VEC;
}

__device__ __forceinline__ fq_chunk fq_load(const uint8_t* __restrict__ buf, int64_t pos, int64_t n, int strip_cr) {
  fq_chunk c;
  c.lo = c.hi = 0;
  c.nl = c.skip = 0;
  c.valid = 0;
  if (pos >= n) return c;
  if (pos + FQ_VEC <= n) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(buf + pos));   // streamed: 16 GB per batch, no reuse
    c.lo = (uint64_t)v.x | ((uint64_t)v.y << 32);
    c.hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
    c.valid = 0xffffu;
  } else {
    for (int j = 0; pos + j < n; ++j) {
      const uint64_t b = (uint64_t)buf[pos + j] << (8 * (j & 7));
      if (j < 8) c.lo |= b; else c.hi |= b;
      c.valid |= 1u << j;
    }
  }
  c.nl = fq_mask16(c.lo, c.hi, 0x01010101u * FQ_NL) & c.valid;
  c.skip = c.nl;
  if (strip_cr) {
    const uint32_t cr = fq_mask16(c.lo, c.hi, 0x01010101u * FQ_CR) & c.valid;
    uint32_t before_nl = cr & (c.nl >> 1);
    if ((cr >> 15) & 1u) {                                   // CR in the last byte: is the next byte a newline?
      if (pos + FQ_VEC < n && buf[pos + FQ_VEC] == FQ_NL) before_nl |= 1u << 15;
    }
    c.skip |= before_nl;
  }
  return c;
}

__device__ __forceinline__ uint32_t fq_byte(const fq_chunk& c, int j) {
  return (uint32_t)((j < 8 ? c.lo : c.hi) >> (8 * (j & 7))) & 0xffu;
}

// Exclusive prefix, in tile byte order (wave, iteration, lane), of one value per chunk; `smem` needs FQ_WAVES ints.
// Returns the tile total through *total.
__device__ __forceinline__ void fq_tile_prefix(const int v[FQ_ITERS], int ex[FQ_ITERS], int* smem, int* total) {
  int run = 0;
  int inc[FQ_ITERS];
#pragma unroll
  for (int it = 0; it < FQ_ITERS; ++it) {
    inc[it] = wave_inclusive_scan(v[it]);
    ex[it] = run + inc[it] - v[it];
    run += __builtin_amdgcn_readlane(inc[it], 63);
  }
  __syncthreads();                                           // smem may still be read from a previous call
  if (lane_id() == 0) smem[wave_id()] = run;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < FQ_WAVES; ++w) {
    const int x = smem[w];
    if (w < wave_id()) base += x;
    tot += x;
  }
#pragma unroll
  for (int it = 0; it < FQ_ITERS; ++it) ex[it] += base;
  *total = tot;
}

// The same prefix for counts of at most 16 per chunk (newlines): two chunks share a 32-bit scan, 16 bits each.
__device__ __forceinline__ void fq_tile_prefix16(const int v[FQ_ITERS], int ex[FQ_ITERS], int* smem, int* total) {
  static_assert(FQ_ITERS == 4 || FQ_ITERS == 2, "packed scans");
  const unsigned a = (unsigned)v[0] | ((unsigned)v[1] << 16);
  const unsigned ia = wave_inclusive_scan(a);
  const unsigned ta = (unsigned)__builtin_amdgcn_readlane((int)ia, 63);
  const int t0 = (int)(ta & 0xffffu), t1 = (int)(ta >> 16);
  int t2 = 0, t3 = 0;
  ex[0] = (int)(ia & 0xffffu) - v[0];
  ex[1] = t0 + (int)(ia >> 16) - v[1];
  if (FQ_ITERS == 4) {
    const unsigned b = (unsigned)v[FQ_ITERS - 2] | ((unsigned)v[FQ_ITERS - 1] << 16);
    const unsigned ib = wave_inclusive_scan(b);
    const unsigned tb = (unsigned)__builtin_amdgcn_readlane((int)ib, 63);
    t2 = (int)(tb & 0xffffu);
    t3 = (int)(tb >> 16);
    ex[FQ_ITERS - 2] = t0 + t1 + (int)(ib & 0xffffu) - v[FQ_ITERS - 2];
    ex[FQ_ITERS - 1] = t0 + t1 + t2 + (int)(ib >> 16) - v[FQ_ITERS - 1];
  }
  __syncthreads();                                           // smem may still be read from a previous call
  if (lane_id() == 0) smem[wave_id()] = t0 + t1 + t2 + t3;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < FQ_WAVES; ++w) {
    const int x = smem[w];
    if (w < wave_id()) base += x;
    tot += x;
  }
#pragma unroll
  for (int it = 0; it < FQ_ITERS; ++it) ex[it] += base;
  *total = tot;
}

// x mod m for 0 <= x < 2^20 and a small run-time m: a float reciprocal and a fix-up instead of the ~30-instruction
// integer division sequence (the line phases below are computed once per 16-byte chunk)
__device__ __forceinline__ int fq_mod_small(int x, int m, float inv_m) {
  int r = x - (int)((float)x * inv_m) * m;
  if (r < 0) r += m;
  if (r >= m) r -= m;
  return r;
}

// payload bytes of the chunk per line phase; `line` = tile-relative index of the chunk's first line
__device__ __forceinline__ void fq_count_phases(const fq_chunk& c, int line, int lpe, int cnt[FQ_MAXLPE]) {
  uint32_t nl = c.nl, todo = c.valid & ~c.skip;
  int ph = fq_mod_small(line, lpe, 1.0f / (float)lpe);
  while (true) {
    const uint32_t upto = nl ? ((1u << (__ffs(nl) - 1)) - 1u) : 0xffffu;     // bytes before the next newline
    const int m = __popc(todo & upto);
#pragma unroll
    for (int p = 0; p < FQ_MAXLPE; ++p) cnt[p] += (p == ph) ? m : 0;
    if (!nl) break;
    todo &= ~upto;
    nl &= nl - 1;
    ph = (ph + 1 == lpe) ? 0 : ph + 1;
  }
}

// ---- census --------------------------------------------------------------------------------------------------------
// The general census: LIST = false: tile = blockIdx.x; LIST = true: the tiles the fast census below handed back.
template <bool LIST>
__global__ __launch_bounds__(BNPK_BLOCK) void fq_census_kernel(const uint8_t* __restrict__ buf, int64_t n, int lpe,
                                                               const int64_t* __restrict__ flags,
                                                               int64_t* __restrict__ recs,
                                                               int64_t* __restrict__ newlines,
                                                               const unsigned* __restrict__ redo) {
  __shared__ int smem[FQ_WAVES];
  __shared__ int acc[FQ_MAXLPE];
  const int strip_cr = (int)flags[0];
  const unsigned n_list = LIST ? redo[0] : 1u;
  for (unsigned item = LIST ? blockIdx.x : 0u; item < n_list; item += LIST ? gridDim.x : 1u) {
    const int64_t tile = LIST ? (int64_t)redo[1 + item] : (int64_t)blockIdx.x;
    const int64_t tile_base = tile * FQ_TILE;
    if (LIST) __syncthreads();
    if (threadIdx.x < FQ_MAXLPE) acc[threadIdx.x] = 0;
    fq_chunk c[FQ_ITERS];
    int nls[FQ_ITERS], line[FQ_ITERS], total;
#pragma unroll
    for (int it = 0; it < FQ_ITERS; ++it) {
      c[it] = fq_load(buf, fq_chunk_pos(tile_base, it), n, strip_cr);
      nls[it] = __popc(c[it].nl);
    }
    fq_tile_prefix(nls, line, smem, &total);
    int cnt[FQ_MAXLPE] = {0, 0, 0, 0};
#pragma unroll
    for (int it = 0; it < FQ_ITERS; ++it) fq_count_phases(c[it], line[it], lpe, cnt);
#pragma unroll
    for (int p = 0; p < FQ_MAXLPE; ++p) {
      const int s = (int)wave_sum((unsigned)cnt[p]);
      if (lane_id() == 0 && s) atomicAdd(&acc[p], s);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int64_t* r = recs + tile * FQ_TREC;
      newlines[tile] = total;
#pragma unroll
      for (int p = 0; p < FQ_MAXLPE; ++p) r[1 + p] = acc[p];
    }
  }
}

// does one of the first lpe entries' header lines end in CR? (_modify_for_carriage_return, one_line_buffer.py:176-182)
// One wavefront walks the text 64 bytes at a time until it has seen (lpe-1)*lpe + 1 line ends.
__global__ void fq_detect_cr_kernel(const uint8_t* __restrict__ buf, int64_t n, int lpe, int64_t* __restrict__ flags) {
  const int lane = threadIdx.x;
  int64_t line = 0;
  const int64_t last_line = (int64_t)(lpe - 1) * lpe;
  bool found = false;
  uint32_t prev_last = 0;                                    // byte before the current 64-byte window
  for (int64_t base = 0; base < n && line <= last_line && !found; base += 64) {
    const int64_t i = base + lane;
    const uint32_t b = i < n ? buf[i] : 0;
    uint32_t prev = __shfl_up(b, 1, 64);
    if (lane == 0) prev = prev_last;
    uint64_t nlm = __ballot(b == FQ_NL);
    const uint64_t crm = __ballot(b == FQ_NL && prev == FQ_CR);
    while (nlm && line <= last_line) {
      const int j = __ffsll((long long)nlm) - 1;
      if (line % lpe == 0 && ((crm >> j) & 1ull)) found = true;
      nlm &= nlm - 1;
      ++line;
    }
    prev_last = __shfl(b, 63, 64);
  }
  if (lane == 0) flags[0] = found ? 1 : 0;
}

// after the scan of the newline counts: absolute first line of every tile, number of lines that take part
// (a multiple of lpe), the tile's sequence-byte count
__global__ void fq_select_kernel(const uint8_t* __restrict__ buf, int64_t n, int lpe, int seq_line, int64_t n_tiles,
                                 const int64_t* __restrict__ line_base, const int64_t* __restrict__ flags,
                                 int64_t* __restrict__ recs, int64_t* __restrict__ seq_count,
                                 int64_t* __restrict__ totals) {
  const int64_t n_newlines = line_base[n_tiles];
  const int64_t used = n_newlines - n_newlines % lpe;
  const int strip_cr = (int)flags[0];
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; t < n_tiles; t += stride) {
    int64_t* r = recs + t * FQ_TREC;
    const int64_t g = line_base[t];                          // line index of the tile's first byte
    r[0] = g;
    int64_t seq = 0;
    // every line touching the tile takes part; a tile that ends on line `used` itself only adds bytes of a header
    // line (used is a multiple of lpe), which are not sequence bytes
    if (line_base[t + 1] < used || (line_base[t + 1] == used && seq_line != 0)) {
      const int ph = (int)(((seq_line - g) % lpe + lpe) % lpe);
      seq = r[1 + ph];
    } else if (g < used) {                                   // the tile holds the end of the last entry: recount
      int64_t line = g;
      const int64_t end = min((t + 1) * (int64_t)FQ_TILE, n);
      for (int64_t i = t * (int64_t)FQ_TILE; i < end && line < used; ++i) {
        const uint8_t b = buf[i];
        if (b == FQ_NL) { ++line; continue; }
        if (line % lpe != seq_line) continue;
        if (strip_cr && b == FQ_CR && i + 1 < n && buf[i + 1] == FQ_NL) continue;
        ++seq;
      }
    }
    seq_count[t] = seq;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    totals[0] = n_newlines;
    totals[1] = used;
  }
}

// ---- encode --------------------------------------------------------------------------------------------------------
constexpr int FQ_SWORDS = (FQ_TILE + 64) / 16;               // staged 2-bit codes (32-bit words), aligned like the packed words
constexpr int FQ_EWORDS = (FQ_TILE + 128) / 32;              // read-end bits (32-bit words), aligned like the mask words

constexpr uint64_t FQ_REP01 = 0x0101010101010101ull, FQ_REP7F = 0x7f7f7f7f7f7f7f7full, FQ_REP80 = 0x8080808080808080ull;
__device__ __forceinline__ uint64_t fq_eq_bytes(uint64_t x, uint64_t rep) {     // 0x80 in every byte equal to rep's
  const uint64_t z = x ^ rep;
  const uint64_t t = (z & FQ_REP7F) + FQ_REP7F;
  return ~(t | z | FQ_REP7F);
}
// eight bytes -> sixteen bits of 2-bit codes (A C G T / a c g t -> 0 1 2 3) + a bit per byte that is none of them
__device__ __forceinline__ uint32_t fq_codes8(uint64_t x, uint32_t* invalid) {
  const uint64_t u = x & (0xDFull * FQ_REP01);               // fold lower case onto upper case (exact for A C G T)
  const uint64_t ok = fq_eq_bytes(u, 'A' * FQ_REP01) | fq_eq_bytes(u, 'C' * FQ_REP01) | fq_eq_bytes(u, 'G' * FQ_REP01) |
                      fq_eq_bytes(u, 'T' * FQ_REP01);
  uint64_t bad = (~ok & FQ_REP80) >> 7;                      // one bit per byte, at bit 8j
  bad = (bad | (bad >> 7)) & 0x0003000300030003ull;
  bad = (bad | (bad >> 14)) & 0x0000000F0000000Full;
  *invalid = (uint32_t)((bad | (bad >> 28)) & 0xFFull);
  uint64_t c = ((u >> 1) & (3ull * FQ_REP01)) ^ ((u >> 2) & FQ_REP01);
  c = (c | (c >> 6)) & 0x000F000F000F000Full;
  c = (c | (c >> 12)) & 0x000000FF000000FFull;
  return (uint32_t)((c | (c >> 24)) & 0xFFFFull);
}

// The general tile encoder: every lane classifies its own sixteen bytes whatever the line structure is.  LIST = false:
// tile = blockIdx.x.  LIST = true: the tiles the fast kernel below handed back (redo[0] = how many, redo[1..] = which).
template <bool LIST>
__global__ __launch_bounds__(BNPK_BLOCK) void fq_encode_kernel(const uint8_t* __restrict__ buf, int64_t n, int lpe,
                                                               int seq_line, uint8_t header, int check_plus,
                                                               const int64_t* __restrict__ flags,
                                                               const int64_t* __restrict__ recs,
                                                               const int64_t* __restrict__ seq_base, int64_t used,
                                                               unsigned long long* __restrict__ packed,
                                                               unsigned long long* __restrict__ ends,
                                                               unsigned long long* __restrict__ err,
                                                               const unsigned* __restrict__ redo) {
  __shared__ __attribute__((aligned(16))) unsigned stage[FQ_SWORDS];
  __shared__ unsigned ebits[FQ_EWORDS];
  __shared__ int smem[FQ_WAVES];
  const unsigned n_list = LIST ? redo[0] : 1u;
  for (unsigned item = LIST ? blockIdx.x : 0u; item < n_list; item += LIST ? gridDim.x : 1u) {
  const int64_t tile = LIST ? (int64_t)redo[1 + item] : (int64_t)blockIdx.x;
  if (LIST) __syncthreads();                                 // the staging areas of the previous tile have been read
  const int strip_cr = (int)flags[0];
  const int tid = threadIdx.x;
  const int64_t tile_base = tile * FQ_TILE;
  const int64_t g0 = recs[tile * FQ_TREC];              // absolute line of the tile's first byte
  const int64_t fbase = seq_base[tile];                          // flat base index of the tile's first sequence byte
  const int S = (int)(seq_base[tile + 1] - fbase);               // sequence bytes of the tile
  const int off32 = (int)(fbase & 31), off64 = (int)(fbase & 63);
  // line numbers relative to the tile's first line: 32-bit compares and a cheap phase instead of 64-bit % per chunk
  const int g0_phase = (int)(g0 % lpe);
  const int used_rel = (int)max((int64_t)-1, min(used - g0, (int64_t)1 << 30));      // lines [0, used_rel) take part
  const float inv_lpe = 1.0f / (float)lpe;
  for (int i = tid; i < FQ_SWORDS; i += BNPK_BLOCK) stage[i] = 0;
  for (int i = tid; i < FQ_EWORDS; i += BNPK_BLOCK) ebits[i] = 0;
  if (tile == 0 && tid == 0 && n > 0 && used > 0 && buf[0] != header) atomicMin(&err[0], 0ull);

  fq_chunk c[FQ_ITERS];
  int nls[FQ_ITERS], line[FQ_ITERS], total;
#pragma unroll
  for (int it = 0; it < FQ_ITERS; ++it) {
    c[it] = fq_load(buf, fq_chunk_pos(tile_base, it), n, strip_cr);
    nls[it] = __popc(c[it].nl);
  }
  fq_tile_prefix(nls, line, smem, &total);                   // (also orders the LDS clears above before the writes below)
  // the two bytes that follow every chunk: from the next lane, the next iteration's lane 0, or (last chunk of the
  // wavefront) from memory
  uint32_t follow[FQ_ITERS];
  {
    const int64_t wave_end = tile_base + (int64_t)(wave_id() + 1) * (FQ_ITERS * FQ_WAVE_BYTES);
    uint32_t tail = 0;
    if (wave_end < n) tail = buf[wave_end];
    if (wave_end + 1 < n) tail |= (uint32_t)buf[wave_end + 1] << 8;
#pragma unroll
    for (int it = FQ_ITERS - 1; it >= 0; --it) {
      const uint32_t mine = (uint32_t)c[it].lo & 0xffffu;
      uint32_t nxt = __shfl_down(mine, 1, 64);
      const uint32_t wrap = it + 1 < FQ_ITERS ? (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)c[it + 1 < FQ_ITERS ? it + 1 : it].lo & 0xffffu)) : tail;
      if (lane_id() == 63) nxt = wrap;
      follow[it] = nxt;
    }
  }
  // sequence bytes per chunk -> rank of every chunk's first sequence byte inside the tile
  uint32_t seqbits[FQ_ITERS];
  int nseq[FQ_ITERS], rank[FQ_ITERS];
#pragma unroll
  for (int it = 0; it < FQ_ITERS; ++it) {
    uint32_t nl = c[it].nl, todo = c[it].valid & ~c[it].skip, bits = 0;
    int ln = line[it], ph = fq_mod_small(g0_phase + ln, lpe, inv_lpe);
    while (true) {
      const uint32_t upto = nl ? ((1u << (__ffs(nl) - 1)) - 1u) : 0xffffu;
      if (ln < used_rel && ph == seq_line) bits |= todo & upto;
      if (!nl) break;
      todo &= ~upto;
      nl &= nl - 1;
      ++ln;
      ph = (ph + 1 == lpe) ? 0 : ph + 1;
    }
    seqbits[it] = bits;
    nseq[it] = __popc(bits);
  }
  int tot_seq;
  fq_tile_prefix(nseq, rank, smem, &tot_seq);
  unsigned long long bad = (unsigned long long)BNPK_NONE;
#pragma unroll
  for (int it = 0; it < FQ_ITERS; ++it) {
    const int64_t pos = fq_chunk_pos(tile_base, it);
    // 2-bit codes of the sequence bytes -> staging area; a base followed by a line end (newline, or CR + newline
    // when CRs are stripped) is the last base of its read
    uint32_t bits = seqbits[it];
    uint32_t endbits = bits & (c[it].skip >> 1);
    if ((bits >> 15) & 1u) {
      const uint32_t b16 = pos + 16 < n ? (follow[it] & 0xffu) : 0u, b17 = pos + 17 < n ? (follow[it] >> 8) : 0u;
      if (b16 == FQ_NL || (strip_cr && b16 == FQ_CR && b17 == FQ_NL)) endbits |= 1u << 15;
    }
    // all sixteen codes at once; the sequence bytes of a chunk form (at most a few) contiguous runs, each of which
    // is OR-ed into the staging words at its rank
    uint32_t inv_lo, inv_hi;
    const uint32_t codes = fq_codes8(c[it].lo, &inv_lo) | (fq_codes8(c[it].hi, &inv_hi) << 16);
    const uint32_t invalid = (inv_lo | (inv_hi << 8)) & bits;
    if (invalid) {
      const int j = __ffs(invalid) - 1;
      const unsigned long long at = (unsigned long long)(fbase + rank[it] + __popc(bits & ((1u << j) - 1u)));
      if (at < bad) bad = at;
    }
    int r = rank[it];
    uint32_t runs = bits;
    while (runs) {
      const int a = __ffs(runs) - 1;
      const int len = __ffs(~(runs >> a)) - 1;               // run of set bits starting at a (runs has 16 significant bits)
      const uint32_t m = len >= 16 ? ~0u : ((1u << (2 * len)) - 1u);
      uint32_t v = (codes >> (2 * a)) & m;
      const int bitpos = 2 * (off32 + r);
      const int sh = bitpos & 31;
      atomicOr(&stage[bitpos >> 5], v << sh);
      if (sh + 2 * len > 32) atomicOr(&stage[(bitpos >> 5) + 1], v >> (32 - sh));
      r += len;
      runs &= ~(((len >= 16 ? 0xffffu : ((1u << len) - 1u))) << a);
    }
    uint32_t eb = endbits;
    while (eb) {
      const int j = __ffs(eb) - 1;
      eb &= eb - 1;
      const int e = off64 + rank[it] + __popc(bits & ((1u << j) - 1u));
      atomicOr(&ebits[e >> 5], 1u << (e & 31));
    }
    // line ends: validation of the byte that starts the next line
    uint32_t nl = c[it].nl;
    if (nl) {
      int ln = line[it], ph = fq_mod_small(g0_phase + ln, lpe, inv_lpe);
      while (nl) {
        const int j = __ffs(nl) - 1;
        nl &= nl - 1;
        ++ln;                                                // the line that starts after this newline
        ph = (ph + 1 == lpe) ? 0 : ph + 1;
        if (ln < used_rel && (ph == 0 || (check_plus && ph == 2))) {
          const uint32_t b = j < 15 ? fq_byte(c[it], j + 1) : (pos + 16 < n ? (follow[it] & 0xffu) : 0u);
          if (ph == 0 && b != header) atomicMin(&err[0], (unsigned long long)((g0 + ln) / lpe));
          if (ph == 2 && b != '+') atomicMin(&err[1], (unsigned long long)((g0 + ln) / lpe));
        }
      }
    }
  }
  if (bad != (unsigned long long)BNPK_NONE) atomicMin(&err[2], bad);
  __syncthreads();
  // packed words: 64-bit word w of the tile = staged 32-bit words 2w, 2w + 1
  const int n_words = (off32 + S + 31) / 32;
  const int64_t word0 = fbase >> 5;
  for (int w = tid; w < n_words; w += BNPK_BLOCK) {
    const unsigned long long word = (unsigned long long)stage[2 * w] | ((unsigned long long)stage[2 * w + 1] << 32);
    const bool edge = (w == 0 && off32 != 0) || (w == n_words - 1 && ((off32 + S) & 31) != 0);
    if (edge) { if (word) atomicOr(&packed[word0 + w], word); }
    else packed[word0 + w] = word;
  }
  const int n_ew = (off64 + S + 63) / 64;
  const int64_t eword0 = fbase >> 6;
  for (int w = tid; w < n_ew; w += BNPK_BLOCK) {
    const unsigned long long word = (unsigned long long)ebits[2 * w] | ((unsigned long long)ebits[2 * w + 1] << 32);
    const bool edge = (w == 0 && off64 != 0) || (w == n_ew - 1 && ((off64 + S) & 63) != 0);
    if (edge) { if (word) atomicOr(&ends[eword0 + w], word); }
    else ends[eword0 + w] = word;
  }
  }
}

// ---- encode, the fast kernel --------------------------------------------------------------------------------------------
// The general kernel above spends ~390 vector instructions per sixteen bytes of text, on every byte of the file, and
// that — not memory — is its limit (SQ_INSTS_VALU * 4 cycles = its run time).  Half of a FASTQ file is quality and
// header text that only has to be searched for newlines, and what is left is, to all but one lane in ten, a plain run
// of sixteen bases.  So this kernel splits the work by what it is done on:
//   per byte of text   the newline mask (two dot products gather the flags), a copy of the tile in LDS, the tile-wide
//                      line number of every chunk, the start of every line
//   per line           start, phase, number of bases and (one scan over the lines) the rank of its first base in the
//                      tile and the number of sixteen-byte chunks it touches; the read-end bit (an atomic OR into the
//                      zeroed mask); the check of the byte that starts a header / '+' line; one 32-bit record per
//                      (line, chunk): chunk, first byte, length, rank
//   per record         dense lanes again: sixteen bytes from LDS -> codes (V_PERM_B32 looks the letter of every code
//                      up again and the XOR with the text is the validity test) -> OR-ed into the staging words.
//                      Lines of more than 256 bytes skip the queue: the whole workgroup walks their chunks.
// Tiles with more than FQ2_NLMAX newlines or more records than the queue holds (lines shorter than ~20 bytes) and the
// last tile of the text are handed to the general kernel through a list; the results are the same bits either way
// (tests/test_gpu_parity.py runs both).
constexpr int FQ2_NLMAX = 160 * FQ_ITERS;                               // lines 0 .. FQ2_NLMAX of a tile have a table entry
constexpr int FQ2_QCAP = 160 * FQ_ITERS;                                // records of lines of at most FQ2_SHORT chunks
constexpr int FQ2_SHORT = 16;
constexpr int FQ2_LONGMAX = FQ_TILE / (FQ2_SHORT * FQ_VEC - FQ_VEC) + 2;      // lines touching more than FQ2_SHORT chunks
constexpr int FQ2_CHUNKS = FQ_TILE / FQ_VEC;                 // 1024
constexpr int FQ2_WG_PER_CU = FQ_ITERS == 4 ? 6 : 8;                             // resident workgroups per CU the fast kernel is built for (LDS and registers)

// four text bytes -> four codes (low two bits of every byte); *z != 0 in every byte that is not one of ACGTacgt
__device__ __forceinline__ uint32_t fq_codes4(uint32_t w, uint32_t* z) {
  const uint32_t u = w & 0xDFDFDFDFu;
  const uint32_t c = ((u >> 1) & 0x03030303u) ^ ((u >> 2) & 0x01010101u);
  *z = u ^ __builtin_amdgcn_perm(0u, 0x54474341u, c);        // 'A' 'C' 'G' 'T' selected by the code
  return c;
}
// x mod m for 0 <= x < 4096 + m, 1 <= m <= 4, with inv = ceil(65536 / m): two 24-bit multiplies
__device__ __forceinline__ int fq_mod_tiny(int x, int m, int inv) { return x - (int)__umul24(__umul24((unsigned)x, (unsigned)inv) >> 16, (unsigned)m); }

// Persistent workgroups (the grid is what fits on the chip; workgroup g takes the tiles g, g + grid, ...): the text of
// the next tile is requested into registers as soon as this tile's copy is in LDS, so that a workgroup never sits out
// a memory latency between its tiles — without that the loads are in flight for a third of a tile's life only and the
// kernel runs at the rate of the latency, not of the instructions.
__global__ __launch_bounds__(BNPK_BLOCK, FQ2_WG_PER_CU) void fq_encode_fast_kernel(const uint8_t* __restrict__ buf, int64_t n, int lpe,
                                                                       int seq_line, uint8_t header, int check_plus,
                                                                       const int64_t* __restrict__ flags,
                                                                       const int64_t* __restrict__ recs,
                                                                       const int64_t* __restrict__ seq_base, int64_t used,
                                                                       int64_t tiles,
                                                                       unsigned long long* __restrict__ packed,
                                                                       unsigned long long* __restrict__ ends,
                                                                       unsigned long long* __restrict__ err,
                                                                       unsigned* __restrict__ redo) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  // one block of LDS carved by hand: six workgroups per CU is a budget of 27306 bytes
  constexpr int OFF_STAGE = FQ2_CHUNKS * 16, OFF_LTAB = OFF_STAGE + FQ_SWORDS * 4, OFF_RUNS = OFF_LTAB + (FQ2_NLMAX + 2) * 4;
  constexpr int OFF_LONG = OFF_RUNS + FQ2_QCAP * 4, OFF_SMEM = OFF_LONG + FQ2_LONGMAX * 8, LDS_BYTES = OFF_SMEM + 48;
  static_assert(OFF_STAGE % 16 == 0 && OFF_LTAB % 4 == 0 && OFF_LONG % 8 == 0 && LDS_BYTES * FQ2_WG_PER_CU <= 160 * 1024, "LDS layout");
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  u32x4* text = reinterpret_cast<u32x4*>(lds);               // the tile, chunk c = bytes [16c, 16c + 16)
  unsigned* stage = reinterpret_cast<unsigned*>(lds + OFF_STAGE);
  unsigned* ltab = reinterpret_cast<unsigned*>(lds + OFF_LTAB);                     // line l: its start in the tile (low 16 bits)
  unsigned* runs = reinterpret_cast<unsigned*>(lds + OFF_RUNS);
  uint2* longs = reinterpret_cast<uint2*>(lds + OFF_LONG);
  int* smem = reinterpret_cast<int*>(lds + OFF_SMEM);        // 8 ints of scan scratch, then the long-line and record counters
  int* n_long = smem + 8;
  const int tid = threadIdx.x;
  const int strip_cr = (int)flags[0];
  const int inv_lpe = (65536 + lpe - 1) / lpe;
  const unsigned chunk0 = (unsigned)wave_id() * (FQ_ITERS * BNPK_WAVE) + lane_id();
  const int64_t full_tiles = n >= FQ_TILE + 16 ? (n - 16) / FQ_TILE : 0;     // tiles [0, full_tiles) have their 16 KiB and two more bytes
  if (blockIdx.x == 0 && tid == 0) {
    for (int64_t t = full_tiles; t < tiles; ++t) redo[1 + atomicAdd(&redo[0], 1u)] = (unsigned)t;   // the end of the text: bounds checks live in the general kernel
    if (full_tiles > 0 && used > 0 && buf[0] != header) atomicMin(&err[0], 0ull);
  }
  // the text of the tile about to be worked on and what the tables say of it.  The table words and the two bytes
  // behind the tile are loaded per lane (lane 0: first line, 1: first base, 2: end of the bases; even lanes: the byte
  // behind the tile, odd lanes: the one after) and read with readlane a tile later: a scalar load, or a vector load the
  // compiler knows to be uniform, is waited for where it is issued.
  u32x4 pre[FQ_ITERS];
  int64_t pre_tab = 0;
  uint32_t pre_after = 0;
  auto request = [&](int64_t t) {
    const u32x4* src = reinterpret_cast<const u32x4*>(buf + t * FQ_TILE) + chunk0;
#pragma unroll
    for (int it = 0; it < FQ_ITERS; ++it) pre[it] = __builtin_nontemporal_load(src + it * BNPK_WAVE);
    const int64_t* tab = lane_id() == 0 ? recs + t * FQ_TREC : seq_base + t + (lane_id() == 1 ? 0 : 1);
    pre_tab = *tab;
    pre_after = buf[(t + 1) * FQ_TILE + (lane_id() & 1)];
  };
  auto lane64 = [](int64_t v, int lane) {
    return (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), lane) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, lane));
  };
  if ((int64_t)blockIdx.x < full_tiles) request(blockIdx.x);
  for (int i = tid; i < FQ_SWORDS; i += BNPK_BLOCK) stage[i] = 0;
  if (tid == 0) {
    reinterpret_cast<unsigned short*>(ltab)[0] = 0;          // line 0 starts at byte 0, in every tile
    *n_long = 0;
  }
  __syncthreads();
  // The per-entry phase keeps one wavefront busy and the record loop gives the first wavefronts one round more.  The
  // wavefronts of a workgroup sit on different SIMDs, and always the same ones: so the roles rotate from tile to tile
  // (logical wavefront rw = physical + turn), or one SIMD of every CU carries a third more than the others.
  int turn = 0;
  for (int64_t tile = blockIdx.x; tile < full_tiles; tile += gridDim.x, ++turn) {
  const int rw = (wave_id() + turn) & (FQ_WAVES - 1), rtid = rw * BNPK_WAVE + lane_id();
  const int64_t g0 = lane64(pre_tab, 0), fbase = lane64(pre_tab, 1);
  const int S = (int)(lane64(pre_tab, 2) - fbase);
  const uint32_t after0 = (uint32_t)__builtin_amdgcn_readlane((int)pre_after, 0), after1 = (uint32_t)__builtin_amdgcn_readlane((int)pre_after, 1);
  const int off32 = (int)(fbase & 31);
  const int g0_phase = lpe == 4 ? (int)(g0 & 3) : lpe == 2 ? (int)(g0 & 1) : lpe == 1 ? 0 : (int)(((uint32_t)(g0 >> 32) % 3u + (uint32_t)g0 % 3u) % 3u);
  const int used_rel = (int)max((int64_t)-1, min(used - g0, (int64_t)1 << 30));
  // (no barrier between tiles: the text was last read before the barrier in front of the write-out, the write-out leaves
  // the staging words it read zeroed, and everything else is rewritten at least two barriers from here)
  // ---- per byte: newline masks, the copy in LDS, line numbers, line starts ----
  uint32_t nlm[FQ_ITERS];
  int nls[FQ_ITERS], line[FQ_ITERS], L;
#pragma unroll
  for (int it = 0; it < FQ_ITERS; ++it) {
    const u32x4 v = pre[it];
    text[chunk0 + it * BNPK_WAVE] = v;
    nlm[it] = ~fq_nomatch16((uint64_t)v.x | ((uint64_t)v.y << 32), (uint64_t)v.z | ((uint64_t)v.w << 32), 0x01010101u * FQ_NL) & 0xffffu;
    nls[it] = __popc(nlm[it]);
  }
  if (tile + gridDim.x < full_tiles) request(tile + gridDim.x);      // (uniform) on its way while this tile is worked on
  fq_tile_prefix16(nls, line, smem, &L);                     // L = newlines of the tile = index of its last (open) line
  if (L > FQ2_NLMAX) {                                       // (uniform) too many lines for the table
    if (tid == 0) redo[1 + atomicAdd(&redo[0], 1u)] = (unsigned)tile;
    continue;
  }
#pragma unroll
  for (int it = 0; it < FQ_ITERS; ++it) {
    uint32_t nl = nlm[it];
    int ln = line[it];
    const unsigned at = (chunk0 + it * BNPK_WAVE) * FQ_VEC;
    while (__ballot(nl != 0) != 0ull) {
      if (nl) {
        ++ln;
        reinterpret_cast<unsigned short*>(ltab)[2 * ln] = (unsigned short)(at + __ffs(nl));
        nl &= nl - 1;
      }
    }
  }
  __syncthreads();
  // ---- per record: sixteen bytes of chunk c, of which [a, a + len) are bases; r = rank of the first of them in the tile ----
  unsigned long long bad = (unsigned long long)BNPK_NONE;
  auto encode = [&](unsigned cidx, int a, int len, int r) {
    const u32x4 v = text[cidx];
    uint32_t z0, z1, z2, z3;
    const uint32_t c0 = fq_codes4(v.x, &z0), c1 = fq_codes4(v.y, &z1), c2 = fq_codes4(v.z, &z2), c3 = fq_codes4(v.w, &z3);
    const uint32_t codes = __builtin_amdgcn_udot4(c0, 0x40100401u, 0u, false) | (__builtin_amdgcn_udot4(c1, 0x40100401u, 0u, false) << 8) |
                           (__builtin_amdgcn_udot4(c2, 0x40100401u, 0u, false) << 16) | (__builtin_amdgcn_udot4(c3, 0x40100401u, 0u, false) << 24);
    const uint32_t val = (codes >> (2 * a)) & (~0u >> (32 - 2 * len));
    // bytes of the run that are no base: the flags of the non-zero bytes of z, gathered like the newline flags
    const uint32_t g0z = ((z0 & 0x7f7f7f7fu) + 0x7f7f7f7fu) | z0 | 0x7f7f7f7fu, g1z = ((z1 & 0x7f7f7f7fu) + 0x7f7f7f7fu) | z1 | 0x7f7f7f7fu;
    const uint32_t g2z = ((z2 & 0x7f7f7f7fu) + 0x7f7f7f7fu) | z2 | 0x7f7f7f7fu, g3z = ((z3 & 0x7f7f7f7fu) + 0x7f7f7f7fu) | z3 | 0x7f7f7f7fu;
    const uint32_t bias = 0u - 0x7fu * 255u;
    const uint32_t lo8 = __builtin_amdgcn_udot4(g1z, 0x80402010u, __builtin_amdgcn_udot4(g0z, 0x08040201u, bias, false), false);
    const uint32_t hi8 = __builtin_amdgcn_udot4(g3z, 0x80402010u, __builtin_amdgcn_udot4(g2z, 0x08040201u, bias, false), false);
    const uint32_t nz = (((lo8 >> 7) | (hi8 << 1)) >> a) & (0xffffu >> (16 - len));
    if (nz) {
      const unsigned long long where = (unsigned long long)(fbase + r + (__ffs(nz) - 1));
      if (where < bad) bad = where;
    }
    const int bitpos = 2 * (off32 + r);
    const int sh = bitpos & 31;
    atomicOr(&stage[bitpos >> 5], val << sh);
    if (sh + 2 * len > 32) atomicOr(&stage[(bitpos >> 5) + 1], val >> (32 - sh));
  };
  // ---- per entry: thread k has the lines k * lpe - p0 + {0 .. lpe - 1} of the tile (p0 = phase of the tile's first line),
  // so the lanes that meet a sequence line sit side by side in the first wavefront(s) and the others pass through ----
  const uint8_t* bytes = reinterpret_cast<const uint8_t*>(text);
  const bool open_line_ends = after0 == FQ_NL || (strip_cr && after0 == FQ_CR && after1 == FQ_NL);
  int* n_records = smem + 9;
  {
    const int n_entries = (int)(__umul24((unsigned)(L + g0_phase), (unsigned)inv_lpe) >> 16) + 1;     // (L + p0) / lpe + 1
    // what entry k contributes: bases and chunks of its sequence line (and the checks of its header / '+' bytes)
    auto entry = [&](int k, int& s, int& e, int& len, int& nch, bool& ends_read) {
      s = e = len = nch = 0;
      ends_read = false;
      if (k >= n_entries) return;
      const int l0 = (int)__umul24((unsigned)k, (unsigned)lpe) - g0_phase;
      // the byte that starts a header / '+' line is checked by the tile that holds the newline before it
#pragma unroll
      for (int j = 0; j <= 2; j += 2) {
        const int l = l0 + j;
        if ((j == 0 || (check_plus && lpe > 2)) && l >= 1 && l <= L && l < used_rel) {
          const int at = (int)(ltab[l] & 0xffffu);
          const uint32_t b = at < FQ_TILE ? bytes[at] : after0;
          if (b != (j == 0 ? (uint32_t)header : (uint32_t)'+')) atomicMin(&err[j == 0 ? 0 : 1], (unsigned long long)((g0 + l) / lpe));
        }
      }
      const int l = l0 + seq_line;
      if (l >= 0 && l <= L && l < used_rel) {
        s = (int)(ltab[l] & 0xffffu);
        e = l < L ? (int)(ltab[l + 1] & 0xffffu) - 1 : FQ_TILE;
        if (strip_cr && e > s && bytes[e - 1] == FQ_CR && (l < L || after0 == FQ_NL)) --e;
        len = e - s;
        if (len > 0) nch = ((e - 1) >> 4) - (s >> 4) + 1;
        ends_read = l < L || open_line_ends;
      }
    };
    // rank = bases of the tile before the line, qat = records before its own
    auto emit = [&](int s, int e, int len, int nch, bool ends_read, int rank, int qat) {
      if (len <= 0) return;
      if (ends_read) {                                       // its last base ends a read
        const int64_t eb = fbase + rank + len - 1;
        atomicOr(&ends[eb >> 6], 1ull << (eb & 63));
      }
      if (nch > FQ2_SHORT) {
        longs[atomicAdd(n_long, 1)] = make_uint2((unsigned)s | ((unsigned)e << 16), (unsigned)rank);
      } else {
        const int c0 = s >> 4;
        for (int q = 0; q < nch; ++q) {
          const int lo = q == 0 ? s & 15 : 0, hi = min(e - ((c0 + q) << 4), FQ_VEC);
          const int r = rank + (q == 0 ? 0 : ((c0 + q) << 4) - s);
          if (qat + q < FQ2_QCAP) runs[qat + q] = (unsigned)(c0 + q) | ((unsigned)lo << 10) | ((unsigned)(hi - lo - 1) << 14) | ((unsigned)r << 18);
        }
      }
    };
    if (n_entries <= BNPK_WAVE) {                            // (uniform) the usual tile: one wavefront, no barrier, no LDS round trip
      if (rw == 0) {
        int s, e, len, nch;
        bool ends_read;
        entry(lane_id(), s, e, len, nch, ends_read);
        const int v = len | (nch > FQ2_SHORT ? 0 : nch << 16);
        const int inc = wave_inclusive_scan(v);
        const int x = inc - v;
        emit(s, e, len, nch, ends_read, x & 0xffff, x >> 16);
        if (lane_id() == 63) *n_records = inc >> 16;
      }
    } else {
      int rank0 = 0, R0 = 0;
      for (int base = 0; base < n_entries; base += BNPK_BLOCK) {
        int s, e, len, nch;
        bool ends_read;
        entry(base + rtid, s, e, len, nch, ends_read);
        int total, x;
        {                                                    // exclusive scan in the order of the logical wavefronts
          const int v = len | (nch > FQ2_SHORT ? 0 : nch << 16);
          const int inc = wave_inclusive_scan(v);
          if (lane_id() == 63) smem[rw] = inc;
          __syncthreads();
          int before = 0;
          total = 0;
#pragma unroll
          for (int w = 0; w < FQ_WAVES; ++w) {
            const int t = smem[w];
            before += w < rw ? t : 0;
            total += t;
          }
          x = before + inc - v;
          __syncthreads();
        }
        emit(s, e, len, nch, ends_read, rank0 + (x & 0xffff), R0 + (x >> 16));
        rank0 += total & 0xffff;
        R0 += total >> 16;
      }
      if (tid == 0) *n_records = R0;
    }
  }
  __syncthreads();
  const int R = *n_records;
  if (R > FQ2_QCAP) {                                        // (uniform; the read ends set above are set again, harmlessly)
    __syncthreads();                                         // (everybody has read the count and the long-line counter is at rest)
    if (tid == 0) {
      redo[1 + atomicAdd(&redo[0], 1u)] = (unsigned)tile;
      *n_long = 0;
    }
    continue;
  }
  for (int t = rtid; t < R; t += BNPK_BLOCK) {
    const unsigned rec = runs[t];
    encode(rec & 1023u, (int)((rec >> 10) & 15u), (int)((rec >> 14) & 15u) + 1, (int)(rec >> 18));
  }
  {
    const int nl_long = *n_long;
    for (int i = 0; i < nl_long; ++i) {                      // (uniform) the workgroup walks the chunks of a long line
      const uint2 ll = longs[i];
      const int s = (int)(ll.x & 0xffffu), e = (int)(ll.x >> 16), rank = (int)ll.y;
      const int c0 = s >> 4, nch = ((e - 1) >> 4) - c0 + 1;
      for (int q = rtid; q < nch; q += BNPK_BLOCK) {
        const int lo = q == 0 ? s & 15 : 0, hi = min(e - ((c0 + q) << 4), FQ_VEC);
        encode((unsigned)(c0 + q), lo, hi - lo, rank + (q == 0 ? 0 : ((c0 + q) << 4) - s));
      }
    }
  }
  if (bad != (unsigned long long)BNPK_NONE) atomicMin(&err[2], bad);
  __syncthreads();
  const int n_words = (off32 + S + 31) / 32;
  const int64_t word0 = fbase >> 5;
  if (tid == 0) *n_long = 0;
  for (int w = tid; w < n_words; w += BNPK_BLOCK) {
    const unsigned long long word = (unsigned long long)stage[2 * w] | ((unsigned long long)stage[2 * w + 1] << 32);
    stage[2 * w] = 0;                                        // ready for the next tile
    stage[2 * w + 1] = 0;
    const bool edge = (w == 0 && off32 != 0) || (w == n_words - 1 && ((off32 + S) & 31) != 0);
    if (edge) { if (word) atomicOr(&packed[word0 + w], word); }
    else packed[word0 + w] = word;
  }
  }
}

// ---- census, the fast kernel ---------------------------------------------------------------------------------------------
// The same split for the census: newline masks and line starts per byte of text, the payload of every line (and so of
// every line phase, relative to the tile's first line) per line.  Nothing but the newline masks is computed per byte, the
// text stays in registers, and the workgroups are persistent with the next tile in flight.
constexpr int FQC_WG_PER_CU = 8;

__global__ __launch_bounds__(BNPK_BLOCK, FQC_WG_PER_CU) void fq_census_fast_kernel(const uint8_t* __restrict__ buf, int64_t n, int lpe,
                                                                                    const int64_t* __restrict__ flags,
                                                                                    int64_t* __restrict__ recs,
                                                                                    int64_t* __restrict__ newlines, int64_t tiles,
                                                                                    unsigned* __restrict__ redo) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  __shared__ unsigned short lstart[FQ2_NLMAX + 2];           // line l of the tile starts at byte lstart[l]
  __shared__ int smem[FQ_WAVES + 4];
  __shared__ int acc[FQ_MAXLPE];
  const int tid = threadIdx.x;
  const int strip_cr = (int)flags[0];
  const int inv_lpe = (65536 + lpe - 1) / lpe;
  const unsigned chunk0 = (unsigned)wave_id() * (FQ_ITERS * BNPK_WAVE) + lane_id();
  const int64_t full_tiles = n >= FQ_TILE + 16 ? (n - 16) / FQ_TILE : 0;
  if (blockIdx.x == 0 && tid == 0)
    for (int64_t t = full_tiles; t < tiles; ++t) redo[1 + atomicAdd(&redo[0], 1u)] = (unsigned)t;
  u32x4 pre[FQ_ITERS];
  uint32_t pre_after = 0;
  auto request = [&](int64_t t) {
    const u32x4* src = reinterpret_cast<const u32x4*>(buf + t * FQ_TILE) + chunk0;
#pragma unroll
    for (int it = 0; it < FQ_ITERS; ++it) pre[it] = __builtin_nontemporal_load(src + it * BNPK_WAVE);
    if (strip_cr) pre_after = buf[(t + 1) * FQ_TILE + (lane_id() & 1)];
  };
  if ((int64_t)blockIdx.x < full_tiles) request(blockIdx.x);
  for (int64_t tile = blockIdx.x; tile < full_tiles; tile += gridDim.x) {
    const int64_t tile_base = tile * FQ_TILE;
    const uint32_t after0 = (uint32_t)__builtin_amdgcn_readlane((int)pre_after, 0);
    __syncthreads();                                         // the previous tile's table and sums have been read
    if (tid < FQ_MAXLPE) acc[tid] = 0;
    if (tid == 0) lstart[0] = 0;
    uint32_t nlm[FQ_ITERS];
    int nls[FQ_ITERS], line[FQ_ITERS], L;
#pragma unroll
    for (int it = 0; it < FQ_ITERS; ++it) {
      const u32x4 v = pre[it];
      nlm[it] = ~fq_nomatch16((uint64_t)v.x | ((uint64_t)v.y << 32), (uint64_t)v.z | ((uint64_t)v.w << 32), 0x01010101u * FQ_NL) & 0xffffu;
      nls[it] = __popc(nlm[it]);
    }
    if (tile + gridDim.x < full_tiles) request(tile + gridDim.x);
    fq_tile_prefix16(nls, line, smem, &L);
    if (L > FQ2_NLMAX) {                                     // (uniform)
      if (tid == 0) redo[1 + atomicAdd(&redo[0], 1u)] = (unsigned)tile;
      continue;
    }
#pragma unroll
    for (int it = 0; it < FQ_ITERS; ++it) {
      uint32_t nl = nlm[it];
      int ln = line[it];
      const unsigned at = (chunk0 + it * BNPK_WAVE) * FQ_VEC;
      while (__ballot(nl != 0) != 0ull) {
        if (nl) {
          ++ln;
          lstart[ln] = (unsigned short)(at + __ffs(nl));
          nl &= nl - 1;
        }
      }
    }
    __syncthreads();
    int cnt[FQ_MAXLPE] = {0, 0, 0, 0};
    for (int base = 0; base <= L; base += BNPK_BLOCK) {
      const int l = base + tid;
      if (l <= L) {
        const int s = lstart[l];
        int e = l < L ? (int)lstart[l + 1] - 1 : FQ_TILE;
        if (strip_cr && e > s && buf[tile_base + e - 1] == FQ_CR && (l < L || after0 == FQ_NL)) --e;
        const int ph = fq_mod_tiny(l, lpe, inv_lpe);
#pragma unroll
        for (int p = 0; p < FQ_MAXLPE; ++p) cnt[p] += p == ph ? e - s : 0;
      }
    }
#pragma unroll
    for (int p = 0; p < FQ_MAXLPE; ++p) {
      const int sum = (int)wave_sum((unsigned)cnt[p]);
      if (lane_id() == 0 && sum) atomicAdd(&acc[p], sum);
    }
    __syncthreads();
    if (tid == 0) {
      int64_t* r = recs + tile * FQ_TREC;
      newlines[tile] = L;
#pragma unroll
      for (int p = 0; p < FQ_MAXLPE; ++p) r[1 + p] = acc[p];
    }
  }
}

// The packed words that more than one tile writes (the word a tile's first base falls in) are OR-ed together and have to
// start at zero; every other word is stored whole.  Zeroing just those — one per tile, plus the two pad words behind the
// last base — replaces a memset of the whole packed array (1.9 GB per 50 M reads).
__global__ void fq_zero_edges_kernel(const int64_t* __restrict__ seq_base, int64_t tiles, unsigned long long* __restrict__ packed) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; t <= tiles; t += stride) {
    const int64_t b = seq_base[t];
    packed[b >> 5] = 0ull;
    if (t == tiles) packed[(b >> 5) + 1] = 0ull;
  }
}

// ---- k-mer start mask from the read-end mask --------------------------------------------------------------------------
__global__ __launch_bounds__(BNPK_BLOCK) void fq_starts_kernel(const unsigned long long* __restrict__ ends,
                                                               int64_t n_bases, int k,
                                                               unsigned long long* __restrict__ starts,
                                                               unsigned long long* __restrict__ count) {
  const int64_t n_words = (n_bases + 63) / 64;
  int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned long long local = 0;
  for (; w < n_words; w += stride) {
    unsigned long long lo = ends[w], hi = ends[w + 1];       // the mask has one pad word
    // OR of the window [i, i + k - 2] for every bit i, by doubling
    int have = 1;
    const int need = k - 1;
    unsigned long long x_lo = need > 0 ? lo : 0ull, x_hi = need > 0 ? hi : 0ull;
    while (have < need) {
      const int s = min(have, need - have);
      x_lo |= (x_lo >> s) | (x_hi << (64 - s));
      x_hi |= x_hi >> s;
      have += s;
    }
    unsigned long long v = ~x_lo;
    const int64_t left = n_bases - w * 64;                   // bits of this word that are bases
    if (left < 64) v &= (1ull << left) - 1ull;
    starts[w] = v;
    local += __popcll(v);
  }
  local = wave_reduce_sum(local);
  if (lane_id() == 0 && local) atomicAdd(count, local);
}

}  // namespace

extern "C" {

int64_t bnpk_fastq_tiles(int64_t n_bytes) { return n_bytes <= 0 ? 0 : ceil_div(n_bytes, FQ_TILE); }

int bnpk_fastq_census(bnpk_ctx* ctx, const uint8_t* d_buf, int64_t n, int lines_per_entry, int seq_line,
                      int64_t* d_tile_table, int64_t* h_totals, void* stream) {
  if (!ctx || n < 0 || lines_per_entry < 1 || lines_per_entry > FQ_MAXLPE || seq_line < 0 || seq_line >= lines_per_entry ||
      !d_tile_table || !h_totals || (n > 0 && !d_buf))
    return BNPK_ERR_ARG;
  if (((uintptr_t)d_buf & 15) != 0) return BNPK_ERR_ALIGN;
  h_totals[0] = h_totals[1] = h_totals[2] = h_totals[3] = 0;
  const int64_t tiles = bnpk_fastq_tiles(n);
  if (tiles == 0) return BNPK_OK;
  if (tiles > BNPK_MAX_BLOCKS) return BNPK_ERR_RANGE;
  hipStream_t s = (hipStream_t)stream;
  // table layout: [0 .. 8) flags + totals, records (tiles * FQ_TREC), first lines (tiles + 1), sequence bases (tiles + 1)
  int64_t* flags = d_tile_table;
  int64_t* totals = d_tile_table + 4;
  int64_t* recs = d_tile_table + 8;
  int64_t* line_base = recs + tiles * FQ_TREC;
  int64_t* seq_base = line_base + tiles + 1;
  void* scratch = nullptr;
  const size_t scan_bytes = (bnpk_scan_scratch_bytes(tiles) + 63) & ~(size_t)63;
  BNPK_CHECK(bnpk_scratch(ctx, scan_bytes + (size_t)(tiles + 1) * 4 + 64, &scratch, (hipStream_t)stream));
  int64_t* scan_scratch = (int64_t*)scratch;
  unsigned* redo = (unsigned*)((char*)scratch + scan_bytes);
  {
    bnpk_timer t(ctx, "fastq_census", s);
    hipLaunchKernelGGL(fq_detect_cr_kernel, dim3(1), dim3(64), 0, s, d_buf, n, lines_per_entry, flags);
    if (ctx->fastq_encoder == 0) {
      hipLaunchKernelGGL((fq_census_kernel<false>), dim3((unsigned)tiles), dim3(BNPK_BLOCK), 0, s, d_buf, n, lines_per_entry,
                         (const int64_t*)flags, recs, line_base, (const unsigned*)nullptr);
    } else {                                                 // the fast census, then the general one over the tiles it handed back
      BNPK_HIP(ctx, hipMemsetAsync(redo, 0, 4, s));
      const unsigned fast_grid = (unsigned)std::min<int64_t>(tiles, (int64_t)ctx->compute_units * FQC_WG_PER_CU);
      hipLaunchKernelGGL(fq_census_fast_kernel, dim3(fast_grid), dim3(BNPK_BLOCK), 0, s, d_buf, n, lines_per_entry,
                         (const int64_t*)flags, recs, line_base, tiles, redo);
      const unsigned grid = (unsigned)std::min<int64_t>(tiles, (int64_t)ctx->compute_units * 8);
      hipLaunchKernelGGL((fq_census_kernel<true>), dim3(grid), dim3(BNPK_BLOCK), 0, s, d_buf, n, lines_per_entry,
                         (const int64_t*)flags, recs, line_base, (const unsigned*)redo);
    }
    BNPK_HIP(ctx, hipGetLastError());
    BNPK_CHECK(bnpk_scan_launch(ctx, line_base, tiles, 1, line_base, true, scan_scratch, s));      // newline counts -> first lines
    hipLaunchKernelGGL(fq_select_kernel, dim3(grid_for(ceil_div(tiles, 256))), dim3(256), 0, s, d_buf, n, lines_per_entry,
                       seq_line, tiles, (const int64_t*)line_base, (const int64_t*)flags, recs, seq_base, totals);
    BNPK_HIP(ctx, hipGetLastError());
    BNPK_CHECK(bnpk_scan_launch(ctx, seq_base, tiles, 1, seq_base, true, scan_scratch, s));        // sequence bytes -> flat base indices
  }
  int64_t host[2] = {0, 0}, bases = 0, cr = 0;
  BNPK_HIP(ctx, hipMemcpyAsync(host, totals, sizeof(host), hipMemcpyDeviceToHost, s));
  BNPK_HIP(ctx, hipMemcpyAsync(&bases, seq_base + tiles, 8, hipMemcpyDeviceToHost, s));
  BNPK_HIP(ctx, hipMemcpyAsync(&cr, flags, 8, hipMemcpyDeviceToHost, s));
  BNPK_HIP(ctx, hipStreamSynchronize(s));
  h_totals[0] = host[0];
  h_totals[1] = host[1];
  h_totals[2] = bases;
  h_totals[3] = cr;
  return BNPK_OK;
}

int64_t bnpk_fastq_table_words(int64_t n_bytes) {
  const int64_t tiles = bnpk_fastq_tiles(n_bytes);
  return 8 + tiles * FQ_TREC + 2 * (tiles + 1);
}

int bnpk_fastq_encode(bnpk_ctx* ctx, const uint8_t* d_buf, int64_t n, int lines_per_entry, int seq_line, uint8_t header,
                      int check_plus, const int64_t* d_tile_table, int64_t n_lines_used, int64_t n_bases,
                      uint64_t* d_packed, uint64_t* d_row_ends, int64_t* d_err3, void* stream) {
  if (!ctx || n < 0 || lines_per_entry < 1 || lines_per_entry > FQ_MAXLPE || seq_line < 0 || seq_line >= lines_per_entry ||
      !d_tile_table || !d_packed || !d_row_ends || !d_err3 || n_bases < 0 || n_lines_used < 0 || (n > 0 && !d_buf))
    return BNPK_ERR_ARG;
  if (check_plus && lines_per_entry < 3) return BNPK_ERR_ARG;
  if (((uintptr_t)d_buf & 15) != 0) return BNPK_ERR_ALIGN;
  const int64_t tiles = bnpk_fastq_tiles(n);
  if (tiles > BNPK_MAX_BLOCKS) return BNPK_ERR_RANGE;
  hipStream_t s = (hipStream_t)stream;
  bnpk_timer t(ctx, "fastq_encode", s);
  BNPK_HIP(ctx, hipMemsetAsync(d_row_ends, 0, (size_t)(n_bases / 64 + 2) * 8, s));
  BNPK_CHECK(bnpk_fill_i64(ctx, d_err3, 3, BNPK_NONE, stream));
  if (tiles == 0) {
    BNPK_HIP(ctx, hipMemsetAsync(d_packed, 0, (size_t)(n_bases / 32 + 2) * 8, s));
    return BNPK_OK;
  }
  const int64_t* flags = d_tile_table;
  const int64_t* recs = d_tile_table + 8;
  const int64_t* seq_base = recs + tiles * FQ_TREC + tiles + 1;
  unsigned long long* packed = reinterpret_cast<unsigned long long*>(d_packed);
  hipLaunchKernelGGL(fq_zero_edges_kernel, dim3(grid_for(ceil_div(tiles + 1, 256))), dim3(256), 0, s, seq_base, tiles, packed);
  unsigned long long* ends = reinterpret_cast<unsigned long long*>(d_row_ends);
  unsigned long long* err = reinterpret_cast<unsigned long long*>(d_err3);
  if (ctx->fastq_encoder == 0) {
    hipLaunchKernelGGL((fq_encode_kernel<false>), dim3((unsigned)tiles), dim3(BNPK_BLOCK), 0, s, d_buf, n, lines_per_entry,
                       seq_line, header, check_plus, flags, recs, seq_base, n_lines_used, packed, ends, err,
                       (const unsigned*)nullptr);
  } else {
    // the fast kernel, then the general one over the tiles it handed back (usually none: the grid finds an empty list)
    void* scratch = nullptr;
    BNPK_CHECK(bnpk_scratch(ctx, (size_t)(tiles + 1) * 4 + 64, &scratch, s));
    unsigned* redo = (unsigned*)scratch;
    BNPK_HIP(ctx, hipMemsetAsync(redo, 0, 4, s));
    const unsigned fast_grid = (unsigned)std::min<int64_t>(tiles, (int64_t)ctx->compute_units * FQ2_WG_PER_CU);
    hipLaunchKernelGGL(fq_encode_fast_kernel, dim3(fast_grid), dim3(BNPK_BLOCK), 0, s, d_buf, n, lines_per_entry,
                       seq_line, header, check_plus, flags, recs, seq_base, n_lines_used, tiles, packed, ends, err, redo);
    const unsigned grid = (unsigned)std::min<int64_t>(tiles, (int64_t)ctx->compute_units * 8);
    hipLaunchKernelGGL((fq_encode_kernel<true>), dim3(grid), dim3(BNPK_BLOCK), 0, s, d_buf, n, lines_per_entry, seq_line,
                       header, check_plus, flags, recs, seq_base, n_lines_used, packed, ends, err, (const unsigned*)redo);
  }
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_kmer_starts_from_ends(bnpk_ctx* ctx, const uint64_t* d_row_ends, int64_t n_bases, int k, uint64_t* d_starts,
                               int64_t* d_count, void* stream) {
  if (!ctx || n_bases < 0 || k < 1 || k > 64 || !d_row_ends || !d_starts || !d_count) return BNPK_ERR_ARG;   // (the window OR looks one word ahead)
  hipStream_t s = (hipStream_t)stream;
  bnpk_timer t(ctx, "kmer_starts_from_ends", s);
  BNPK_HIP(ctx, hipMemsetAsync(d_count, 0, 8, s));
  BNPK_HIP(ctx, hipMemsetAsync(d_starts + (n_bases + 63) / 64, 0, (size_t)(n_bases / 64 + 2 - (n_bases + 63) / 64) * 8, s));
  const int64_t n_words = (n_bases + 63) / 64;
  if (n_words == 0) return BNPK_OK;
  const int64_t blocks = std::min<int64_t>(ceil_div(n_words, BNPK_BLOCK), (int64_t)ctx->compute_units * 16);
  hipLaunchKernelGGL(fq_starts_kernel, dim3((unsigned)blocks), dim3(BNPK_BLOCK), 0, s,
                     reinterpret_cast<const unsigned long long*>(d_row_ends), n_bases, k,
                     reinterpret_cast<unsigned long long*>(d_starts), reinterpret_cast<unsigned long long*>(d_count));
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

}  // extern "C"
