// Sparse k-mer histogram (A9 for k > 13 == np.unique(hashes, return_counts=True)) as an MSD radix
// partition through HBM followed by an in-LDS finishing sort, all hand-written for gfx950.
//
// What the hardware rewards (scripts/exp/exp_write.hip, 1.2 G keys on MI355X): a radix scatter whose
// (tile, bucket) runs start at arbitrary 8-byte offsets writes at 1.8-3.1 TB/s; the SAME runs issued as whole
// 128-byte-aligned lines write at 5.3 TB/s (the streaming rate) even with 1024 buckets and 16-key runs.  So the
// partition kernel keeps a software write-combining buffer in LDS: every workgroup owns a contiguous slab of the
// input, stages up to 16 Ki keys grouped by digit, flushes only whole aligned lines of each bucket and carries
// the < 16 leftover keys of every bucket into the next round.  Digits are up to 10 bits wide (1024 buckets), so
// 6e9 31-mers need two passes through HBM (the first fused with k-mer generation: the hashes are never stored
// in read order) before the buckets (~6 K keys) fit the finishing kernel, which sorts them in LDS (12-bit
// counting sort + exact ranking inside the ~1.4-key bins), run-length-counts the duplicates and writes
// (key, count) at the final sorted position.  MSD ranks need no stability, so they come from plain LDS atomics.
#include <algorithm>
#include <type_traits>

#include "common.h"
#include "kmer_gen.h"
#include "scan.h"

#ifndef RP_ABL
#define RP_ABL 0            // experiment builds only: 1 no global stores, 2 no staging of the new keys, 4 no rank atomics, 8 no carry staging / readback, 16 all stores of a workgroup into one 128 KiB window
#endif
#ifndef RP_L1_LINE
#define RP_L1_LINE 16       // flush granule (keys) of the fused first level
#endif
#ifndef RR_ABL
#define RR_ABL 0            // experiment builds only (rp_ring_kernel): 1 no global stores, 2 no flush at all (counts only), 4 no ring writes
#endif
#ifndef KS_ITEMS
#define KS_ITEMS 8          // base positions per lane and round of the k-mer source (experiment builds: 4, 12)
#endif
#ifndef RP_HELD
#define RP_HELD 1           // the fused first level holds two store requests per lane back (see rp_scatter_kernel)
#endif

#ifdef RP_PHASES              // experiment builds only: cycles of wave RP_PHASES of every workgroup between the marks of the scatter kernel
__device__ unsigned long long rp_phase_cycles[8];
#define RP_MARK(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long now__ = __builtin_readcyclecounter(); ph_t[i] += now__ - ph_last; ph_last = now__; __builtin_amdgcn_sched_barrier(0); }
#else
#define RP_MARK(i)
#endif

namespace {

constexpr int RP_THREADS = 1024;
constexpr int RP_MAXBITS = 11;
constexpr int RP_SLAB_UNIT = 24576;                   // slabs are multiples of every source's tile (8 or 12 items per lane)
constexpr int HK_ITEMS = 8, HK_TILE = HK_ITEMS * RP_THREADS;      // rp_hist_kmer_kernel: positions per lane and round
constexpr uint64_t RP_PHANTOM = 1ull << 63;           // placeholder for the slots before a bucket's first key
constexpr size_t RP_CACHE_BYTES = 0;                  // LDS scratch handed to the key sources (none needs it now)

// Shapes of the scatter kernel: the flush granule LINE (keys) and the number of buckets MAXB the tables are sized for.
//   <16, 1024>  whole 128-byte lines: what HBM rewards, for a level that is bound by its bytes (level 2: 96 GB)
//   < 8, 1024>  64-byte granules: the fused level 1 writes half the bytes of level 2 in the same time — it is bound by
//               its LDS rounds, not by HBM — and with 8-key granules a bucket carries 3.5 keys instead of 7.5 from
//               round to round, which is a third of the round's LDS traffic (21.3 -> 19.5 ms per 6e9 k-mers; 32-byte
//               granules: 47 ms, HBM does not take them; level 2 with 64-byte granules: 9.7 -> 12.7 ms per 3e9 keys)
//   < 8, 2048>  11-bit digits: the carried keys (< LINE per bucket) of 2048 buckets have to fit LDS
template <int LINE_, int MAXB_>
struct rp_cfg {
  static constexpr int LINE = LINE_;
  static constexpr int LOG_LINE = LINE_ == 16 ? 4 : LINE_ == 8 ? 3 : 2;
  static constexpr int MAXB = MAXB_;
  static constexpr int NBT = MAXB / RP_THREADS;              // buckets owned by one lane
  static constexpr int CARRY = LINE_ - 1;                    // most keys a bucket carries into the next round
  static constexpr int STAGE = MAXB_ == 1024 ? 16384 : 15360;  // keys staged in LDS
  // LDS carve-up (dynamic, 16-byte aligned pieces)
  static constexpr size_t OFF_META = (size_t)STAGE * 8;
  static constexpr size_t OFF_CNT = OFF_META + (size_t)MAXB * 8;
  static constexpr size_t OFF_LINE = OFF_CNT + (size_t)MAXB * 4;
  static constexpr size_t OFF_WSUM = OFF_LINE + (size_t)MAXB * 4;
  static constexpr size_t OFF_SLAB = OFF_WSUM + 32 * 4;
  static constexpr size_t OFF_CACHE = OFF_SLAB + 8 * 8;
  static constexpr size_t LDS = OFF_CACHE + RP_CACHE_BYTES;
  static constexpr size_t OFF_OVF = LDS;                       // claiming form: {keys that fit the bucket, bag position} per bucket
  static constexpr size_t LDS_CLAIM = OFF_OVF + (size_t)MAXB * 8;
};
constexpr int RP_HIST_BINS = 1 << RP_MAXBITS;
constexpr size_t RP_HIST_LDS = (size_t)RP_HIST_BINS * 4 + 8 * 8 + RP_CACHE_BYTES;

struct slab_t {
  int64_t seg;           // the parent segment
  int64_t lo, hi;        // key range of the slab (inside one parent segment)
  int64_t hbase;         // first histogram entry of the segment
  int64_t nsl, local;    // slabs in the segment, index of this one
};

// Slab s of the launch -> its segment and key range.  seg_slabs[p] = number of slabs before segment p.
__device__ __forceinline__ bool find_slab(const int64_t* __restrict__ seg_off, const int64_t* __restrict__ seg_slabs,
                                          int64_t n_seg, int64_t slab_keys, int B, int64_t* sh, slab_t& sl) {
  if (threadIdx.x == 0) {
    const int64_t s = blockIdx.x;
    if (s >= seg_slabs[n_seg]) {
      sh[0] = -1;
    } else {
      int64_t lo = 0, hi = n_seg - 1;                  // last p with seg_slabs[p] <= s (skips empty segments)
      while (lo < hi) {
        int64_t mid = lo + ((hi - lo + 1) >> 1);
        if (seg_slabs[mid] <= s) lo = mid; else hi = mid - 1;
      }
      const int64_t first = seg_slabs[lo];
      sh[0] = lo;
      sh[1] = first;
      sh[2] = seg_slabs[lo + 1] - first;
      sh[3] = seg_off[lo];
      sh[4] = seg_off[lo + 1];
    }
  }
  __syncthreads();
  if (sh[0] < 0) return false;
  sl.seg = sh[0];
  sl.local = (int64_t)blockIdx.x - sh[1];
  sl.nsl = sh[2];
  sl.hbase = sh[1] * B;
  sl.lo = sh[3] + sl.local * slab_keys;
  sl.hi = min(sl.lo + slab_keys, sh[4]);
  return true;
}

// ---- key sources ---------------------------------------------------------------------------------------------
// A tile is the item range [t0, t0 + T) ∩ [.., hi), T = items * RP_THREADS <= ITEMS * RP_THREADS.  issue() starts the global
// loads of a whole full tile into registers (no use of the values, so nothing waits), finish() turns the
// first T items' worth of them into keys: k[q] for every bit q of the returned mask.  The partition kernels issue two tiles ahead of the one they
// finish, so a full round of LDS work hides the HBM latency.
struct mem_source {
  static constexpr int ITEMS = 8;                      // keys per lane and round
  const uint64_t* __restrict__ keys;
  struct raw_t { uint64_t v[ITEMS]; };
  template <bool STREAM = false>
  __device__ __forceinline__ void issue(int64_t t0, int64_t hi, raw_t& raw) const {
#pragma unroll
    for (int q = 0; q < ITEMS; ++q) {
      const int64_t i = t0 + threadIdx.x + (int64_t)q * RP_THREADS;
      if (i < hi) raw.v[q] = STREAM ? __builtin_nontemporal_load(&keys[i]) : keys[i];   // STREAM: last use of the keys
    }
  }
  // the same loads, all ITEMS of them by every lane (indices clamped to the slab's last key: finish() ignores what lies behind
  // hi): the claiming scatter wants to know HOW MANY vector memory instructions were issued behind its claims
  __device__ __forceinline__ void issue_all(int64_t t0, int64_t hi, raw_t& raw) const {
#pragma unroll
    for (int q = 0; q < ITEMS; ++q) {
      const int64_t i = min(t0 + threadIdx.x + (int64_t)q * RP_THREADS, hi - 1);
      raw.v[q] = __builtin_nontemporal_load(&keys[i]);
    }
  }
  __device__ __forceinline__ static void landed(const raw_t&) {}
  __device__ __forceinline__ unsigned finish(int64_t t0, int64_t hi, int items, const raw_t& raw, uint64_t k[ITEMS]) const {
    unsigned vm = 0;
#pragma unroll
    for (int q = 0; q < ITEMS; ++q) {
      const int64_t i = t0 + threadIdx.x + (int64_t)q * RP_THREADS;
      if (q < items && i < hi) { k[q] = raw.v[q]; vm |= 1u << q; }
    }
    return vm;
  }
  // only the digit (key >> shift) & mask of every item: what the histogram pass needs
  __device__ __forceinline__ unsigned digits(int64_t t0, int64_t hi, const raw_t& raw, int shift, unsigned mask,
                                             unsigned d[ITEMS]) const {
    uint64_t k[ITEMS];
    const unsigned vm = finish(t0, hi, ITEMS, raw, k);
#pragma unroll
    for (int q = 0; q < ITEMS; ++q) d[q] = (unsigned)(k[q] >> shift) & mask;
    return vm;
  }
};

// The k-mer hashes of the ragged read set, generated on the fly from the packed 2-bit reads (A8).  Items are the
// flat BASE positions of the packed stream, not output indices: a bit mask (one bit per base, bnpk_kmer_start_mask)
// says at which positions a k-mer starts, so the generation needs no row lookup at all — four coalesced word
// loads per lane, one funnel shift for the first k-mer and a 2-bit roll for each of the next seven.  A lane owns
// eight consecutive positions; a tile of T positions yields at most T keys.
// CANON: every hash is replaced by min(h, hash of the reverse complement k-mer) — strand-independent k-mers.
template <bool CANON>
struct kmer_source {
  // base positions per lane and round.  (Twelve fit the staging area too — a fifth of the positions start no k-mer and
  // 64-byte granules halve what a bucket carries — and were measured: 20.1 instead of 19.5 ms, the rounds get longer
  // faster than they get fewer.)
  static constexpr int ITEMS = KS_ITEMS;
  const uint64_t* __restrict__ W;          // 2 bits per base
  const uint8_t* __restrict__ V;           // 1 bit per base: a k-mer starts here
  int64_t n_words;                         // words of W
  int k;
  struct raw_t { uint64_t w0, w1, w2; unsigned v; };
  template <bool STREAM = false>
  __device__ __forceinline__ void issue(int64_t t0, int64_t hi, raw_t& raw) const {
    const int64_t o = t0 + (int64_t)threadIdx.x * ITEMS;
    if (o < hi) {
      const int64_t wi = o >> 5;
      raw.w0 = W[wi];
      raw.w1 = W[wi + 1];
      raw.w2 = wi + 2 < n_words ? W[wi + 2] : 0;
      if (ITEMS == 8) {
        raw.v = V[o >> 3];
      } else {                                               // ITEMS mask bits from bit o of the mask (o a multiple of 4): two bytes
        const unsigned two = (unsigned)V[o >> 3] | ((unsigned)V[(o >> 3) + 1] << 8);
        raw.v = (two >> (int)(o & 7)) & ((1u << ITEMS) - 1u);
      }
    }
  }
  // the 64 bits at bit offset sh (< 128) of the 192-bit window
  __device__ __forceinline__ static uint64_t window(const raw_t& raw, int sh) {
    const uint64_t lo = sh < 64 ? raw.w0 : raw.w1, hi = sh < 64 ? raw.w1 : raw.w2;
    const int s6 = sh & 63;
    return s6 ? (lo >> s6) | (hi << (64 - s6)) : lo;
  }
  // Every lane waits for its loads HERE, whether or not it has k-mers in the tile: a wait that only some paths contain
  // makes the compiler repeat it (as vmcnt(0), behind whatever stores were issued since) where the registers are reused.
  __device__ __forceinline__ static void landed(const raw_t& raw) { asm volatile("" : : "v"(raw.v), "v"(raw.w2)); }
  // The k-mer at the lane's position o + q is bits [2q, 2q + 2k) of the 96 bits of the stream that start at base o: three
  // 32-bit words cut out of the lane's five with one V_ALIGNBIT each, then two V_ALIGNBITs by a constant and the mask
  // per k-mer — nothing is rolled from one k-mer to the next.
  __device__ __forceinline__ unsigned finish(int64_t t0, int64_t hi, int items, const raw_t& raw, uint64_t kk[ITEMS]) const {
    static_assert(2 * (ITEMS - 1) + 62 <= 96, "the k-mers of a lane lie in 96 bits of the stream");
    const int64_t o = t0 + (int64_t)threadIdx.x * ITEMS;
    const int64_t end = min(t0 + (int64_t)items * RP_THREADS, hi);
    if (o >= end) return 0;
    unsigned valid = raw.v;
    if (end - o < ITEMS) valid &= (1u << (int)(end - o)) - 1u;
    if (valid == 0) return 0;
    const unsigned sb = 2u * (unsigned)(o & 31);
    const bool up = sb >= 32u;
    const unsigned d1 = (unsigned)(raw.w0 >> 32), d2 = (unsigned)raw.w1, d3 = (unsigned)(raw.w1 >> 32);
    const unsigned e0 = up ? d1 : (unsigned)raw.w0, e1 = up ? d2 : d1, e2 = up ? d3 : d2, e3 = up ? (unsigned)raw.w2 : d3;
    const unsigned x0 = __builtin_amdgcn_alignbit(e1, e0, sb & 31u), x1 = __builtin_amdgcn_alignbit(e2, e1, sb & 31u),
                   x2 = __builtin_amdgcn_alignbit(e3, e2, sb & 31u);
    const uint64_t mask = (1ull << (2 * k)) - 1ull;
    const unsigned mask_lo = (unsigned)mask, mask_hi = (unsigned)(mask >> 32);
#pragma unroll
    for (int q = 0; q < ITEMS; ++q) {
      const unsigned lo = q ? __builtin_amdgcn_alignbit(x1, x0, 2 * q) : x0, hh = q ? __builtin_amdgcn_alignbit(x2, x1, 2 * q) : x1;
      const uint64_t h = ((uint64_t)(hh & mask_hi) << 32) | (lo & mask_lo);
      kk[q] = CANON ? min(h, ~(reverse_2bit_groups(h) >> (64 - 2 * k)) & mask) : h;   // slots of invalid positions are never read
    }
    return valid;
  }
  // The histogram pass only needs the digit.  For plain hashes bits [shift, shift + bits) of the k-mer at position p
  // are bits [2p + shift, ..) of the packed stream itself: one 64-bit window serves the lane's eight positions and
  // nothing is rolled.  (Canonical hashes have to be built.)
  __device__ __forceinline__ unsigned digits(int64_t t0, int64_t hi, const raw_t& raw, int shift, unsigned dmask,
                                             unsigned d[ITEMS]) const {
    if (CANON) {
      uint64_t kk[ITEMS];
      const unsigned vm = finish(t0, hi, ITEMS, raw, kk);
#pragma unroll
      for (int q = 0; q < ITEMS; ++q) d[q] = (unsigned)(kk[q] >> shift) & dmask;
      return vm;
    }
    const int64_t o = t0 + (int64_t)threadIdx.x * ITEMS;
    const int64_t end = min(t0 + (int64_t)(ITEMS * RP_THREADS), hi);
    if (o >= end) {                                          // (the caller adds zeros: spread them over the bins, not all on bin 0)
#pragma unroll
      for (int q = 0; q < ITEMS; ++q) d[q] = (threadIdx.x + q) & dmask;
      return 0;
    }
    unsigned valid = raw.v;
    if (end - o < ITEMS) valid &= (1u << (int)(end - o)) - 1u;
    const uint64_t win = window(raw, 2 * (int)(o & 31) + shift);       // (< 128: shift <= 2k - bits <= 61)
#pragma unroll
    for (int q = 0; q < ITEMS; ++q) d[q] = (unsigned)(win >> (2 * q)) & dmask;
    return valid;
  }
};

// (k-mer, row) pairs for the k-mer index (round 6): the first level of a partition that carries a payload.  The pair's sort key
// is 2k + row bits wide — more than a word — but inside one bucket of the first level the top `bits` bits of the k-mer are the
// same for every key, so what is WRITTEN is  word = (k-mer's remaining bits : row : tag)  — a complete sort key of its bucket in
// at most 63 bits — while the DIGIT that ranks it comes from the k-mer itself.  Only rp_ring_kernel can do that: its flush
// never looks at a key's digit again (rp_scatter_kernel re-derives the bucket from the staged key).  tag = parity of the bucket's
// rank among the non-empty buckets: the bit flips at every bucket boundary of the (compacted, de-duplicated) output, which is how
// the k-mer's top bits are put back afterwards (sparse.hip).
struct pair_source {
  static constexpr int ITEMS = 8;
  const uint64_t* __restrict__ keys;       // k-mers, < 2^key_bits
  const int64_t* __restrict__ rows;        // their rows, < 2^row_bits
  const unsigned* __restrict__ tags;       // one bit per first-level bucket (written between the histogram and the scatter)
  int dshift;                              // k-mer >> dshift = first-level digit
  int row_bits;
  struct raw_t { uint64_t v[ITEMS]; unsigned r[ITEMS]; };
  template <bool STREAM = false>
  __device__ __forceinline__ void issue(int64_t t0, int64_t hi, raw_t& raw) const {
#pragma unroll
    for (int q = 0; q < ITEMS; ++q) {
      const int64_t i = t0 + threadIdx.x + (int64_t)q * RP_THREADS;
      if (i < hi) {
        raw.v[q] = keys[i];
        raw.r[q] = (unsigned)rows[i];
      }
    }
  }
  __device__ __forceinline__ static void landed(const raw_t&) {}
  __device__ __forceinline__ unsigned digits(int64_t t0, int64_t hi, const raw_t& raw, int, unsigned mask, unsigned d[ITEMS]) const {
    unsigned vm = 0;
#pragma unroll
    for (int q = 0; q < ITEMS; ++q) {
      const int64_t i = t0 + threadIdx.x + (int64_t)q * RP_THREADS;
      d[q] = (threadIdx.x + q) & mask;                        // (the caller adds zeros for positions past the end: spread them)
      if (i < hi) { d[q] = (unsigned)(raw.v[q] >> dshift) & mask; vm |= 1u << q; }
    }
    return vm;
  }
  // words and digits of a tile; lds_tags: the tag bits, staged in LDS by the kernel
  __device__ __forceinline__ unsigned finish_split(int64_t t0, int64_t hi, int items, const raw_t& raw, uint64_t k[ITEMS],
                                                   unsigned dg[ITEMS], const unsigned* lds_tags) const {
    const uint64_t low_mask = (1ull << dshift) - 1ull;
    unsigned vm = 0;
#pragma unroll
    for (int q = 0; q < ITEMS; ++q) {
      const int64_t i = t0 + threadIdx.x + (int64_t)q * RP_THREADS;
      dg[q] = 0;
      if (q < items && i < hi) {
        const unsigned d = (unsigned)(raw.v[q] >> dshift);
        const unsigned tag = (lds_tags[d >> 5] >> (d & 31u)) & 1u;
        k[q] = (((((raw.v[q] & low_mask) << row_bits) | (uint64_t)raw.r[q]) << 1) | tag);
        dg[q] = d;
        vm |= 1u << q;
      }
    }
    return vm;
  }
};
template <typename S> struct split_digit : std::false_type {};
template <> struct split_digit<pair_source> : std::true_type {};

// ---- pass 1 of a level: digit counts per slab ------------------------------------------------------------------
template <typename Source>
__global__ __launch_bounds__(RP_THREADS) void rp_hist_kernel(Source src, const int64_t* __restrict__ seg_off,
                                                             const int64_t* __restrict__ seg_slabs, int64_t n_seg,
                                                             int64_t slab_keys, int shift, int bits,
                                                             int64_t* __restrict__ H) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned* h = reinterpret_cast<unsigned*>(smem);
  int64_t* sh = reinterpret_cast<int64_t*>(smem + (size_t)RP_HIST_BINS * 4);
  const int B = 1 << bits;
  slab_t sl;
  if (!find_slab(seg_off, seg_slabs, n_seg, slab_keys, B, sh, sl)) return;
  for (int c = threadIdx.x; c < B; c += RP_THREADS) h[c] = 0;
  __syncthreads();
  typename Source::raw_t raw, raw_next;
  src.issue(sl.lo, sl.hi, raw);
  for (int64_t t0 = sl.lo; t0 < sl.hi; t0 += (Source::ITEMS * RP_THREADS)) {
    if (t0 + (Source::ITEMS * RP_THREADS) < sl.hi) src.issue(t0 + (Source::ITEMS * RP_THREADS), sl.hi, raw_next);      // one tile ahead
    unsigned d[Source::ITEMS] = {};
    const unsigned vm = src.digits(t0, sl.hi, raw, shift, (unsigned)(B - 1), d);
    // branch-free: a position where no k-mer starts adds zero (one field extract instead of an exec-mask round trip per key)
#pragma unroll
    for (int q = 0; q < Source::ITEMS; ++q) atomicAdd(&h[d[q]], (vm >> q) & 1u);
    raw = raw_next;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < B; c += RP_THREADS) H[sl.hbase + (int64_t)c * sl.nsl + sl.local] = h[c];
}

// The level-1 counts of plain (not canonical) k-mer hashes.  Bits [shift, shift + bits) of the k-mer at base p are bits
// [2p + shift, ..) of the packed stream, so a lane's eight digits lie in the 25 bits behind bit 2 * o + shift of it: two
// 32-bit words, one V_ALIGNBIT, one field extract per digit.  The bit offset inside the word is the same in every round (a
// round advances all lanes by 16384 bits), the addresses are a scalar base plus a per-lane offset, and positions where no
// k-mer starts add zero instead of being branched around: ~4 vector instructions per k-mer, where the generic kernel
// above (64-bit indices, three words, a 64-bit funnel shift) spent 15 and was bound by them.
__global__ __launch_bounds__(RP_THREADS) void rp_hist_kmer_kernel(const uint32_t* __restrict__ W32, const uint8_t* __restrict__ V,
                                                                  const int64_t* __restrict__ seg_off, const int64_t* __restrict__ seg_slabs,
                                                                  int64_t n_seg, int64_t slab_keys, int shift, int bits,
                                                                  int64_t* __restrict__ H) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned* h = reinterpret_cast<unsigned*>(smem);
  int64_t* sh = reinterpret_cast<int64_t*>(smem + (size_t)RP_HIST_BINS * 4);
  const int B = 1 << bits;
  const unsigned dmask = (unsigned)(B - 1);
  slab_t sl;
  if (!find_slab(seg_off, seg_slabs, n_seg, slab_keys, B, sh, sl)) return;
  for (int c = threadIdx.x; c < B; c += RP_THREADS) h[c] = 0;
  __syncthreads();
  // slab-relative 32-bit arithmetic: position r = round * HK_TILE + 8 * tid of [0, len)
  const unsigned len = (unsigned)(sl.hi - sl.lo);
  const int64_t bit0 = 2 * (sl.lo + (int64_t)threadIdx.x * HK_ITEMS) + shift;        // of this lane's round-0 window
  const uint32_t* wp = W32 + (bit0 >> 5);
  const uint8_t* vp = V + ((sl.lo >> 3) + threadIdx.x);
  const unsigned sh5 = (unsigned)(bit0 & 31);
  constexpr unsigned W_STEP = HK_TILE * 2 / 32, V_STEP = HK_TILE / 8;                       // per round: words of the stream, bytes of the mask
  unsigned r = threadIdx.x * HK_ITEMS;
  uint2 w = make_uint2(0, 0);
  unsigned v = 0;
  if (r < len) {
    w = *reinterpret_cast<const uint2*>(wp);
    v = *vp;
  }
  while (r < len) {                                          // (per lane; a lane that ran out just waits at the barrier below)
    const uint2 w_cur = w;
    unsigned valid = v;
    const unsigned r_next = r + HK_TILE;
    wp += W_STEP;
    vp += V_STEP;
    if (r_next < len) {                                      // one round ahead
      w = *reinterpret_cast<const uint2*>(wp);
      v = *vp;
    }
    if (len - r < HK_ITEMS) valid &= (1u << (len - r)) - 1u;
    const unsigned win = __builtin_amdgcn_alignbit(w_cur.y, w_cur.x, sh5);
#pragma unroll
    for (int q = 0; q < HK_ITEMS; ++q) atomicAdd(&h[(win >> (2 * q)) & dmask], (valid >> q) & 1u);
    r = r_next;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < B; c += RP_THREADS) H[sl.hbase + (int64_t)c * sl.nsl + sl.local] = h[c];
}

// The same counts for keys that lie in memory, with 16-byte loads (two keys per lane and request: the 8-byte version reads
// at 5.5 TB/s, this one nearer the rate a pure read stream reaches) and the next tile's requests in flight while the
// current tile is counted.  A slab that starts or ends on an odd key has that key counted by lane 0.
__global__ __launch_bounds__(RP_THREADS) void rp_hist_mem_kernel(const uint64_t* __restrict__ keys, const int64_t* __restrict__ seg_off,
                                                                 const int64_t* __restrict__ seg_slabs, int64_t n_seg,
                                                                 int64_t slab_keys, int shift, int bits, int64_t* __restrict__ H) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned* h = reinterpret_cast<unsigned*>(smem);
  int64_t* sh = reinterpret_cast<int64_t*>(smem + (size_t)RP_HIST_BINS * 4);
  const int B = 1 << bits;
  const unsigned dmask = (unsigned)(B - 1);
  slab_t sl;
  if (!find_slab(seg_off, seg_slabs, n_seg, slab_keys, B, sh, sl)) return;
  for (int c = threadIdx.x; c < B; c += RP_THREADS) h[c] = 0;
  __syncthreads();
  int64_t lo = sl.lo, hi = sl.hi;
  if (lo < hi && (lo & 1)) {
    if (threadIdx.x == 0) atomicAdd(&h[(unsigned)(keys[lo] >> shift) & dmask], 1u);
    ++lo;
  }
  if (lo < hi && ((hi - lo) & 1)) {
    --hi;
    if (threadIdx.x == 0) atomicAdd(&h[(unsigned)(keys[hi] >> shift) & dmask], 1u);
  }
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  const u64x2* pairs = reinterpret_cast<const u64x2*>(keys + lo);
  const int64_t n_pairs = (hi - lo) >> 1;
  constexpr int PER = mem_source::ITEMS / 2;                       // pairs per lane and tile
  u64x2 cur[PER], nxt[PER];
  auto issue = [&](int64_t base, u64x2 v[PER]) {
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int64_t i = base + threadIdx.x + (int64_t)q * RP_THREADS;
      if (i < n_pairs) v[q] = __builtin_nontemporal_load(pairs + i);
    }
  };
  issue(0, cur);
  for (int64_t base = 0; base < n_pairs; base += (int64_t)PER * RP_THREADS) {
    if (base + (int64_t)PER * RP_THREADS < n_pairs) issue(base + (int64_t)PER * RP_THREADS, nxt);
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int64_t i = base + threadIdx.x + (int64_t)q * RP_THREADS;
      if (i < n_pairs) {
        atomicAdd(&h[(unsigned)(cur[q].x >> shift) & dmask], 1u);
        atomicAdd(&h[(unsigned)(cur[q].y >> shift) & dmask], 1u);
      }
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) cur[q] = nxt[q];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < B; c += RP_THREADS) H[sl.hbase + (int64_t)c * sl.nsl + sl.local] = h[c];
}

// ---- pass 2 of a level: write-combining scatter ------------------------------------------------------------------
// One round = one tile of new keys merged with the keys carried over from the previous round:
//   rank   every new key takes a rank inside its bucket from an LDS counter (done right after the tile is loaded,
//          i.e. at the end of the previous round, so the loads / the k-mer generation overlap the store drain)
//   layout per bucket: nfl = keys that complete whole lines, the rest is carried; one packed scan gives every
//          bucket a slice of the FLUSH region (a multiple of LINE keys, line-aligned in LDS) and a slice of the
//          CARRY region behind it
//   stage  carried + new keys are written to their slices; every flush slice is rotated by a bucket-dependent even
//          offset, otherwise all slices start on LDS bank 0 and equal ranks collide 16-way
//   flush  the FLUSH region leaves the CU as aligned 16-byte-per-lane stores (whole lines only); the CARRY region is
//          read back into the owning lanes' registers
// CLAIM (a level that reads its keys from memory, whole lines, 1024 buckets): no histogram pass ran before this kernel and
// nothing says where a (slab, bucket) run belongs.  Every child bucket owns RPC_STRIDE slots of `out` — lines at the front,
// [0, RPC_CAP_LO), the < 16 leftover keys of every slab behind them, [RPC_CAP_LO, RPC_STRIDE) — and a workgroup CLAIMS the
// place of the lines it flushes for a bucket in a round with one returning atomicAdd on the bucket's fill counter (fill[2c];
// the leftovers of its last round on fill[2c + 1]).  The claims are multiples of a line, so every line stays 128-byte aligned;
// a bucket is two dense runs and no holes: [0, min(fill[2c], CAP_LO)) and [CAP_LO, CAP_LO + min(fill[2c+1], TAIL)).  What does
// not fit a bucket's slots (a bucket that would be over the finishing kernels' capacity anyway; more slabs per segment than the
// tail has room for) goes to an unordered BAG of keys that the caller counts on its own and adds to the result.
constexpr int RPC_CAP_LO = 7552, RPC_TAIL = 128, RPC_STRIDE = RPC_CAP_LO + RPC_TAIL;      // = the fast finishing kernels' 7680
struct rp_claim_t {
  unsigned* fill;                  // [2 * buckets] {keys claimed at the front, keys claimed in the tail}
  uint64_t* bag;                   // keys without a place in their bucket
  unsigned long long* bag_fill;    // claimed slots of the bag (may run past bag_cap: the caller looks)
  int64_t bag_cap;
};

template <typename Source, int LINE, int MAXB, bool CLAIM = false>
__global__ __launch_bounds__(RP_THREADS) void rp_scatter_kernel(Source src, const int64_t* __restrict__ seg_off,
                                                                const int64_t* __restrict__ seg_slabs, int64_t n_seg,
                                                                int64_t slab_keys, int shift, int bits,
                                                                const int64_t* __restrict__ offs,
                                                                uint64_t* __restrict__ out, rp_claim_t claim) {
  using C = rp_cfg<LINE, MAXB>;
  static_assert(!CLAIM || (std::is_same<Source, mem_source>::value && LINE == 16 && MAXB == 1024), "the claiming form is level 2's");
  // Two schedules of a round (see the loop): the fused first level, which is bound by its rounds, takes the next tile's
  // keys before it stores and has the loads of the tile after next in flight for a whole round; a level that reads its
  // keys from memory is bound by HBM either way and keeps the shorter-lived registers of the plain order.
  constexpr bool AHEAD = !std::is_same<Source, mem_source>::value;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* stage = reinterpret_cast<uint64_t*>(smem);
  uint64_t* meta = reinterpret_cast<uint64_t*>(smem + C::OFF_META);   // {flush start:16 | nfl:16 | carry start - nfl:16}
  unsigned* newcnt = reinterpret_cast<unsigned*>(smem + C::OFF_CNT);  // keys of the bucket in this round, the carried ones included
  unsigned* line = reinterpret_cast<unsigned*>(smem + C::OFF_LINE);   // write cursor of the bucket / LINE
  unsigned* wsum = reinterpret_cast<unsigned*>(smem + C::OFF_WSUM);
  int64_t* sh = reinterpret_cast<int64_t*>(smem + C::OFF_SLAB);
  uint64_t* ovf = reinterpret_cast<uint64_t*>(smem + C::OFF_OVF);     // (claiming form) {lo keys that fit:16 | tail keys that fit:8 | bag position:40}
  const int B = 1 << bits;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  slab_t sl;
  if (!find_slab(seg_off, seg_slabs, n_seg, slab_keys, B, sh, sl)) return;
  if (sl.lo >= sl.hi) return;
  constexpr uint64_t NO_BAG = (1ull << 40) - 1;              // bag position of keys that found no room in the bag either
  auto tile_size = [](unsigned carried) {
    return min((unsigned)(Source::ITEMS * RP_THREADS), ((unsigned)C::STAGE - carried) & ~(unsigned)(RP_THREADS - 1));   // >= RP_THREADS
  };

  // A lane owns buckets tid, tid + 1024, ..: their write cursors (kept LINE-aligned; the slots between the aligned
  // cursor and the bucket's true first position are phantom keys that are staged like real ones but never
  // stored), the number of carried keys and those keys themselves.
  int64_t cursor[C::NBT];
  unsigned rem[C::NBT];
  uint64_t left[C::NBT][C::CARRY];
  unsigned carried = 0;
#pragma unroll
  for (int b = 0; b < C::NBT; ++b) {
    const int d = tid + b * RP_THREADS;
    cursor[b] = 0;
    rem[b] = 0;
    if (d < B) {
      const int64_t c0 = CLAIM ? 0 : offs[sl.hbase + (int64_t)d * sl.nsl + sl.local];
      cursor[b] = c0 & ~(int64_t)(LINE - 1);
      rem[b] = (unsigned)(c0 & (LINE - 1));
      newcnt[d] = rem[b];
    }
    carried += rem[b];
#pragma unroll
    for (int j = 0; j < C::CARRY; ++j) left[b][j] = RP_PHANTOM | ((uint64_t)d << shift);
  }
  carried = wave_sum(carried);
  if (lane == 0) wsum[wave] = carried;
  __syncthreads();
  carried = 0;
#pragma unroll
  for (int w = 0; w < RP_THREADS / 64; ++w) carried += wsum[w];
  __syncthreads();

  int64_t t0 = sl.lo;
  unsigned T = tile_size(carried);
  typename Source::raw_t raw;                     // loads of the next tile, in flight while this one is staged + flushed
  src.template issue<true>(t0, sl.hi, raw);
  // A key's rank is its index among the bucket's keys of the round: the counter starts at the number of carried keys.
  // Branch-free and back to back (a position where no key starts adds zero to some bucket), so the eight LDS round trips
  // of a lane overlap.
  uint64_t k[Source::ITEMS] = {};
  unsigned r[Source::ITEMS];
  unsigned vm;
  auto digit = [&](uint64_t key) { return (unsigned)(key >> shift) & (unsigned)(B - 1); };
  auto rank_tile = [&]() {
    Source::landed(raw);
    vm = src.finish(t0, sl.hi, (int)(T / RP_THREADS), raw, k);
#pragma unroll
    for (int q = 0; q < Source::ITEMS; ++q) r[q] = (RP_ABL & 4) ? (unsigned)(tid & 7) : atomicAdd(&newcnt[digit(k[q])], (vm >> q) & 1u);
  };
  rank_tile();
  if (AHEAD) src.template issue<true>(t0 + T, sl.hi, raw);   // (one round ahead: consumed at the end of the first round)
  __syncthreads();

  // A CU drains its stores at ~20 GB/s however many of the other CUs are storing (that times 256 is what HBM takes), and a
  // round's 52 KB issued in one burst stall the wavefronts that issue them for 45 % of the round (measured with cycle
  // counters: flush + the barrier behind it), with nothing else to run in this one-workgroup-per-CU kernel.  So the level
  // that is bound by its rounds and not by HBM (the fused first one) holds two of a lane's pairs back in registers and
  // issues them later: one after the carried keys are read back, one after the next round's carried keys are staged —
  // but none in the phase before the wait for the next tile's loads, which would sit through the stores' round trip.
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  constexpr bool HELD = RP_HELD && AHEAD && MAXB == 1024;
  uint64_t* held_dst[2] = {nullptr, nullptr};
  u64x2 held_val[2];
  auto emit_held = [&](int h) {
    if (HELD && held_dst[h]) {
      __builtin_nontemporal_store(held_val[h], reinterpret_cast<u64x2*>(held_dst[h]));
      held_dst[h] = nullptr;
    }
  };
#ifdef RP_PHASES
  unsigned long long ph_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_last = __builtin_readcyclecounter();
#endif
  while (true) {
    const bool last = t0 + T >= sl.hi;
    // layout of the round
    unsigned nfl[C::NBT], nrem[C::NBT], packed = 0;
    unsigned c_whole[C::NBT] = {}, c_tail[C::NBT] = {}, c_lo[C::NBT] = {}, c_hi[C::NBT] = {};   // (claiming form) what was claimed, and where
#pragma unroll
    for (int b = 0; b < C::NBT; ++b) {
      const int d = tid + b * RP_THREADS;
      unsigned tot = 0;
      if (d < B) tot = newcnt[d];
      nfl[b] = last ? tot : (tot & ~(unsigned)(LINE - 1));     // whole lines only, except in the slab's last round
      nrem[b] = tot - nfl[b];
      if (d < B) newcnt[d] = nrem[b];                           // where the next round's ranks start
      packed += nfl[b] | (nrem[b] << 16);
      if (CLAIM && d < B) {
        const int64_t c = sl.seg * B + d;
        c_whole[b] = nfl[b] & ~(unsigned)(LINE - 1);           // lines go to the front of the bucket's slots ...
        c_tail[b] = nfl[b] - c_whole[b];                       // ... the slab's last < 16 keys to its tail
        if (RP_ABL & 32) {                                     // (experiment: the atomics fire and forget, the places are made up)
          if (c_whole[b]) atomicAdd(&claim.fill[2 * c], c_whole[b]);
          c_lo[b] = (unsigned)((cursor[b] + sl.local * 1600) % 6000) & ~15u;
          cursor[b] += c_whole[b];
        } else if (RP_ABL & 64) {                              // (experiment: no atomics at all)
          c_lo[b] = (unsigned)((cursor[b] + sl.local * 1600) % 6000) & ~15u;
          cursor[b] += c_whole[b];
        } else {
          if (c_whole[b]) c_lo[b] = atomicAdd(&claim.fill[2 * c], c_whole[b]);
          if (c_tail[b]) c_hi[b] = atomicAdd(&claim.fill[2 * c + 1], c_tail[b]);
        }
      }
    }
    const unsigned inc = wave_inclusive_scan(packed);
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    RP_MARK(0)
    // the sixteen wave totals: every row of sixteen lanes scans them (four DPP steps), the wave picks its own prefix
    static_assert(RP_THREADS / 64 == 16, "one DPP row holds the wave totals");
    unsigned ws = wsum[lane & 15];
    ws += (unsigned)__builtin_amdgcn_update_dpp(0, (int)ws, 0x111, 0xf, 0xf, false);
    ws += (unsigned)__builtin_amdgcn_update_dpp(0, (int)ws, 0x112, 0xf, 0xf, false);
    ws += (unsigned)__builtin_amdgcn_update_dpp(0, (int)ws, 0x114, 0xf, 0xf, false);
    ws += (unsigned)__builtin_amdgcn_update_dpp(0, (int)ws, 0x118, 0xf, 0xf, false);
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)ws, 15);
    const unsigned wbase = wave_s ? (unsigned)__builtin_amdgcn_readlane((int)ws, wave_s - 1) : 0u;
    const unsigned T_next = tile_size(total >> 16);           // the next tile's extent is known
    unsigned ex = wbase + inc - packed;
    const unsigned total_f = total & 0xffffu;
    unsigned cpos[C::NBT];
#pragma unroll
    for (int b = 0; b < C::NBT; ++b) {
      const int d = tid + b * RP_THREADS;
      const unsigned fpos = ex & 0xffffu;
      cpos[b] = total_f + (ex >> 16);
      ex += nfl[b] | (nrem[b] << 16);
      if (d < B) {
        meta[d] = (uint64_t)(fpos | (nfl[b] << 16)) | ((uint64_t)(cpos[b] - nfl[b]) << 32);
        if (!CLAIM) line[d] = (unsigned)(cursor[b] >> C::LOG_LINE);
        // carried keys precede the new ones (rem < LINE <= nfl): in-bucket indices 0 .. rem-1.  Index j goes to slot
        // (j + rot) mod nfl of the flush slice — it wraps at most once, from index nfl - rot on — or, when the bucket
        // flushes nothing this round, to slot j of its carry slice.
        const unsigned rot = last ? 0u : 2u * ((unsigned)d & (unsigned)(LINE / 2 - 1));
        const unsigned p0 = nfl[b] ? fpos + rot : cpos[b];
        const unsigned wrap = nfl[b] ? nfl[b] - rot : 0xffffu;
#pragma unroll
        for (int j = 0; j < C::CARRY; ++j)
          if (!(RP_ABL & 8) && j < (int)rem[b]) stage[p0 + j - ((unsigned)j >= wrap ? nfl[b] : 0u)] = left[b][j];
      }
    }
    emit_held(1);
    __syncthreads();
    RP_MARK(1)
    if (!AHEAD && !CLAIM && !last) src.template issue<true>(t0 + T, sl.hi, raw);   // plain order: the loads land while this tile is staged and flushed
    // (claiming form) the same place — BEHIND the claims of this round, which are returning atomics whose answers are needed
    // before the flush: vector memory operations complete in order, so the wait for the claims is s_waitcnt vmcnt(loads issued
    // since), and the compiler can only count them if every lane issues all eight (kernel trick of common practice: clamp, not branch)
    // — in the slab's last round too (eight loads of its last key), or the two paths would differ in that number
    if constexpr (CLAIM) src.issue_all(t0 + T, sl.hi, raw);
    // stage the new keys behind the carried ones, four at a time: the table reads first (one wait for the four of them),
    // then the arithmetic, then the stores
    constexpr int SG = MAXB > 1024 ? 2 : 4;                   // (two owned buckets per lane leave fewer registers)
    const unsigned rot_mask = last ? 0u : (unsigned)(LINE / 2 - 1);
#pragma unroll
    for (int q0 = 0; q0 < Source::ITEMS; q0 += SG) {
      uint64_t mq[SG];
#pragma unroll
      for (int u = 0; u < SG; ++u) mq[u] = meta[digit(k[q0 + u])];
#pragma unroll
      for (int u = 0; u < SG; ++u) {
        const int q = q0 + u;
        const unsigned lo = (unsigned)mq[u], hi = (unsigned)(mq[u] >> 32);
        const unsigned j = r[q], f = lo >> 16;
        unsigned jr = j + 2u * (digit(k[q]) & rot_mask);                                // slot inside the (rotated) flush slice:
        jr = min(jr, jr - f);                                                           // (j + rot) mod f, j + rot < 2 f
        const unsigned slot = j < f ? (lo & 0xffffu) + jr : hi + j;                     // hi = carry start - f
        if (!(RP_ABL & 2) && ((vm >> q) & 1u)) stage[slot] = k[q];
      }
    }
    if (CLAIM) {
      // the answers of the claims: where the bucket's lines of this round go.  line[d] = first line of the claimed place (bit 31:
      // not everything fits — ovf[d] says how much does and where the rest lies in the bag); in the slab's last round newcnt[d]
      // (free by now) = {whole-line keys:16 | leftover keys:4 | their place in the bucket's tail:12}
      // (the answers are looked at HERE and not a cycle earlier: left to itself the scheduler moves this arithmetic up behind
      // the atomics and the wait for them with it — the whole latency of a device-scope atomic, once per round: 4 of 25 ms)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < C::NBT; ++b) asm volatile("" : "+v"(c_lo[b]), "+v"(c_hi[b]));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < C::NBT; ++b) {
        const int d = tid + b * RP_THREADS;
        if (d < B) {
          const int64_t c = sl.seg * B + d;
          const unsigned fit = c_lo[b] >= (unsigned)RPC_CAP_LO ? 0u : min(c_whole[b], (unsigned)RPC_CAP_LO - c_lo[b]);
          const unsigned hb = min(c_hi[b], (unsigned)RPC_TAIL);
          const unsigned tfit = min(c_tail[b], (unsigned)RPC_TAIL - hb);
          unsigned ln = (unsigned)((c * RPC_STRIDE + (int64_t)min(c_lo[b], (unsigned)RPC_CAP_LO)) >> C::LOG_LINE);
          const unsigned spill = (c_whole[b] - fit) + (c_tail[b] - tfit);
          if (spill) {                                         // (rare: a bucket over the capacity, a segment of very many slabs)
            unsigned long long bp = atomicAdd(claim.bag_fill, (unsigned long long)spill);
            if ((int64_t)(bp + spill) > claim.bag_cap) bp = NO_BAG;
            ovf[d] = (uint64_t)fit | ((uint64_t)tfit << 16) | ((uint64_t)bp << 24);
            ln |= 0x80000000u;
          }
          line[d] = ln;
          if (last) newcnt[d] = c_whole[b] | (c_tail[b] << 16) | (hb << 20);
        }
      }
    }
    __syncthreads();
    RP_MARK(2)
    if (last) {
      for (unsigned i = tid; i < total_f; i += RP_THREADS) {
        const uint64_t key = stage[i];
        const unsigned d = (unsigned)(key >> shift) & (B - 1);
        if (key >> 63) continue;
        const unsigned j = i - ((unsigned)meta[d] & 0xffffu);  // index among the bucket's keys of this (last) round
        if (!CLAIM) {
          out[((int64_t)line[d] << C::LOG_LINE) + j] = key;
          continue;
        }
        const unsigned info = newcnt[d], whole = info & 0xffffu, hb = info >> 20, ln = line[d];
        unsigned fit = whole, tfit = (info >> 16) & 15u;
        uint64_t bp = 0;
        if (ln >> 31) { const uint64_t ov = ovf[d]; fit = (unsigned)ov & 0xffffu; tfit = (unsigned)(ov >> 16) & 0xffu; bp = ov >> 24; }
        uint64_t* dst;
        if (j < whole) dst = j < fit ? out + (((int64_t)(ln & 0x7fffffffu) << C::LOG_LINE) + j) : (bp == NO_BAG ? nullptr : claim.bag + bp + (j - fit));
        else {
          const unsigned t = j - whole;
          dst = t < tfit ? out + ((sl.seg * B + d) * (int64_t)RPC_STRIDE + RPC_CAP_LO + hb + t)
                         : (bp == NO_BAG ? nullptr : claim.bag + bp + (whole - fit) + (t - tfit));
        }
        if (dst) *dst = key;
      }
      break;
    }
    // The next tile's keys and ranks BEFORE this round's stores are issued.  Its loads are awaited with vmcnt(0) — the
    // count is unknown at compile time, and the compiler repeats the wait where it reuses the registers for the loads of
    // the tile after — and behind the stores such a wait sits through their whole round trip to HBM, once per round:
    // 8 of the 19.7 ms of the fused level 1 (without its stores: 11.8 ms).  So: wait here, a whole round after the loads
    // and the last stores were issued; start the loads of the tile after next; only then store.  The stores drain while
    // the next round is laid out and staged.
    if (AHEAD) {
      t0 += T;
      T = T_next;
      rank_tile();
      RP_MARK(3)
      src.template issue<true>(t0 + T, sl.hi, raw);
      RP_MARK(4)
    }
    // flush: whole lines, 16 bytes per lane; two independent LDS -> HBM chains per lane and call.  With HELD, the
    // second and third pair of a lane (a round has ~3.2 per lane) only get as far as the lane's registers here.
    auto flush2 = [&](unsigned i0, int hold0, int hold1) {
      ulonglong2 kk[2];
      unsigned d[2], f[2], nf[2], ln[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const unsigned i = i0 + u * 2 * RP_THREADS;
        if (i < total_f) kk[u] = *reinterpret_cast<const ulonglong2*>(stage + i);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const unsigned i = i0 + u * 2 * RP_THREADS;
        if (i < total_f) {
          d[u] = (unsigned)(kk[u].x >> shift) & (B - 1);
          const unsigned m = (unsigned)meta[d[u]];
          f[u] = m & 0xffffu;
          nf[u] = m >> 16;
          ln[u] = line[d[u]];
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const unsigned i = i0 + u * 2 * RP_THREADS;
        const int hold = u ? hold1 : hold0;
        if (i < total_f) {
          unsigned j = i - f[u] - 2u * (d[u] & (unsigned)(LINE / 2 - 1));              // undo the rotation of the slice:
          j = min(j, j + nf[u]);                                                       // (slot - rot) mod nfl
          uint64_t* dst = out + (((int64_t)(CLAIM ? ln[u] & 0x7fffffffu : ln[u]) << C::LOG_LINE) + j);
          if (CLAIM && (ln[u] >> 31)) {                         // part of the bucket's lines of this round lie in the bag
            const uint64_t ov = ovf[d[u]];
            const unsigned fit = (unsigned)ov & 0xffffu;
            if (j >= fit) {
              if ((ov >> 24) == NO_BAG) continue;              // (no room anywhere: the caller sees bag_fill > bag_cap and counts another way)
              dst = claim.bag + (ov >> 24) + (j - fit);
            }
          }
          if (RP_ABL & 16) dst = out + (size_t)blockIdx.x * 16384 + (i & 0x3ffeu);
          if (RP_ABL & 1) {
            if (dst == nullptr) dst[0] = kk[u].x;
          } else if (!((kk[u].x | kk[u].y) >> 63)) {
            u64x2 pair;
            pair.x = kk[u].x;
            pair.y = kk[u].y;
            if (hold >= 0) {
              held_dst[hold >= 0 ? hold : 0] = dst;
              held_val[hold >= 0 ? hold : 0] = pair;
            } else {
              __builtin_nontemporal_store(pair, reinterpret_cast<u64x2*>(dst));
            }
          } else {                                             // phantom slots before the bucket's first key
            if (!(kk[u].x >> 63)) dst[0] = kk[u].x;
            if (!(kk[u].y >> 63)) dst[1] = kk[u].y;
          }
        }
      }
    };
    if (HELD) {
      if (2u * tid < total_f) flush2(2 * tid, -1, 0);
      if (2u * tid + 4 * RP_THREADS < total_f) flush2(2 * tid + 4 * RP_THREADS, 1, -1);
      for (unsigned i0 = 2 * tid + 8 * RP_THREADS; i0 < total_f; i0 += 4 * RP_THREADS) flush2(i0, -1, -1);
    } else {
      for (unsigned i0 = 2 * tid; i0 < total_f; i0 += 4 * RP_THREADS) flush2(i0, -1, -1);
    }
    RP_MARK(5)
#pragma unroll
    for (int b = 0; b < C::NBT; ++b) {
      if (tid + b * RP_THREADS < B) {
        // (64-byte granules: seven unconditional reads from one address — what lies behind the nrem carried keys is
        // never staged — cost less than seven compares and branches)
#pragma unroll
        for (int j = 0; j < C::CARRY; ++j)
          if (!(RP_ABL & 8) && (LINE == 8 || j < (int)nrem[b])) left[b][j] = stage[cpos[b] + j];
        cursor[b] += nfl[b];
        rem[b] = nrem[b];
      }
    }
    emit_held(0);
    if (!AHEAD) {                                            // plain order: the next tile's keys and ranks after the stores
      t0 += T;
      T = T_next;
      rank_tile();
    }
    __syncthreads();
    RP_MARK(6)
  }
#ifdef RP_PHASES
  if (tid == 64 * RP_PHASES) for (int i = 0; i < 8; ++i) atomicAdd(&rp_phase_cycles[i], ph_t[i]);
#endif
}

// ---- pass 2 of the fused first level, second form (round 6): every bucket has ONE 128-byte line at a fixed place in LDS ----
// rp_scatter_kernel lays every round out anew: a packed scan over the buckets, a table look-up per key, the keys that do not
// complete a line carried through registers into the next round's layout — 70 vector and 11 LDS instructions per k-mer, four
// barriers per round, and what the SQ counters show is both pipes half busy and neither filling the other's gaps.  Here
// nothing is laid out.  Bucket d owns the sixteen slots ring[16 d .. 16 d + 15] for the whole slab; cnt[d] = {keys of the
// bucket that were ranked and not yet flushed : 16 | lines flushed so far : 16}.  A key takes its rank with one returning LDS
// atomic, rel = the low half of the answer, and
//     rel < 16        it is written to slot rel                                         (this round)
//     16 <= rel < 32  its slot (rel - 16) is free after this round's flush: the lane keeps it in registers and writes it
//                     first thing in the next round — the bucket's line of THIS round is complete without it
//     rel >= 32       rare (a bucket that takes more than two lines of keys in one round: skewed digits): the workgroup
//                     loops {write whoever's turn it is, flush} until nobody is left — correct for any input, fast for none
// The flush needs no list and no owner either: eight lanes look at bucket d (d = tid / 8, + 128, ...), and if it holds
// sixteen keys they read its line (16 bytes each, conflict-free), store it to line line_tab[d] of the output and the first of
// them takes 16 off the count.  Two barriers per round: ranks + writes | flush.  The next tile's words are awaited, turned into
// k-mers and the tile after next is requested BETWEEN the two (before the round's stores are issued: s_waitcnt vmcnt counts
// stores, rp_scatter_kernel's comment), so the wait covers loads and stores that are a whole round old.
// The slots in front of a bucket's first key (the bucket starts in the middle of a line that the previous slab completes)
// hold RP_PHANTOM and are never stored; a slab's last round writes every bucket's remaining < 16 keys one by one.
struct rr_cfg {
  static constexpr int MAXB = 1024;
  static constexpr size_t OFF_CNT = (size_t)MAXB * 16 * 8;
  static constexpr size_t OFF_LINE = OFF_CNT + (size_t)MAXB * 4;
  static constexpr size_t OFF_MISC = OFF_LINE + (size_t)MAXB * 4;
  static constexpr size_t OFF_TAGS = OFF_MISC + 8 * 8 + 16;    // pair_source: one bit per bucket
  static constexpr size_t LDS = OFF_TAGS + MAXB / 8;
};

template <typename Source>
__global__ __launch_bounds__(RP_THREADS) void rp_ring_kernel(Source src, const int64_t* __restrict__ seg_off,
                                                             const int64_t* __restrict__ seg_slabs, int64_t n_seg,
                                                             int64_t slab_keys, int shift, int bits,
                                                             const int64_t* __restrict__ offs, uint64_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* ring = reinterpret_cast<uint64_t*>(smem);
  unsigned* cnt = reinterpret_cast<unsigned*>(smem + rr_cfg::OFF_CNT);
  unsigned* line_tab = reinterpret_cast<unsigned*>(smem + rr_cfg::OFF_LINE);
  int64_t* sh = reinterpret_cast<int64_t*>(smem + rr_cfg::OFF_MISC);
  unsigned* flags = reinterpret_cast<unsigned*>(smem + rr_cfg::OFF_MISC + 64);
  unsigned* tagl = reinterpret_cast<unsigned*>(smem + rr_cfg::OFF_TAGS);
  const int B = 1 << bits;
  const int tid = threadIdx.x;
  slab_t sl;
  if (!find_slab(seg_off, seg_slabs, n_seg, slab_keys, B, sh, sl)) return;
  if (sl.lo >= sl.hi) return;
  if constexpr (split_digit<Source>::value) {
    if (tid < rr_cfg::MAXB / 32) tagl[tid] = tid < (B + 31) / 32 ? src.tags[tid] : 0u;
  }
  for (int i = tid; i < B * 16; i += RP_THREADS) ring[i] = RP_PHANTOM;
  for (int d = tid; d < B; d += RP_THREADS) {
    const int64_t c0 = offs[sl.hbase + (int64_t)d * sl.nsl + sl.local];
    cnt[d] = (unsigned)(c0 & 15);
    line_tab[d] = (unsigned)(c0 >> 4);
  }
  if (tid == 0) { flags[0] = 0; flags[1] = 0; }
  __syncthreads();

  constexpr int IT = Source::ITEMS;
  constexpr int64_t TILE = (int64_t)IT * RP_THREADS;
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  const unsigned dmask = (unsigned)(B - 1);
  auto digit = [&](uint64_t key) { return (unsigned)(key >> shift) & dmask; };
  // eight lanes per bucket: a bucket that holds a whole line gives it up.  The counts of four of a lane's buckets are read
  // together, then the lines of those that are full (reads under their own exec masks, nothing waited for in between), then
  // the stores: three LDS round trips per four buckets instead of two per bucket.
  auto flush_lines = [&]() {
    const int sub = tid & 7;
    constexpr int G = RP_THREADS / 8;                            // buckets looked at per pass of the workgroup
    constexpr int U = 4;
    for (int d0 = tid >> 3; d0 < B; d0 += U * G) {
      unsigned w[U], ln[U];
      ulonglong2 kk[U];
#pragma unroll
      for (int j = 0; j < U; ++j) w[j] = d0 + j * G < B ? cnt[d0 + j * G] : 0u;
#pragma unroll
      for (int j = 0; j < U; ++j) {
        if ((w[j] & 0xffffu) >= 16u && !(RR_ABL & 2)) {
          ln[j] = line_tab[d0 + j * G];
          kk[j] = *reinterpret_cast<const ulonglong2*>(ring + (d0 + j * G) * 16 + 2 * sub);
        }
      }
#pragma unroll
      for (int j = 0; j < U; ++j) {
        if ((w[j] & 0xffffu) >= 16u) {
          const int d = d0 + j * G;
          if (sub == 0) {
            cnt[d] = w[j] - 16u;
            if (!(RR_ABL & 2)) line_tab[d] = ln[j] + 1u;
          }
          if (RR_ABL & 3) continue;
          uint64_t* dst = out + (((uint64_t)ln[j] << 4) + (unsigned)(2 * sub));
          if (!((kk[j].x | kk[j].y) >> 63)) {
            u64x2 pair;
            pair.x = kk[j].x;
            pair.y = kk[j].y;
            __builtin_nontemporal_store(pair, reinterpret_cast<u64x2*>(dst));
          } else {                                             // phantom slots in front of the bucket's first key
            if (!(kk[j].x >> 63)) dst[0] = kk[j].x;
            if (!(kk[j].y >> 63)) dst[1] = kk[j].y;
          }
        }
      }
    }
  };

  int64_t t0 = sl.lo;
  typename Source::raw_t raw;
  src.template issue<true>(t0, sl.hi, raw);
  // a late key waits in kd[q] for its slot: ad[q] = {slot in the ring : 16 | lines its bucket has to give up first : 16}.
  // Every call of flush_lines takes exactly one line off a bucket that has late keys (its count is over 16 as long as they
  // wait), so the lane counts the flushes down itself.
  uint64_t k[IT] = {}, kd[IT] = {};
  unsigned ad[IT] = {};
  unsigned dm = 0;                                             // bit q: kd[q] is a late key
  auto one_flush_passed = [&]() {
#pragma unroll
    for (int q = 0; q < IT; ++q) ad[q] -= ((dm >> q) & 1u) << 16;
  };
  // writes the late keys whose turn has come; returns whether any of this lane's still wait
  auto write_late = [&]() {
    unsigned still = 0;
#pragma unroll
    for (int q = 0; q < IT; ++q) {
      if ((dm >> q) & 1u) {
        if ((ad[q] >> 16) == 0u) {
          if (!(RR_ABL & 4)) ring[ad[q]] = kd[q];
          dm &= ~(1u << q);
        } else {
          still = 1u;
        }
      }
    }
    return still;
  };
  unsigned dg[IT] = {};                                        // the keys' digits (a pair's digit is not part of the word it writes)
  auto make_keys = [&]() -> unsigned {
    if constexpr (split_digit<Source>::value) {
      return src.finish_split(t0, sl.hi, IT, raw, k, dg, tagl);
    } else {
      const unsigned m = src.finish(t0, sl.hi, IT, raw, k);
#pragma unroll
      for (int q = 0; q < IT; ++q) dg[q] = digit(k[q]);
      return m;
    }
  };
  Source::landed(raw);
  unsigned vm = make_keys();
  if (t0 + TILE < sl.hi) src.template issue<true>(t0 + TILE, sl.hi, raw);
  while (true) {
    // the previous round's late keys: their bucket's line left at the end of that round
    write_late();
    // ranks, branch-free and back to back (a position where no k-mer starts adds zero to some bucket)
#pragma unroll
    for (int q = 0; q < IT; ++q) ad[q] = atomicAdd(&cnt[dg[q]], (vm >> q) & 1u);
    unsigned ndm = 0, slow = 0;
#pragma unroll
    for (int q = 0; q < IT; ++q) {
      const unsigned rel = ad[q];
      const unsigned slot = dg[q] * 16u + (rel & 15u);
      const bool valid = (vm >> q) & 1u;
      if (!(RR_ABL & 4) && valid && rel < 16u) ring[slot] = k[q];
      ad[q] = slot | ((rel >> 4) << 16);
      kd[q] = k[q];
      ndm |= (valid && rel >= 16u) ? (1u << q) : 0u;
      slow |= (valid && rel >= 32u) ? 1u : 0u;
    }
    dm = ndm;
    if (slow) flags[0] = 1u;
    __syncthreads();
    const bool last = t0 + TILE >= sl.hi;
    if (!last) {                                               // the next tile's k-mers, the loads of the tile after next
      t0 += TILE;
      Source::landed(raw);
      vm = make_keys();
      if (t0 + TILE < sl.hi) src.template issue<true>(t0 + TILE, sl.hi, raw);
    }
    flush_lines();
    one_flush_passed();
    __syncthreads();
    if (flags[0] != 0u) {                                      // (uniform) somebody's key is more than a line behind
      while (true) {
        __syncthreads();
        if (tid == 0) { flags[0] = 0u; flags[1] = 0u; }
        __syncthreads();
        if (write_late()) flags[1] = 1u;
        __syncthreads();
        flush_lines();
        one_flush_passed();
        __syncthreads();
        if (flags[1] == 0u) break;
      }
    }
    if (last) break;
  }
  write_late();
  __syncthreads();
  for (int i = tid; i < B * 16; i += RP_THREADS) {              // what is left of every bucket: fewer than sixteen keys
    const int d = i >> 4, sl_ = i & 15;
    if ((unsigned)sl_ < cnt[d]) {
      const uint64_t key = ring[i];
      if (!(key >> 63)) out[((uint64_t)line_tab[d] << 4) + (unsigned)sl_] = key;
    }
  }
}

// seg_slabs[p] = slabs before segment p (p <= n_seg); one workgroup, any n_seg
__global__ __launch_bounds__(RP_THREADS) void rp_slab_table_kernel(const int64_t* __restrict__ seg_off, int64_t n_seg,
                                                                   int64_t slab_keys, int64_t* __restrict__ seg_slabs) {
  __shared__ int64_t smem[RP_THREADS / 64 + 1];
  __shared__ int64_t run;
  if (threadIdx.x == 0) run = 0;
  __syncthreads();
  for (int64_t p0 = 0; p0 < n_seg; p0 += RP_THREADS) {
    const int64_t p = p0 + threadIdx.x;
    int64_t c = 0;
    if (p < n_seg) c = (seg_off[p + 1] - seg_off[p] + slab_keys - 1) / slab_keys;
    const int64_t inc = wave_inclusive_scan(c);
    if (lane_id() == 63) smem[wave_id()] = inc;
    __syncthreads();
    int64_t base = run;
    for (int w = 0; w < wave_id(); ++w) base += smem[w];
    if (p < n_seg) seg_slabs[p] = base + inc - c;
    __syncthreads();
    if (threadIdx.x == RP_THREADS - 1) run = base + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) seg_slabs[n_seg] = run;
}

__global__ void rp_single_segment_kernel(int64_t n, int64_t* seg_off) {
  seg_off[0] = 0;
  seg_off[1] = n;
}

// child_off[p*B + c] = first output position of child bucket c of segment p (scanned histogram at slab 0)
__global__ void rp_child_offsets_kernel(const int64_t* __restrict__ scanned, const int64_t* __restrict__ seg_slabs,
                                        int64_t n_seg, int B, int64_t hn, int64_t* __restrict__ child_off) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, m = n_seg * B;
  for (; i < m; i += stride) {
    const int64_t p = i / B, c = i - p * B;
    const int64_t first = seg_slabs[p], nsl = seg_slabs[p + 1] - first;
    child_off[i] = scanned[first * B + c * nsl];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) child_off[m] = scanned[hn];      // == number of keys
}

int64_t rp_slab_keys(int64_t n) {
  int64_t k = ceil_div(ceil_div(n, 4096), RP_SLAB_UNIT) * RP_SLAB_UNIT;
  return std::max<int64_t>(k, (int64_t)RP_SLAB_UNIT * 4);
}

size_t align64(size_t x) { return (x + 63) & ~(size_t)63; }

// scratch needed by one level (seg_off for the single-segment case, slab table, histogram + scan partials)
size_t rp_level_scratch(int64_t n, int64_t n_seg, int bits) {
  const int64_t bound = n / rp_slab_keys(n) + n_seg + 1;
  const int64_t hn = bound << bits;
  return align64(16) + align64((size_t)(n_seg + 1) * 8) + align64((size_t)(hn + 1) * 8) + align64(bnpk_scan_scratch_bytes(hn));
}

// one level: slab table -> per-slab digit histogram -> scan -> child offsets -> write-combining scatter
template <typename Source>
int rp_level(bnpk_ctx* ctx, const Source& src, int64_t n, const int64_t* d_seg_off, int64_t n_seg, int shift, int bits,
             int64_t* d_out, int64_t* d_child_off, char* scratch, const char* hist_name, const char* scatter_name,
             hipStream_t s) {
  const int B = 1 << bits;
  const int64_t slab_keys = rp_slab_keys(n);
  const int64_t bound = n / slab_keys + n_seg + 1;
  if (bound > BNPK_MAX_BLOCKS / (RP_THREADS / 256)) return BNPK_ERR_RANGE;
  const int64_t hn = bound << bits;
  int64_t* own_seg = reinterpret_cast<int64_t*>(scratch);
  int64_t* seg_slabs = reinterpret_cast<int64_t*>(scratch + align64(16));
  int64_t* H = reinterpret_cast<int64_t*>(reinterpret_cast<char*>(seg_slabs) + align64((size_t)(n_seg + 1) * 8));
  int64_t* scan_scratch = reinterpret_cast<int64_t*>(reinterpret_cast<char*>(H) + align64((size_t)(hn + 1) * 8));
  bool* attr_set = ctx->launch_attr_set;
  constexpr int which = std::is_same<Source, mem_source>::value ? 0 : (std::is_same<Source, kmer_source<false>>::value ? 1 : 2);
  if (!attr_set[which]) {
    auto allow = [&](auto line_c, auto maxb_c) {
      constexpr int L_ = decltype(line_c)::value, M_ = decltype(maxb_c)::value;
      auto kernel = rp_scatter_kernel<Source, L_, M_>;
      return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rp_cfg<L_, M_>::LDS);
    };
    using std::integral_constant;
    BNPK_HIP(ctx, allow(integral_constant<int, 16>{}, integral_constant<int, 1024>{}));
    BNPK_HIP(ctx, allow(integral_constant<int, 8>{}, integral_constant<int, 1024>{}));
    BNPK_HIP(ctx, allow(integral_constant<int, 8>{}, integral_constant<int, 2048>{}));
    attr_set[which] = true;
  }
  if (!d_seg_off) {
    hipLaunchKernelGGL(rp_single_segment_kernel, dim3(1), dim3(1), 0, s, n, own_seg);
    d_seg_off = own_seg;
  }
  {
    bnpk_timer t(ctx, hist_name, s);
    hipLaunchKernelGGL(rp_slab_table_kernel, dim3(1), dim3(RP_THREADS), 0, s, d_seg_off, n_seg, slab_keys, seg_slabs);
    BNPK_HIP(ctx, hipMemsetAsync(H, 0, (size_t)(hn + 1) * 8, s));
    if constexpr (std::is_same<Source, kmer_source<false>>::value)
      hipLaunchKernelGGL(rp_hist_kmer_kernel, dim3((unsigned)bound), dim3(RP_THREADS), RP_HIST_LDS, s,
                         reinterpret_cast<const uint32_t*>(src.W), src.V, d_seg_off, (const int64_t*)seg_slabs, n_seg, slab_keys,
                         shift, bits, H);
    else if constexpr (std::is_same<Source, mem_source>::value)
      hipLaunchKernelGGL(rp_hist_mem_kernel, dim3((unsigned)bound), dim3(RP_THREADS), RP_HIST_LDS, s, src.keys, d_seg_off,
                         (const int64_t*)seg_slabs, n_seg, slab_keys, shift, bits, H);
    else
      hipLaunchKernelGGL((rp_hist_kernel<Source>), dim3((unsigned)bound), dim3(RP_THREADS), RP_HIST_LDS, s, src, d_seg_off,
                         (const int64_t*)seg_slabs, n_seg, slab_keys, shift, bits, H);
    BNPK_HIP(ctx, hipGetLastError());
    BNPK_CHECK(bnpk_scan_launch(ctx, H, hn, 1, H, true, scan_scratch, s));
    if (d_child_off)
      hipLaunchKernelGGL(rp_child_offsets_kernel, dim3(grid_for(ceil_div(n_seg * B, 256))), dim3(256), 0, s,
                         (const int64_t*)H, (const int64_t*)seg_slabs, n_seg, B, hn, d_child_off);
  }
  bnpk_timer t(ctx, scatter_name, s);
  if constexpr (!std::is_same<Source, mem_source>::value) {
    if (ctx->l1_ring && bits <= 10) {                          // the fixed-line form of the fused first level (rp_ring_kernel)
      auto kernel = rp_ring_kernel<Source>;
      if (!attr_set[4 + which]) {
        BNPK_HIP(ctx, hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rr_cfg::LDS));
        attr_set[4 + which] = true;
      }
      hipLaunchKernelGGL(kernel, dim3((unsigned)bound), dim3(RP_THREADS), rr_cfg::LDS, s, src, d_seg_off, (const int64_t*)seg_slabs,
                         n_seg, slab_keys, shift, bits, (const int64_t*)H, reinterpret_cast<uint64_t*>(d_out));
      BNPK_HIP(ctx, hipGetLastError());
      return BNPK_OK;
    }
  }
  // (the granule: 16 keys for levels that read their keys from memory, 8 for the fused first level — see rp_cfg)
  constexpr int line = std::is_same<Source, mem_source>::value ? 16 : RP_L1_LINE;
  auto launch = [&](auto line_c, auto maxb_c) {
    constexpr int L_ = decltype(line_c)::value, M_ = decltype(maxb_c)::value;
    constexpr size_t lds = rp_cfg<L_, M_>::LDS;
    auto kernel = rp_scatter_kernel<Source, L_, M_>;
    hipLaunchKernelGGL(kernel, dim3((unsigned)bound), dim3(RP_THREADS), lds, s, src, d_seg_off, (const int64_t*)seg_slabs, n_seg,
                       slab_keys, shift, bits, (const int64_t*)H, reinterpret_cast<uint64_t*>(d_out), rp_claim_t{});
  };
  using std::integral_constant;
  if (bits > 10) launch(integral_constant<int, 8>{}, integral_constant<int, 2048>{});
  else if (line == 16) launch(integral_constant<int, 16>{}, integral_constant<int, 1024>{});
  else launch(integral_constant<int, 8>{}, integral_constant<int, 1024>{});
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

// sizes of the claimed buckets -> (dense-equivalent) bucket offsets: n_b = min(front, CAP_LO) + min(tail, TAIL)
__global__ void rp_claimed_sizes_kernel(const unsigned* __restrict__ fill, int64_t n_buckets, int64_t* __restrict__ sizes) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n_buckets; i += stride)
    sizes[i] = (int64_t)min(fill[2 * i], (unsigned)RPC_CAP_LO) + (int64_t)min(fill[2 * i + 1], (unsigned)RPC_TAIL);
}

// a bucket's leftover keys move from the tail of its slots to right behind its lines: the bucket is ONE dense run from then
// on and the finishing kernels read it like any other, only at slot b * RPC_STRIDE instead of at its offset.  One wavefront
// per bucket takes the <= 128 keys into registers before it stores any of them (source and destination may overlap).
__global__ __launch_bounds__(256) void rp_claimed_tails_kernel(uint64_t* __restrict__ buckets, const unsigned* __restrict__ fill,
                                                               int64_t n_buckets) {
  const int lane = threadIdx.x & 63;
  const int64_t waves = (int64_t)gridDim.x * (256 / 64);
  for (int64_t b = (int64_t)blockIdx.x * (256 / 64) + (threadIdx.x >> 6); b < n_buckets; b += waves) {
    const unsigned lo = min(fill[2 * b], (unsigned)RPC_CAP_LO), hi = min(fill[2 * b + 1], (unsigned)RPC_TAIL);
    if (hi == 0 || lo == (unsigned)RPC_CAP_LO) continue;
    uint64_t* base = buckets + b * RPC_STRIDE;
    uint64_t v0 = 0, v1 = 0;
    if ((unsigned)lane < hi) v0 = base[RPC_CAP_LO + lane];
    if ((unsigned)lane + 64 < hi) v1 = base[RPC_CAP_LO + lane + 64];
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((unsigned)lane < hi) base[lo + lane] = v0;
    if ((unsigned)lane + 64 < hi) base[lo + lane + 64] = v1;
  }
}

// the claiming level: slab table -> scatter (no histogram, no scan)
int rp_level_claimed(bnpk_ctx* ctx, const mem_source& src, int64_t n, const int64_t* d_seg_off, int64_t n_seg, int shift, int bits,
                     uint64_t* d_buckets, const rp_claim_t& claim, char* scratch, hipStream_t s) {
  const int64_t slab_keys = rp_slab_keys(n);
  const int64_t bound = n / slab_keys + n_seg + 1;
  if (bound > BNPK_MAX_BLOCKS / (RP_THREADS / 256)) return BNPK_ERR_RANGE;
  int64_t* own_seg = reinterpret_cast<int64_t*>(scratch);
  int64_t* seg_slabs = reinterpret_cast<int64_t*>(scratch + align64(16));
  auto kernel = rp_scatter_kernel<mem_source, 16, 1024, true>;
  constexpr size_t lds = rp_cfg<16, 1024>::LDS_CLAIM;
  if (!ctx->launch_attr_set[4]) {
    BNPK_HIP(ctx, hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ctx->launch_attr_set[4] = true;
  }
  if (!d_seg_off) {
    hipLaunchKernelGGL(rp_single_segment_kernel, dim3(1), dim3(1), 0, s, n, own_seg);
    d_seg_off = own_seg;
  }
  hipLaunchKernelGGL(rp_slab_table_kernel, dim3(1), dim3(RP_THREADS), 0, s, d_seg_off, n_seg, slab_keys, seg_slabs);
  bnpk_timer t(ctx, "radix_scatter_claimed", s);
  hipLaunchKernelGGL(kernel, dim3((unsigned)bound), dim3(RP_THREADS), lds, s, src, d_seg_off, (const int64_t*)seg_slabs, n_seg, slab_keys,
                     shift, bits, (const int64_t*)nullptr, d_buckets, claim);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

// ---- one more level over MANY SMALL segments ------------------------------------------------------------------------
// When some buckets come out larger than the finishing kernel's capacity (skewed / duplicate-heavy keys) the host
// splits all of them once more by a few bits.  With ~10^6 segments of a few thousand keys the slab kernels above
// spend their time on per-slab setup and unaligned edge lines; here one workgroup takes a whole segment (at most
// RS_CAP keys) into registers, ranks the keys with wave ballots (no atomics: match lanes with the same digit),
// groups them in LDS and writes the segment back as one contiguous run.
constexpr int RS_THREADS = 1024;
constexpr int RS_ITEMS = 16;
constexpr int RS_CAP = RS_THREADS * RS_ITEMS;           // 16384 keys = 128 KiB
constexpr int RS_MAXBITS = 4;
constexpr int RS_MAXB = 1 << RS_MAXBITS;

__global__ __launch_bounds__(RS_THREADS) void rp_split_small_kernel(const uint64_t* __restrict__ keys,
                                                                    const int64_t* __restrict__ seg_off, int64_t n_seg,
                                                                    int shift, int bits, uint64_t* __restrict__ out,
                                                                    int64_t* __restrict__ child_off,
                                                                    unsigned long long* __restrict__ too_big) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* stage = reinterpret_cast<uint64_t*>(smem);                               // RS_CAP keys
  unsigned* wcnt = reinterpret_cast<unsigned*>(smem + (size_t)RS_CAP * 8);          // [digit][wave] counts -> offsets
  unsigned* dstart = wcnt + RS_MAXB * (RS_THREADS / 64);                            // [digit] start inside the segment
  const int B = 1 << bits;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = RS_THREADS / 64;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  for (int64_t seg = blockIdx.x; seg < n_seg; seg += gridDim.x) {
    const int64_t lo = seg_off[seg], hi = seg_off[seg + 1];
    int n = (int)min(hi - lo, (int64_t)RS_CAP + 1);
    if (n > RS_CAP) {                                  // (the host only calls this for small segments)
      if (tid == 0) atomicOr(too_big, 1ull);
      n = RS_CAP;
    }
    for (int i = tid; i < RS_MAXB * NW; i += RS_THREADS) wcnt[i] = 0;
    __syncthreads();
    uint64_t k[RS_ITEMS];
    unsigned pos[RS_ITEMS];                            // digit << 16 | rank among the wavefront's keys of that digit
#pragma unroll
    for (int q = 0; q < RS_ITEMS; ++q) {
      const int i = tid + q * RS_THREADS;
      const bool ok = i < n;
      k[q] = ok ? keys[lo + i] : 0ull;
      const unsigned d = (unsigned)(k[q] >> shift) & (B - 1);
      uint64_t m = __ballot(ok);                       // lanes of this wavefront holding the same digit
      for (int b = 0; b < bits; ++b) {
        const uint64_t bb = __ballot((d >> b) & 1u);
        m &= ((d >> b) & 1u) ? bb : ~bb;
      }
      pos[q] = 0;
      if (ok) {
        const unsigned before = wcnt[d * NW + wave];   // every lane of the group reads, then its first lane adds
        pos[q] = (d << 16) | (before + (unsigned)__popcll(m & lt_mask));
        __builtin_amdgcn_wave_barrier();
        if ((m & lt_mask) == 0) wcnt[d * NW + wave] = before + (unsigned)__popcll(m);
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    __syncthreads();
    if (wave == 0) {                                   // exclusive scan over (digit, wave): B * 16 <= 256 entries
      unsigned run = 0;
      for (int base = 0; base < B * NW; base += 64) {
        const unsigned c = wcnt[base + lane];
        const unsigned inc = wave_inclusive_scan(c);
        wcnt[base + lane] = run + inc - c;
        run += (unsigned)__builtin_amdgcn_readlane((int)inc, 63);
      }
    }
    __syncthreads();
    if (tid < B) {
      dstart[tid] = wcnt[tid * NW];
      child_off[seg * B + tid] = lo + wcnt[tid * NW];
    }
#pragma unroll
    for (int q = 0; q < RS_ITEMS; ++q) {
      const int i = tid + q * RS_THREADS;
      if (i < n) stage[wcnt[(pos[q] >> 16) * NW + wave] + (pos[q] & 0xffffu)] = k[q];
    }
    __syncthreads();
    for (int i = tid; i < n; i += RS_THREADS) out[lo + i] = stage[i];
    __syncthreads();
  }
}


// tags of the pair level: bucket b's bit = parity of the number of non-empty buckets before it; list[j] = the j-th non-empty
// bucket, list_n[0] = how many there are.  One workgroup (B <= 1024 buckets).
__global__ __launch_bounds__(RP_THREADS) void rp_pair_tags_kernel(const int64_t* __restrict__ child_off, int B, unsigned* __restrict__ tags,
                                                                  int64_t* __restrict__ list, int64_t* __restrict__ list_n) {
  __shared__ unsigned wsum[RP_THREADS / 64];
  __shared__ unsigned bits[rr_cfg::MAXB / 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < rr_cfg::MAXB / 32) bits[tid] = 0;
  const unsigned full = tid < B && child_off[tid + 1] > child_off[tid] ? 1u : 0u;
  const unsigned inc = wave_inclusive_scan(full);
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  unsigned before = inc - full;
  for (int w = 0; w < wave; ++w) before += wsum[w];
  if (tid < B && (before & 1u)) atomicOr(&bits[tid >> 5], 1u << (tid & 31));
  if (full) list[before] = tid;
  if (tid == RP_THREADS - 1) list_n[0] = before + full;
  __syncthreads();
  if (tid < rr_cfg::MAXB / 32) tags[tid] = bits[tid];
}

}  // namespace

// (internal, used by sparse.hip) the first level of the k-mer index's partition: n (k-mer, row) pairs -> n words
// ((k-mer's low key_bits - bits bits : row : tag), see pair_source) grouped by the k-mer's top `bits` bits; d_child_off
// (2^bits + 1), d_list (2^bits: the non-empty buckets in order), d_list_n (1).  bits <= 10; key_bits - bits + row_bits + 1 <= 63.
int bnpk_pairs_partition_launch(bnpk_ctx* ctx, const int64_t* d_keys, const int64_t* d_rows, int64_t n, int key_bits, int bits,
                                int row_bits, int64_t* d_words, int64_t* d_child_off, unsigned* d_tags, int64_t* d_list,
                                int64_t* d_list_n, hipStream_t s) {
  if (bits < 1 || bits > 10 || key_bits - bits + row_bits + 1 > 63 || key_bits <= bits) return BNPK_ERR_ARG;
  void* scratch_v = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, rp_level_scratch(n, 1, bits), &scratch_v, s));
  char* scratch = (char*)scratch_v;
  const int B = 1 << bits;
  const int64_t slab_keys = rp_slab_keys(n);
  const int64_t bound = n / slab_keys + 2;
  if (bound > BNPK_MAX_BLOCKS / (RP_THREADS / 256)) return BNPK_ERR_RANGE;
  const int64_t hn = bound << bits;
  int64_t* own_seg = reinterpret_cast<int64_t*>(scratch);
  int64_t* seg_slabs = reinterpret_cast<int64_t*>(scratch + align64(16));
  int64_t* H = reinterpret_cast<int64_t*>(reinterpret_cast<char*>(seg_slabs) + align64((size_t)2 * 8));
  int64_t* scan_scratch = reinterpret_cast<int64_t*>(reinterpret_cast<char*>(H) + align64((size_t)(hn + 1) * 8));
  pair_source src{reinterpret_cast<const uint64_t*>(d_keys), d_rows, d_tags, key_bits - bits, row_bits};
  auto kernel = rp_ring_kernel<pair_source>;
  if (!ctx->launch_attr_set[7]) {
    BNPK_HIP(ctx, hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rr_cfg::LDS));
    ctx->launch_attr_set[7] = true;
  }
  hipLaunchKernelGGL(rp_single_segment_kernel, dim3(1), dim3(1), 0, s, n, own_seg);
  {
    bnpk_timer t(ctx, "pairs_partition_hist", s);
    hipLaunchKernelGGL(rp_slab_table_kernel, dim3(1), dim3(RP_THREADS), 0, s, (const int64_t*)own_seg, (int64_t)1, slab_keys, seg_slabs);
    BNPK_HIP(ctx, hipMemsetAsync(H, 0, (size_t)(hn + 1) * 8, s));
    hipLaunchKernelGGL((rp_hist_kernel<pair_source>), dim3((unsigned)bound), dim3(RP_THREADS), RP_HIST_LDS, s, src, (const int64_t*)own_seg,
                       (const int64_t*)seg_slabs, (int64_t)1, slab_keys, key_bits - bits, bits, H);
    BNPK_HIP(ctx, hipGetLastError());
    BNPK_CHECK(bnpk_scan_launch(ctx, H, hn, 1, H, true, scan_scratch, s));
    hipLaunchKernelGGL(rp_child_offsets_kernel, dim3(grid_for(ceil_div(B, 256))), dim3(256), 0, s, (const int64_t*)H,
                       (const int64_t*)seg_slabs, (int64_t)1, B, hn, d_child_off);
    hipLaunchKernelGGL(rp_pair_tags_kernel, dim3(1), dim3(RP_THREADS), 0, s, (const int64_t*)d_child_off, B, d_tags, d_list, d_list_n);
  }
  bnpk_timer t(ctx, "pairs_partition_scatter", s);
  hipLaunchKernelGGL(kernel, dim3((unsigned)bound), dim3(RP_THREADS), rr_cfg::LDS, s, src, (const int64_t*)own_seg, (const int64_t*)seg_slabs,
                     (int64_t)1, slab_keys, key_bits - bits, bits, (const int64_t*)H, reinterpret_cast<uint64_t*>(d_words));
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

extern "C" {

#ifdef RP_PHASES
int bnpk_debug_radix_phases(unsigned long long* out8) {   // reads and clears the counters
  unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(rp_phase_cycles), sizeof(zero)) != hipSuccess) return BNPK_ERR_HIP;
  return hipMemcpyToSymbol(HIP_SYMBOL(rp_phase_cycles), zero, sizeof(zero)) == hipSuccess ? BNPK_OK : BNPK_ERR_HIP;
}
#endif

int64_t bnpk_radix_max_bits(void) { return RP_MAXBITS; }

int bnpk_radix_partition(bnpk_ctx* ctx, const int64_t* d_keys, int64_t n, const int64_t* d_seg_offsets, int64_t n_seg,
                         int shift, int bits, int64_t* d_out, int64_t* d_child_offsets, void* stream) {
  if (!ctx || n < 0 || n_seg < 1 || bits < 0 || bits > RP_MAXBITS || shift < 0 || shift + bits > 63) return BNPK_ERR_ARG;
  if (n >= (1ll << 35)) return BNPK_ERR_RANGE;
  if (n > 0 && (!d_keys || !d_out || d_keys == d_out)) return BNPK_ERR_ARG;
  if (n_seg > 1 && !d_seg_offsets) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  void* scratch = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, rp_level_scratch(n, n_seg, bits), &scratch, (hipStream_t)stream));
  mem_source src{reinterpret_cast<const uint64_t*>(d_keys)};
  return rp_level(ctx, src, n, d_seg_offsets, n_seg, shift, bits, d_out, d_child_offsets, (char*)scratch,
                  "radix_hist", "radix_scatter", s);
}

int bnpk_kmers_partition(bnpk_ctx* ctx, const uint64_t* d_packed, const uint64_t* d_kmer_starts, int64_t n_bases, int k,
                         int canonical, int shift, int bits, int64_t* d_out, int64_t* d_child_offsets, void* stream) {
  if (!ctx || k < 1 || k > 31 || n_bases < 0 || bits < 0 || bits > RP_MAXBITS || shift < 0 || shift + bits > 2 * k)
    return BNPK_ERR_ARG;
  if (n_bases >= (1ll << 35)) return BNPK_ERR_RANGE;
  if (n_bases > 0 && (!d_packed || !d_kmer_starts || !d_out)) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  void* scratch = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, rp_level_scratch(n_bases, 1, bits), &scratch, (hipStream_t)stream));
  if (canonical) {
    kmer_source<true> src{d_packed, reinterpret_cast<const uint8_t*>(d_kmer_starts), n_bases / 32 + 2, k};
    return rp_level(ctx, src, n_bases, nullptr, 1, shift, bits, d_out, d_child_offsets, (char*)scratch,
                    "kmers_partition_hist", "kmers_partition_scatter", s);
  }
  kmer_source<false> src{d_packed, reinterpret_cast<const uint8_t*>(d_kmer_starts), n_bases / 32 + 2, k};
  return rp_level(ctx, src, n_bases, nullptr, 1, shift, bits, d_out, d_child_offsets, (char*)scratch,
                  "kmers_partition_hist", "kmers_partition_scatter", s);
}

int64_t bnpk_claimed_stride(void) { return RPC_STRIDE; }
int64_t bnpk_claimed_cap_lo(void) { return RPC_CAP_LO; }

int bnpk_radix_partition_claimed(bnpk_ctx* ctx, const int64_t* d_keys, int64_t n, const int64_t* d_seg_offsets, int64_t n_seg,
                                 int shift, int bits, int64_t* d_buckets, uint32_t* d_fill, int64_t* d_bag, int64_t bag_cap,
                                 int64_t* d_bag_fill, void* stream) {
  if (!ctx || n < 0 || n_seg < 1 || bits < 1 || bits > 10 || shift < 0 || shift + bits > 63 || !d_fill || !d_bag_fill || bag_cap < 0)
    return BNPK_ERR_ARG;
  // The claims of a child bucket are counted in 32 bits (fill[2c], fill[2c+1]).  A counter can only wrap after 2^32 keys of
  // ONE bucket were claimed — and all but the first 7680 of those went to the bag, whose fill is counted in 64 bits: with
  // bag_cap <= n / 8 < 2^32 (n < 2^35) *d_bag_fill is over bag_cap long before, which every caller must treat as "keys were
  // dropped, take bnpk_radix_partition" (ops / sparse.hip do).  So a wrapped counter never yields an accepted result, and the
  // 6e9-key batch of the headline keeps this level (round 6 briefly refused n >= 2^32 here: 77 -> 83 ms).
  if (n >= (1ll << 35)) return BNPK_ERR_RANGE;
  if (bag_cap >= (1ll << 32)) return BNPK_ERR_ARG;
  if (n > 0 && (!d_keys || !d_buckets || (bag_cap > 0 && !d_bag))) return BNPK_ERR_ARG;
  if (n_seg > 1 && !d_seg_offsets) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  BNPK_HIP(ctx, hipMemsetAsync(d_fill, 0, (size_t)(n_seg << bits) * 2 * sizeof(uint32_t), s));
  BNPK_HIP(ctx, hipMemsetAsync(d_bag_fill, 0, sizeof(int64_t), s));
  if (n == 0) return BNPK_OK;
  void* scratch = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, align64(16) + align64((size_t)(n_seg + 1) * 8), &scratch, s));
  mem_source src{reinterpret_cast<const uint64_t*>(d_keys)};
  rp_claim_t claim{d_fill, reinterpret_cast<uint64_t*>(d_bag), reinterpret_cast<unsigned long long*>(d_bag_fill), bag_cap};
  return rp_level_claimed(ctx, src, n, d_seg_offsets, n_seg, shift, bits, reinterpret_cast<uint64_t*>(d_buckets), claim, (char*)scratch, s);
}

int bnpk_claimed_finalize(bnpk_ctx* ctx, int64_t* d_buckets, const uint32_t* d_fill, int64_t n_buckets, int64_t* d_bucket_offsets,
                          void* stream) {
  if (!ctx || n_buckets < 1 || !d_fill || !d_bucket_offsets || !d_buckets) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  void* scratch = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, align64((size_t)n_buckets * 8) + align64(bnpk_scan_scratch_bytes(n_buckets)), &scratch, s));
  int64_t* sizes = reinterpret_cast<int64_t*>(scratch);
  int64_t* scan_scratch = reinterpret_cast<int64_t*>((char*)scratch + align64((size_t)n_buckets * 8));
  bnpk_timer t(ctx, "claimed_finalize", s);
  hipLaunchKernelGGL(rp_claimed_tails_kernel, dim3(grid_for(std::min<int64_t>(ceil_div(n_buckets, 4), (int64_t)ctx->compute_units * 32))),
                     dim3(256), 0, s, reinterpret_cast<uint64_t*>(d_buckets), d_fill, n_buckets);
  hipLaunchKernelGGL(rp_claimed_sizes_kernel, dim3(grid_for(ceil_div(n_buckets, 256))), dim3(256), 0, s, d_fill, n_buckets, sizes);
  BNPK_HIP(ctx, hipGetLastError());
  return bnpk_scan_launch(ctx, sizes, n_buckets, 1, d_bucket_offsets, true, scan_scratch, s);
}

int64_t bnpk_radix_small_capacity(void) { return RS_CAP; }

int bnpk_radix_partition_small(bnpk_ctx* ctx, const int64_t* d_keys, int64_t n, const int64_t* d_seg_offsets,
                               int64_t n_seg, int shift, int bits, int64_t* d_out, int64_t* d_child_offsets,
                               void* stream) {
  if (!ctx || n < 0 || n_seg < 1 || bits < 1 || bits > RS_MAXBITS || shift < 0 || shift + bits > 63 || !d_seg_offsets ||
      !d_child_offsets)
    return BNPK_ERR_ARG;
  if (n > 0 && (!d_keys || !d_out || d_keys == d_out)) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  void* scratch = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, 64, &scratch, (hipStream_t)stream));
  const size_t lds = (size_t)RS_CAP * 8 + (size_t)RS_MAXB * (RS_THREADS / 64) * 4 + RS_MAXB * 4;
  if (!ctx->launch_attr_set[3]) {
    BNPK_HIP(ctx, hipFuncSetAttribute((const void*)rp_split_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ctx->launch_attr_set[3] = true;
  }
  BNPK_HIP(ctx, hipMemsetAsync(scratch, 0, 8, s));
  {
    bnpk_timer t(ctx, "radix_split_small", s);
    const unsigned grid = (unsigned)std::min<int64_t>(n_seg, (int64_t)ctx->compute_units * 8);
    hipLaunchKernelGGL(rp_split_small_kernel, dim3(grid), dim3(RS_THREADS), lds, s, reinterpret_cast<const uint64_t*>(d_keys),
                       d_seg_offsets, n_seg, shift, bits, reinterpret_cast<uint64_t*>(d_out), d_child_offsets,
                       reinterpret_cast<unsigned long long*>(scratch));
    BNPK_HIP(ctx, hipGetLastError());
    // the last boundary
    BNPK_HIP(ctx, hipMemcpyAsync(d_child_offsets + (n_seg << bits), d_seg_offsets + n_seg, 8, hipMemcpyDeviceToDevice, s));
  }
  unsigned long long flag = 0;
  BNPK_HIP(ctx, hipMemcpyAsync(&flag, scratch, 8, hipMemcpyDeviceToHost, s));
  BNPK_HIP(ctx, hipStreamSynchronize(s));
  return flag ? BNPK_ERR_RANGE : BNPK_OK;
}

}  // extern "C"
