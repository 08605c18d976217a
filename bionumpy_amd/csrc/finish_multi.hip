// Finishing kernel of the sparse k-mer histogram (A9 for k > 13) for keys that are NEARLY all distinct but repeat often
// enough that most buckets hold a repeat: random 21-mers (0.07 % repeats, ~4 per bucket), reads at a coverage near 1x,
// metagenomes.  np.unique(return_counts=True) semantics (bionumpy/sequence/count_encoded.py:150-188 extended to k > 8).
//
// The fast kernel (finish.hip) ranks such a bucket at full speed but cannot emit it: it knows THAT a key repeats, not
// how often, and its output positions ("bucket_off[b] minus the duplicates announced so far") only hold while buckets
// with repeats are rare.  The duplicate-aware tables (finish_wave.hip, finish_dup.hip) overflow on ~5.7 K distinct keys.
// What was left was the general kernel at 0.13 of the HBM peak.  This kernel is the fast kernel's counting sort and
// slot-owner ranking with three changes:
//   * multiplicities.  A slot owner counts, next to the smaller keys on either side, the EQUAL keys on its left: its
//     place among the bucket's keys with repeats is  s - min(t, s) + smaller + equal-left  (unique per key; equal keys
//     end up adjacent, first occurrence first), and the number of distinct keys D = n - (keys with an equal key on
//     their left) is exact.
//   * exact output positions, and nobody waits for anybody who waits.  A bucket's output position is the number of
//     distinct keys of ALL earlier buckets.  Every bucket publishes its D (status[b] = D + 1) the moment its walk is
//     over, and only then looks for its own prefix.  What makes that possible is a PARKING place: the owners do not
//     store their keys to the output (whose position is not known yet) but, sorted, to one of two slots of the
//     workgroup in a small ring in global memory (2 x 61 KB per workgroup, 63 MB in all: rewritten every other bucket,
//     so it lives in the L2 / Infinity Cache and costs no HBM traffic to speak of).  The bucket is emitted from there
//     TWO iterations later: by then the buckets before it, which are sorted by other workgroups at about the same
//     time, have had a whole round to publish, and the prefix — summed chain-free from the status words, the
//     workgroup remembers how far it got — is there without a wait almost always.  (First version of this kernel:
//     the sorted bucket waited in the LDS stage and had to leave before the next bucket could be placed, i.e. the wait
//     for the prefix came BEFORE the next publication; every stall propagated, the chip ran in lock step: 72 ms per
//     6e9 keys against 38 without the waits.  Measured then: without waiting only 5 % of the buckets find their
//     prefix complete at the top of the next iteration — on average 350 of the ~512 buckets of the same round are
//     still missing.)
//   * run-length emission from the parked bucket: position p is a first occurrence iff park[p - 1] != park[p], its
//     count the length of the run behind it, its output slot the number of first occurrences before it: the owners
//     counted the repeats per 960-position range (rare LDS atomics) so a wavefront knows where its range starts
//     without a barrier, and ballots do the rest.  The stores are contiguous runs.
// Buckets are handed out by a ticket counter only (never by blockIdx), so a bucket is always owned by a RUNNING
// workgroup and the waits cannot deadlock whatever the grid size or whoever else occupies CUs.
// A bucket with a bin of more than 64 keys (a key repeated that often, or skewed low bits) is not sorted here: its
// duplicates are counted exactly, it takes its place in the output and goes to finish_sorted_kernel<REDO> by a list.
#include <algorithm>

#include "finish.h"

namespace {

constexpr int FM_THREADS = 512;
constexpr int FM_WAVES = FM_THREADS / 64;
constexpr int FM_ITEMS = 15;
constexpr int FM_CAP = FM_THREADS * FM_ITEMS;            // 7680 keys
constexpr int FM_SLICE = 64 * FM_ITEMS;                  // slots owned (and sorted positions emitted) by one wavefront
constexpr int FM_MAXBITS = 13;
constexpr int FM_MAXBINS = 1 << FM_MAXBITS;
constexpr int FM_SCAN_DW = FM_MAXBINS / 2 / FM_THREADS;  // packed bin words scanned by one lane (8)
constexpr int FM_NEAR = 64;                              // guard slots around the stage
constexpr int FM_WG = 3;                                 // chunks of 64 slots whose neighbour walks advance together
constexpr int FM_USUAL = 12;                             // items the usual bucket fills
constexpr int FM_PARK = FM_CAP + 3 * FM_NEAR;            // keys of one parking slot: [0, 64) front pad (a sentinel at 63), the bucket, a sentinel, pad (run walks read up to 128 past the bucket)
#ifndef FM_DEFER
#define FM_DEFER 3                                       // a bucket is emitted this many iterations after it was sorted
#endif
constexpr int FM_SLOTS = FM_DEFER + 1;                   // parking slots of a workgroup (the emission overlaps the next bucket's walk)
#ifndef FM_UNROLL
#define FM_UNROLL 4
#endif
#ifndef FM_ABL
#define FM_ABL 0                                         // experiment builds: 1 = no look-back (bases = bucket offsets), 4 = no output stores, 32 = walks of one step
#endif
#ifndef FM_GROUP_WALK
#define FM_GROUP_WALK 1                                  // the neighbour walks of a group of chunks go as far as ITS longest bin (0: the bucket's)
#endif
#ifndef FM_LB
#define FM_LB 2                                          // status words per lane in flight: a poll covers 64 * 8 * FM_LB buckets
#endif
constexpr unsigned FM_SPIN_LIMIT = 1u << 22;
static_assert(FM_ITEMS % FM_WG == 0 && FM_USUAL % FM_WG == 0, "whole groups");
static_assert(FM_DEFER >= 2, "the prefix of a bucket is looked for while the next one is sorted");
constexpr size_t FM_OFF_STAGE = (size_t)FM_NEAR * 8;
constexpr size_t FM_OFF_P = FM_OFF_STAGE + (size_t)(FM_CAP + FM_NEAR) * 8;
constexpr size_t FM_OFF_WSUM = FM_OFF_P + (((size_t)(FM_MAXBINS + 2) * 2 + 15) & ~(size_t)15);
constexpr size_t FM_OFF_DUPC = FM_OFF_WSUM + 3 * FM_WAVES * 4;                        // repeats per range of sorted positions, per parking slot
constexpr size_t FM_OFF_LBS = FM_OFF_DUPC + FM_SLOTS * FM_WAVES * 4;                         // per polled chunk of 64 status words: 0 = incomplete, else sum + 1
constexpr size_t FM_OFF_SH = FM_OFF_LBS + (size_t)FM_WAVES * FM_LB * 4;
constexpr size_t FM_LDS = FM_OFF_SH + 8 * 8;
static_assert(2 * FM_LDS <= 160 * 1024, "two workgroups per CU");

__device__ __forceinline__ int64_t fm_uniform(int64_t v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uint64_t)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ int fm_fresh(int x) {
  asm volatile("" : "+v"(x));
  return x;
}

// a bucket on its way out: what the emission needs
struct fm_pending {
  int kind;                                              // 0 nothing, 1 parked (sorted), 2 pre-counted by the caller, 3 left to the general kernel, 4 only the total
  int b, nb, slot;
  unsigned D;
  int64_t src;                                           // kind 2: where its pairs lie; experiments: the bucket's offset
};

// status[b]: 0 until bucket b's number of distinct keys D is known, then D + 1 (zeroed by the caller).
// header: [FS_FLAGS] 1 = over-capacity bucket without a pre-counted entry, 2 = a wait gave up; [FS_UNIQUE] distinct keys;
// [FS_REDO] length of the redo list (ids, output bases); [FS_FTICKET] the ticket counter (a 128-byte line of its own).
// park: gridDim.x * FM_SLOTS slots of FM_PARK keys.
__global__ __launch_bounds__(FM_THREADS, 4) void finish_multi_kernel(
    const uint64_t* __restrict__ A, const int64_t* __restrict__ bucket_off, int64_t n_buckets, int sshift, int sbits,
    unsigned long long* __restrict__ header, unsigned* __restrict__ status, uint64_t* __restrict__ keys_out,
    int64_t* __restrict__ counts_out, const int64_t* __restrict__ big_table, int n_big,
    const uint64_t* __restrict__ big_keys, const int64_t* __restrict__ big_counts, unsigned* __restrict__ redo_ids,
    int64_t* __restrict__ redo_bases, uint64_t* park, int64_t pstride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* stage = reinterpret_cast<uint64_t*>(smem + FM_OFF_STAGE);
  unsigned* P32 = reinterpret_cast<unsigned*>(smem + FM_OFF_P);
  const unsigned short* P16 = reinterpret_cast<const unsigned short*>(smem + FM_OFF_P);
  unsigned* wsum = reinterpret_cast<unsigned*>(smem + FM_OFF_WSUM);             // [0..7] scan, [8..15] repeats, [16..23] longest bins
  unsigned* dupc = reinterpret_cast<unsigned*>(smem + FM_OFF_DUPC);             // [slot][wavefront range]
  unsigned* lbs = reinterpret_cast<unsigned*>(smem + FM_OFF_LBS);
  long long* sh = reinterpret_cast<long long*>(smem + FM_OFF_SH);               // [1] abort, [2] next ticket, [3] prefix of the distinct counts, [7] ... up to which bucket
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned SB = 1u << sbits;
  const unsigned n_dw = SB > 1 ? SB >> 1 : 1u;
  const unsigned dwl = n_dw >= FM_THREADS ? n_dw / FM_THREADS : 1u;
  // the bucket of slot j lies at mypark + j * FM_PARK + FM_NEAR
  uint64_t* mypark = park + (size_t)((FM_ABL & 16) ? blockIdx.x % 32 : blockIdx.x) * (FM_SLOTS * FM_PARK);   // (16: experiment — a ring small enough for the L2, shared and wrong)

  for (unsigned i = tid; i <= n_dw; i += FM_THREADS) P32[i] = 0;
  if (tid < FM_NEAR) stage[tid - FM_NEAR] = ~0ull;       // guard keys in front of slot 0 (keys are < 2^63)
  if (tid < FM_SLOTS * FM_WAVES) dupc[tid] = 0;
  if (tid == 0) { sh[1] = 0; sh[3] = 0; sh[7] = 0; }
  if (tid < FM_SLOTS) mypark[tid * FM_PARK + FM_NEAR - 1] = ~0ull;   // the sentinel in front of every parking slot

  uint64_t k[FM_ITEMS];
  struct bucket_t { int64_t lo; int nb; int64_t size; int64_t src; };   // src: where the keys lie in A (lo, or b * pstride)
  const int nbk = (int)n_buckets;                        // (< 2^30: the launcher checks)
  auto fetch_offsets = [&](int bb, int64_t& o0, int64_t& o1) {
    o0 = 0; o1 = 0;
    if (bb < nbk) { o0 = bucket_off[bb]; o1 = bucket_off[bb + 1]; }
  };
  auto open_bucket = [&](int bb, int64_t o0, int64_t o1) {
    bucket_t x;
    x.lo = fm_uniform(o0);
    x.size = fm_uniform(o1) - x.lo;
    x.nb = x.size > FM_CAP ? 0 : (int)x.size;
    x.src = pstride ? (int64_t)bb * pstride : x.lo;
    return x;
  };
  auto load_keys = [&](const bucket_t& x) {
    const uint64_t* Ab = A + x.src;
    const int t = fm_fresh(tid);
    if (x.nb > 0) {
#pragma unroll
      for (int q = 0; q < FM_USUAL; ++q) k[q] = __builtin_nontemporal_load(&Ab[(unsigned)min(t + q * FM_THREADS, x.nb - 1)]);
      if (x.nb > FM_USUAL * FM_THREADS) {
#pragma unroll
        for (int q = FM_USUAL; q < FM_ITEMS; ++q) k[q] = __builtin_nontemporal_load(&Ab[(unsigned)min(t + q * FM_THREADS, x.nb - 1)]);
      }
    }
  };
#pragma unroll
  for (int q = 0; q < FM_ITEMS; ++q) k[q] = 0;

  // ---- tickets: three in flight (ticket -> offsets -> keys -> sort) -------------------------------------------------------
  if (tid == 0) {
    sh[4] = (long long)min(atomicAdd(&header[FS_FTICKET], 1ull), (unsigned long long)n_buckets);     // (ids past the end: all "n_buckets")
    sh[5] = (long long)min(atomicAdd(&header[FS_FTICKET], 1ull), (unsigned long long)n_buckets);
    sh[6] = (long long)min(atomicAdd(&header[FS_FTICKET], 1ull), (unsigned long long)n_buckets);
  }
  __syncthreads();
  int b = (int)fm_uniform(sh[4]), b_nxt = (int)fm_uniform(sh[5]), b_n2 = (int)fm_uniform(sh[6]);
  int64_t f0, f1;
  fetch_offsets(b, f0, f1);
  bucket_t cur = open_bucket(b, f0, f1);
  load_keys(cur);
  fetch_offsets(b_nxt, f0, f1);

  // ---- the buckets on their way out -------------------------------------------------------------------------------------
  // q[FM_DEFER - 1] was finished in the last iteration (its D is published), ..., q[0] leaves in this one: its output base
  // was completed at the end of the last iteration, when it had been published for FM_DEFER - 1 iterations.
  fm_pending q[FM_DEFER];
#pragma unroll
  for (int i = 0; i < FM_DEFER; ++i) q[i] = fm_pending{0, 0, 0, 0, 0u, 0};
  int cur_slot = 0;

  // ---- the prefix of the distinct counts ------------------------------------------------------------------------------
  // Status words [0, pref_b) are summed up in pref_v (every wavefront keeps the same copy).  A poll covers the next
  // 64 * FM_WAVES * FM_LB words: wavefront w loads chunks w * FM_LB ... of 64 words (FM_LB registers per lane, in flight
  // during the neighbour walks), reports per chunk "all there, and their sum" through LDS, and wavefront 0 adds up the
  // leading chunks that are complete.  What is still missing then — a predecessor that is late, or a workgroup that fell
  // more than a poll behind — wavefront 0 fetches chunk by chunk on its own.
  int pref_b = 0;                                        // (32-bit on purpose: hipcc 7.2 drops the VCC -> SCC copy of a uniform 64-bit
  long long pref_v = 0;                                  //  select whose compare also feeds a branch — scripts/check_scc.py)
  unsigned lbv[FM_LB];
#pragma unroll
  for (int i = 0; i < FM_LB; ++i) lbv[i] = 1u;
  auto lb_issue = [&](int target) {                      // every wavefront
    if (FM_ABL & 1) return;
    const unsigned* first = status + pref_b + wave * (FM_LB * 64);
    const int l = fm_fresh(lane);
#pragma unroll
    for (int i = 0; i < FM_LB; ++i)
      lbv[i] = pref_b + wave * (FM_LB * 64) + 64 * i + l < target
                   ? __hip_atomic_load(first + (l + 64 * i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1u;
  };
  auto lb_report = [&]() {                               // every wavefront, before a barrier
    if (FM_ABL & 1) return;
#pragma unroll
    for (int i = 0; i < FM_LB; ++i) {
      const bool there = __all(lbv[i] != 0u);
      const unsigned sum = wave_sum(lbv[i] - 1u);
      if (lane == 0) lbs[wave * FM_LB + i] = there ? sum + 1u : 0u;
    }
  };
  auto lb_finish = [&](int target, int64_t lo) -> bool { // wavefront 0, behind that barrier; false: gave up
    if (FM_ABL & 1) { pref_b = target; pref_v = lo; return true; }
    for (int c = 0; c < FM_WAVES * FM_LB && pref_b < target; ++c) {
      const unsigned v = (unsigned)__builtin_amdgcn_readfirstlane((int)lbs[c]);
      if (v == 0u) break;
      pref_v += (long long)(v - 1u);
      pref_b = min(pref_b + 64, target);
    }
#ifdef FM_COUNT_WAITS
    if (lane == 0) atomicAdd(&header[72 + (pref_b >= target ? 0 : 1)], 1ull);
#endif
    unsigned spins = 0;
    while (pref_b < target) {
      const unsigned v = pref_b + lane < target ? __hip_atomic_load(status + pref_b + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1u;
      if (__all(v != 0u)) {
        pref_v += (long long)wave_sum(v - 1u);
        pref_b = min(pref_b + 64, target);
      } else {
        if (++spins > FM_SPIN_LIMIT) return false;
        __builtin_amdgcn_s_sleep(4);
      }
    }
    return true;
  };
  // wavefront 0 hands the result to the others (sh[3] base, sh[7] how far the prefix reaches), who take it behind a barrier
  auto lb_share = [&](bool ok) {
    if (tid == 0) {
      sh[3] = pref_v;
      sh[7] = (long long)pref_b;
      if (!ok) { sh[1] = 1; atomicOr(&header[FS_FLAGS], 2ull); }
    }
  };
  auto lb_take = [&]() {
    pref_v = fm_uniform(sh[3]);
    pref_b = (int)fm_uniform(sh[7]);
  };

  // ---- emission of the leaving bucket q[0] at its final place, in pieces --------------------------------------------------
  // (all of a bucket's output issued at once is a burst of 90 KB of stores that blocks the issuing wavefronts while the CU
  // drains it — DESIGN §4a; so a wavefront emits three chunks of 64 positions of its range behind each group of its walks)
  unsigned e_r0 = 0;                                     // first occurrences before the next chunk this wavefront emits
  int64_t e_base = 0;
  uint64_t e_carry = ~0ull;                              // the key in front of the next chunk this wavefront emits
  auto emit_begin = [&](int64_t base) {
    const fm_pending& p = q[0];
    e_base = base;
    if (p.kind == 1) {
      const unsigned* dc = dupc + p.slot * FM_WAVES;
      unsigned r0 = 0;                                   // first occurrences in the ranges of the wavefronts before this one
#pragma unroll
      for (int w = 0; w < FM_WAVES; ++w) {
        const int in_range = max(0, min(FM_SLICE, p.nb - w * FM_SLICE));
        r0 += w < wave ? (unsigned)in_range - dc[w] : 0u;
      }
      e_r0 = (unsigned)__builtin_amdgcn_readfirstlane((int)r0);
      // the key in front of this wavefront's range (the sentinel in front of the slot for wavefront 0)
      const uint64_t c0 = mypark[p.slot * FM_PARK + FM_NEAR + min(wave * FM_SLICE, p.nb) - 1];
      e_carry = (uint64_t)fm_uniform((int64_t)c0);
    } else if (p.kind == 2) {
      uint64_t* ko = keys_out + base;
      int64_t* co = counts_out + base;
      for (unsigned i = (unsigned)fm_fresh(tid); i < p.D; i += FM_THREADS) {
        ko[i] = big_keys[p.src + i];
        co[i] = big_counts[p.src + i];
      }
    } else if (p.kind == 3) {
      if (tid == 0) {
        const unsigned long long at = atomicAdd(&header[FS_REDO], 1ull);
        redo_ids[at] = (unsigned)p.b;
        redo_bases[at] = base;
      }
    }
    if (p.kind != 0 && p.b == nbk - 1 && tid == 0) header[FS_UNIQUE] = (unsigned long long)(base + (int64_t)p.D);
  };
  // One piece = FM_WG chunks of 64 sorted positions.  All loads of a piece are issued together — the keys of its chunks and
  // of the chunk behind them —, everything else is register work: the key in front of a position comes from the lane
  // below (DPP wave shift; lane 0 takes the last key of the chunk before, carried from piece to piece), a position is a
  // BOUNDARY if it lies behind the bucket or its key differs from the one in front, a boundary inside the bucket is a first
  // occurrence, and its count is the distance to the next boundary (in this chunk's mask, else in the next chunk's: runs
  // are at most 64 long).  (The first version loaded key, predecessor and successor per chunk and waited for them chunk by
  // chunk: on gfx9 that wait counts the stores of the chunk before, 16 ms of 41.)
  auto emit_piece = [&](int c_lo) {
    const fm_pending& p = q[0];
    if (p.kind != 1) return;
    const int nbp = p.nb;
    const int start = wave * FM_SLICE + c_lo * 64;       // (uniform)
    if (start >= nbp) return;
    uint64_t* ko = keys_out + e_base;
    int64_t* co = counts_out + e_base;
    const int l0 = fm_fresh(lane);
    const uint64_t* mid = mypark + p.slot * FM_PARK + FM_NEAR + start + l0;
    uint64_t K[FM_WG + 1];
#pragma unroll
    for (int u = 0; u <= FM_WG; ++u) K[u] = start + 64 * u < nbp ? mid[64 * u] : ~0ull;     // (uniform conditions)
    unsigned long long B[FM_WG + 1];
    uint64_t carry = e_carry;
#pragma unroll
    for (int u = 0; u <= FM_WG; ++u) {
      const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)carry, (int)(unsigned)K[u], 0x138, 0xf, 0xf, false);
      const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(carry >> 32), (int)(unsigned)(K[u] >> 32), 0x138, 0xf, 0xf, false);
      const uint64_t before = ((uint64_t)hi << 32) | lo;
      B[u] = __ballot(start + 64 * u + l0 >= nbp || K[u] != before);
      const unsigned clo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)K[u], 63);
      const unsigned chi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(K[u] >> 32), 63);
      if (u == FM_WG - 1) e_carry = ((uint64_t)chi << 32) | clo;
      carry = ((uint64_t)chi << 32) | clo;
    }
#pragma unroll
    for (int u = 0; u < FM_WG; ++u) {
      if (start + 64 * u < nbp) {                        // (uniform)
        const int pos = start + 64 * u + l0;
        const bool first = pos < nbp && ((B[u] >> l0) & 1ull);
        const unsigned long long rest = (B[u] >> l0) >> 1;
        const unsigned next_in = B[u + 1] ? (unsigned)__builtin_ctzll(B[u + 1]) : 0u;
        const unsigned cnt = rest ? (unsigned)__builtin_ctzll(rest) + 1u : (unsigned)(64 - l0) + next_in;
        const unsigned long long m = __ballot(first);
        const unsigned r = e_r0 + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (first && !(FM_ABL & 4)) {
          __builtin_nontemporal_store(K[u], &ko[r]);
          __builtin_nontemporal_store((int64_t)cnt, &co[r]);
        }
        e_r0 += (unsigned)__popcll(m);
      }
    }
  };
  auto emit_all = [&]() {
#pragma unroll 1
    for (int c = 0; c < FM_ITEMS; c += FM_WG) emit_piece(c);
  };
  auto publish = [&](int bb, unsigned D) {               // one lane
    __hip_atomic_store(status + bb, D + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // end of an iteration, behind the barrier that follows the publication of `now`: wavefront 0 completes the prefix of the
  // bucket that leaves next (every wavefront has reported its part of the poll before that barrier) and shares it
  // — taken by everybody behind the next barrier; then the pending buckets move up.
  auto rotate = [&](const fm_pending& now) {
    if (wave == 0) lb_share(q[1].kind == 0 || lb_finish(q[1].b, q[1].src));
#pragma unroll
    for (int i = 0; i + 1 < FM_DEFER; ++i) q[i] = q[i + 1];
    q[FM_DEFER - 1] = now;
  };
  __syncthreads();
  for (; b < nbk;) {
    const bucket_t nxt = open_bucket(b_nxt, f0, f1);     // bucket b_nxt (offsets fetched an iteration ago)
    fetch_offsets(b_n2, f0, f1);
    unsigned long long tk = 0;
    if (tid == 0) tk = atomicAdd(&header[FS_FTICKET], 1ull);
    int b_n3 = 0;
    const int nb = cur.nb;
    fm_pending now = {0, b, nb, cur_slot, 0u, (FM_ABL & 1) ? cur.lo : 0};
    if (nb == 0) {
      // ---- an empty bucket, or a heavy-hitter bucket the caller counted beforehand: nothing to sort ------------------
      if (cur.size > 0) {
        int lo_i = 0, hi_i = n_big;
        while (lo_i < hi_i) {
          const int mid_i = (lo_i + hi_i) >> 1;
          if (big_table[3 * mid_i] < b) lo_i = mid_i + 1; else hi_i = mid_i;
        }
        if (lo_i < n_big && big_table[3 * lo_i] == b) {
          now.D = (unsigned)fm_uniform(big_table[3 * lo_i + 1]);
          if (!(FM_ABL & 1)) now.src = fm_uniform(big_table[3 * lo_i + 2]);
          now.kind = 2;
        } else if (tid == 0) {
          atomicOr(&header[FS_FLAGS], 1ull);
        }
      }
      if (now.kind == 0 && b == nbk - 1) now.kind = 4;   // (nothing to write, but the last bucket reports the total)
      if (tid == 0) publish(b, now.D);
      if (tid == 0) sh[2] = (long long)min(tk, (unsigned long long)n_buckets);
      __syncthreads();                                   // (the prefix wavefront 0 shared at the end of the last iteration is there)
      b_n3 = (int)fm_uniform(sh[2]);
      if (fm_uniform(sh[1])) return;
      lb_take();
      emit_begin(pref_v);
      emit_all();
      if (q[1].kind != 0) { lb_issue(q[1].b); lb_report(); }
      load_keys(nxt);
      __syncthreads();
      rotate(now);
    } else {
      // ---- counting sort on the next sbits bits (as in finish_fast_kernel) ------------------------------------------------
      unsigned rb[FM_ITEMS];
      const bool large = nb > FM_USUAL * FM_THREADS;
      {
        const int t0 = fm_fresh(tid);
        unsigned old[FM_ITEMS];
#pragma unroll
        for (int q = 0; q < FM_USUAL; ++q) {
          const unsigned bin = (unsigned)(k[q] >> sshift) & (SB - 1);
          rb[q] = bin;
          old[q] = atomicAdd(&P32[bin >> 1], (t0 + q * FM_THREADS < nb ? 1u : 0u) << ((bin & 1u) * 16u));
        }
        if (large) {
#pragma unroll
          for (int q = FM_USUAL; q < FM_ITEMS; ++q) {
            const unsigned bin = (unsigned)(k[q] >> sshift) & (SB - 1);
            rb[q] = bin;
            old[q] = atomicAdd(&P32[bin >> 1], (t0 + q * FM_THREADS < nb ? 1u : 0u) << ((bin & 1u) * 16u));
          }
        } else {
#pragma unroll
          for (int q = FM_USUAL; q < FM_ITEMS; ++q) { rb[q] = 0; old[q] = 0; }
        }
#pragma unroll
        for (int q = 0; q < FM_ITEMS; ++q) rb[q] |= __builtin_amdgcn_ubfe(old[q], (rb[q] & 1u) * 16u, 16u) << 13;
      }
      __syncthreads();                                   // (1) every rank is taken; the prefix of the leaving bucket is there
      if (fm_uniform(sh[1])) return;                     // (uniform) a wait gave up: the caller falls back
      lb_take();
      emit_begin(pref_v);
      bool short_bins;
      int t_bucket;                                        // the bucket's longest bin - 1
      {
        unsigned c[FM_SCAN_DW], sum = 0, longest = 0;
        const unsigned t1 = (unsigned)fm_fresh(tid);
#pragma unroll
        for (int j = 0; j < FM_SCAN_DW; ++j) {
          const unsigned w = t1 * dwl + j;
          c[j] = ((unsigned)j < dwl && w < n_dw) ? P32[w] : 0u;
          sum += (c[j] & 0xffffu) + (c[j] >> 16);
          longest = max(longest, max(c[j] & 0xffffu, c[j] >> 16));
        }
        const unsigned inc = wave_inclusive_scan(sum);
        longest = wave_max(longest);
        if (lane == 63) { wsum[wave] = inc; wsum[2 * FM_WAVES + wave] = longest; }
        __syncthreads();                                 // (2)
        unsigned run = inc - sum, longs = 0;
#pragma unroll
        for (int w = 0; w < FM_WAVES; ++w) {
          run += w < wave ? wsum[w] : 0u;
          longs = max(longs, wsum[2 * FM_WAVES + w]);
        }
        t_bucket = __builtin_amdgcn_readfirstlane((int)longs) - 1;
        short_bins = t_bucket < FM_NEAR;
        if (FM_ABL & 32) t_bucket = min(t_bucket, 1);      // (experiment: what the neighbour walks cost — wrong results)
#pragma unroll
        for (int j = 0; j < FM_SCAN_DW; ++j) {
          const unsigned w = t1 * dwl + j;
          if ((unsigned)j < dwl && w < n_dw) {
            const unsigned c0 = c[j] & 0xffffu;
            P32[w] = run | ((run + c0) << 16);
            run += c0 + (c[j] >> 16);
          }
        }
        if (tid == 0) reinterpret_cast<unsigned short*>(P32)[SB] = (unsigned short)nb;
      }
      __syncthreads();                                   // (3) the bin offsets are in place
      {
        const int t2 = fm_fresh(tid);
        unsigned slot[FM_ITEMS];
#pragma unroll
        for (int q = 0; q < FM_USUAL; ++q) slot[q] = P16[rb[q] & 0x1fffu] + (rb[q] >> 13);
#pragma unroll
        for (int q = 0; q < FM_USUAL; ++q)
          if (t2 + q * FM_THREADS < nb) stage[slot[q]] = k[q];
        if (large) {
#pragma unroll
          for (int q = FM_USUAL; q < FM_ITEMS; ++q) slot[q] = P16[rb[q] & 0x1fffu] + (rb[q] >> 13);
#pragma unroll
          for (int q = FM_USUAL; q < FM_ITEMS; ++q)
            if (t2 + q * FM_THREADS < nb) stage[slot[q]] = k[q];
        }
      }
      if (wave == FM_WAVES - 1) stage[nb + fm_fresh(lane)] = ~0ull;   // guard keys behind the bucket
      if (tid < FM_WAVES) dupc[cur_slot * FM_WAVES + tid] = 0;
      if (tid == 0) sh[2] = (long long)min(tk, (unsigned long long)n_buckets);
      load_keys(nxt);
      if (q[1].kind != 0) lb_issue(q[1].b);              // (uniform) in flight during the walks
      __syncthreads();                                   // (4) the keys are grouped by bin
      b_n3 = (int)fm_uniform(sh[2]);
      // ---- slot owners: place = s - min(t, s) + smaller keys within t on either side + equal keys within t on the left;
      // the key goes to that place of the parking slot
      unsigned ndup = 0;
      const int l3 = fm_fresh(lane);
      const int slice0 = wave * FM_SLICE;
      now.kind = 1;
      if (short_bins) {
        if (!FM_GROUP_WALK)
          for (unsigned i = (unsigned)fm_fresh(tid); i <= n_dw; i += FM_THREADS) P32[i] = 0;
        uint64_t* pk = mypark + cur_slot * FM_PARK + FM_NEAR;
        unsigned* dc = dupc + cur_slot * FM_WAVES;
        if (tid == 0) pk[nb] = ~0ull;                    // the sentinel behind the bucket
#pragma unroll
        for (int c0 = 0; c0 < FM_ITEMS; c0 += FM_WG) {
          if (slice0 + c0 * 64 < nb) {
            uint64_t x[FM_WG];
            unsigned cnt[FM_WG], eq[FM_WG];
            const int sl0 = slice0 + c0 * 64 + l3;
            const uint64_t* mid = stage + sl0;
#pragma unroll
            for (int u = 0; u < FM_WG; ++u) { x[u] = mid[64 * u]; cnt[u] = 0; eq[u] = 0; }
            // how far THIS group's keys can reach: the longest bin among its 192 slots' bins, not the bucket's longest.  On reads at
            // 3x coverage the bucket's longest bin holds ~14 keys (a k-mer that occurs ten times and its bin mates), a group's ~9
            // (NOTES round 6): two 16-bit reads per slot and one wave reduction for five steps less of the walk below.
            int t_walk = t_bucket;
            if (FM_GROUP_WALK) {
              unsigned len = 0;
#pragma unroll
              for (int u = 0; u < FM_WG; ++u) {
                const unsigned bin = (unsigned)(x[u] >> sshift) & (SB - 1);
                const unsigned l = (unsigned)P16[bin + 1] - (unsigned)P16[bin];
                len = max(len, sl0 + 64 * u < nb ? l : 1u);
              }
              t_walk = min(t_bucket, __builtin_amdgcn_readfirstlane((int)wave_max(len)) - 1);
            }
#pragma unroll FM_UNROLL
            for (int d = 1; d <= t_walk; ++d) {
              uint64_t y[FM_WG], z[FM_WG];
#pragma unroll
              for (int u = 0; u < FM_WG; ++u) { y[u] = mid[64 * u - d]; z[u] = mid[64 * u + d]; }
#pragma unroll
              for (int u = 0; u < FM_WG; ++u) {
                cnt[u] += (y[u] < x[u] ? 1u : 0u) + (z[u] < x[u] ? 1u : 0u);
                eq[u] += y[u] == x[u] ? 1u : 0u;
              }
            }
#pragma unroll
            for (int u = 0; u < FM_WG; ++u) {
              const int s = sl0 + 64 * u;
              const unsigned place = (unsigned)(s - min(t_walk, s)) + cnt[u] + eq[u];
              const bool rep = eq[u] != 0 && s < nb;
              ndup += rep ? 1u : 0u;
              if (rep) atomicAdd(&dc[place / FM_SLICE], 1u);
              if (s < nb) pk[place] = x[u];
            }
          }
          emit_piece(c0);                                // the leaving bucket, three chunks at a time
        }
      } else {
        // a bin longer than a chunk: only the exact number of repeats is taken here; the general kernel sorts the bucket
        now.kind = 3;
#pragma unroll 1
        for (int c = 0; c < FM_ITEMS; ++c) {
          const int s = slice0 + c * 64 + l3;
          if (slice0 + c * 64 >= nb) break;
          unsigned a = 0;
          uint64_t xv = 0;
          if (s < nb) {
            xv = stage[s];
            a = (unsigned)s - P16[(unsigned)(xv >> sshift) & (SB - 1)];
          }
          bool is_dup = false;
          for (unsigned d = 1; __any(d <= a && !is_dup); ++d)
            if (d <= a && !is_dup && stage[s - (int)d] == xv) is_dup = true;
          ndup += is_dup ? 1u : 0u;
        }
        emit_all();
      }
      ndup = wave_sum(ndup);
      if (lane == 0) wsum[FM_WAVES + wave] = ndup;
      if (q[1].kind != 0) lb_report();                   // (uniform)
      __syncthreads();                                   // (5) every neighbour has been read; the bucket is parked
      unsigned dups = 0;
#pragma unroll
      for (int w = 0; w < FM_WAVES; ++w) dups += wsum[FM_WAVES + w];
      dups = (unsigned)__builtin_amdgcn_readfirstlane((int)dups);
      now.D = (unsigned)nb - dups;
      if (tid == 0) publish(b, now.D);
      if (!short_bins || FM_GROUP_WALK) {                  // (the walks read the bins until barrier 5)
        for (unsigned i = (unsigned)fm_fresh(tid); i <= n_dw; i += FM_THREADS) P32[i] = 0;
        __syncthreads();
      }
      rotate(now);                                       // (wavefront 0 may wait here — AFTER the publication: nobody waits for it in turn)
      cur_slot = cur_slot + 1 == FM_SLOTS ? 0 : cur_slot + 1;
    }
    cur = nxt;
    b = b_nxt;
    b_nxt = b_n2;
    b_n2 = b_n3;
  }
  // ---- the buckets still on their way out ---------------------------------------------------------------------------------
  __syncthreads();
  if (fm_uniform(sh[1])) return;
  lb_take();
  emit_begin(pref_v);
  emit_all();
#pragma unroll
  for (int i = 1; i < FM_DEFER; ++i) {
    if (q[i].kind != 0) {
      q[0] = q[i];
      lb_issue(q[0].b);
      lb_report();
      __syncthreads();
      if (wave == 0) lb_share(lb_finish(q[0].b, q[0].src));
      __syncthreads();
      if (fm_uniform(sh[1])) return;
      lb_take();
      emit_begin(pref_v);
      emit_all();
      __syncthreads();
    }
  }
}

}  // namespace

int64_t bnpk_finish_multi_park_bytes(int grid) { return (int64_t)grid * FM_SLOTS * FM_PARK * 8; }

int bnpk_finish_multi_launch(bnpk_ctx* ctx, const uint64_t* part, const int64_t* bucket_off, int64_t n_buckets, int low_bits,
                             unsigned long long* header, unsigned* status, uint64_t* keys_out, int64_t* counts_out,
                             const int64_t* big_table, int n_big, const uint64_t* big_keys, const int64_t* big_counts,
                             unsigned* redo_ids, int64_t* redo_bases, int64_t pstride, hipStream_t s) {
  if (n_buckets >= (1ll << 30)) return BNPK_ERR_RANGE;
  if (!ctx->finish_multi_ready) {
    BNPK_HIP(ctx, hipFuncSetAttribute((const void*)finish_multi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FM_LDS));
    int per_cu = 0;
    BNPK_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)finish_multi_kernel, FM_THREADS, FM_LDS));
    ctx->finish_multi_grid = ctx->compute_units * std::max(1, per_cu);
    ctx->finish_multi_ready = true;
  }
  const int sbits = std::min(low_bits, FM_MAXBITS), sshift = low_bits - sbits;
  const unsigned grid = (unsigned)std::min<int64_t>(n_buckets, (int64_t)ctx->finish_multi_grid);
  void* park = nullptr;                                  // two parking slots per workgroup (the arena is the caller's until its next bnpk_scratch)
  BNPK_CHECK(bnpk_scratch(ctx, (size_t)bnpk_finish_multi_park_bytes((int)grid), &park, s));
  hipLaunchKernelGGL(finish_multi_kernel, dim3(grid), dim3(FM_THREADS), FM_LDS, s, part, bucket_off, n_buckets, sshift, sbits,
                     header, status, keys_out, counts_out, big_table, n_big, big_keys, big_counts, redo_ids, redo_bases,
                     (uint64_t*)park, pstride);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}
