// Duplicate-aware finishing, one WAVEFRONT per bucket: the path of buckets with few distinct keys — the k-mers of reads
// that cover a genome many times over (SURVEY §8(d)'s S-genome: 6e9 keys, 1e8 distinct, ~95 distinct keys per bucket).
//
// finish_dup.hip does this with a workgroup per bucket and three barriers per bucket; measured, it is bound by neither
// HBM (its loads alone run at 4.9 TB/s), nor the LDS atomics (a third of their rate), nor the VALU — but by its
// phases: insert, rank, clear, each behind a barrier, with two workgroups per CU to fill the gaps.  Here nobody waits
// for anybody.  A wavefront owns a small table (704 slots) of its own, streams a bucket's keys through it — 16-byte
// loads, three requests of 4 keys per lane in flight, running on across bucket boundaries — and ranks, emits and clears
// the table's entries by itself: the LDS executes one wavefront's operations in order, so there is not a single barrier
// in the kernel, and sixteen wavefronts per CU drift apart until the chip's memory, LDS and ALU work overlap.
//
// The table is finish_dup.hip's: open addressing, home slot = the key's leading free bits scaled to the table
// (monotone), linear probing, so clusters ascend and an entry's place is the number of used slots before it, corrected
// inside its cluster by comparing neighbours.  Used slots are listed as they are claimed; a bitmap of the used slots,
// one word per lane, gives the rank (popcount scan in registers, two DS_BPERMUTEs per entry).
// Outputs are "loose" as in finish_dup.hip: keys back over the bucket's own keys, counts to the same positions of the
// future key array, Dv[b] = distinct keys; a bucket this table cannot hold is left — untouched — to the next kernel
// through a list (finish_dup_kernel, then the general one).  Whether the keys are duplicate-heavy at all is found out
// beforehand by the same kernel in PROBE mode: a sample of the buckets goes through the table, nothing is written but
// the number of buckets it could not hold.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "finish.h"

namespace {

#ifndef FW_ABL
#define FW_ABL 0                                         // experiments: 1 = no loads after the first buckets, 2 = no inserts
#endif
constexpr int FW_CHUNKS = 11;
constexpr int FW_TS = 64 * FW_CHUNKS;                    // table slots of one wavefront (704)
constexpr int FW_PROBES = 32;                            // longest probe sequence; beyond it the bucket is left to the next kernel
constexpr int FW_HOMES = (FW_TS - FW_PROBES - 8) / 2;    // home pairs: slots [0, 2 FW_HOMES) — the last slots of the table stay empty
constexpr int FW_GUARD = 2;                              // empty slots in front of slot 0 (the left walks end there)
constexpr int FW_HBITS = 16;                             // leading free key bits that make up the home slot
constexpr int FW_BM_WORDS = FW_TS / 32;                  // bitmap of the used slots: one word per lane (22)
constexpr int FW_LIST = 448;                             // used slots listed in the order they were claimed
constexpr int FW_RING = 3;                               // requests in flight
constexpr int FW_ITEM = 256;                             // keys per request: two 16-byte loads per lane
constexpr unsigned long long FW_EMPTY = ~0ull;           // (keys are < 2^63)
static_assert(FW_BM_WORDS <= 64 && FW_HOMES < (1 << 24) && FW_HBITS <= 16, "one bitmap word per lane; 24-bit home product");
constexpr size_t FW_OFF_T = (size_t)FW_GUARD * 8;
static_assert(FW_OFF_T % 16 == 0, "the home pairs are read as 16 aligned bytes");
constexpr size_t FW_OFF_C = FW_OFF_T + (size_t)FW_TS * 8;               // 32-bit counts, one per slot
constexpr size_t FW_OFF_BM = FW_OFF_C + (size_t)FW_TS * 4;
constexpr size_t FW_OFF_LIST = FW_OFF_BM + (size_t)FW_BM_WORDS * 4;
constexpr size_t FW_OFF_SH = FW_OFF_LIST + (size_t)FW_LIST * 2;         // [0] used slots
constexpr size_t FW_LDS = (FW_OFF_SH + 8 + 63) & ~(size_t)63;
static_assert(16 * FW_LDS <= 160 * 1024, "sixteen wavefronts per CU");

__device__ __forceinline__ int64_t fw_uniform(int64_t v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uint64_t)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}
template <int N> struct fw_int { static constexpr int value = N; };

struct fw_item {           // one request: keys [c, c + FW_ITEM) of bucket b (all fields wave-uniform)
  int64_t b, lo, size;
  int nb, c;
  int64_t src;             // where the bucket's keys lie in A: lo, or b * pstride (finish.h)
};
typedef unsigned long long fw_v2 __attribute__((ext_vector_type(2), aligned(8)));

// A bucket is in its wavefront's table: rank, emit and clear the entries (or leave the bucket to the next kernel).
template <bool PROBE>
__device__ __noinline__ void fw_finalize(uint64_t* A, int64_t b, int64_t lo, int64_t src_pos, int64_t size, int nb, bool gave_up,
                                         unsigned long long* __restrict__ header, int64_t* __restrict__ Dv,
                                         unsigned* __restrict__ todo_ids, int64_t* __restrict__ loose_counts,
                                         const int64_t* __restrict__ big_table, int n_big,
                                         const uint64_t* __restrict__ big_keys, const int64_t* __restrict__ big_counts) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* T = reinterpret_cast<unsigned long long*>(smem + FW_OFF_T);
  unsigned* C = reinterpret_cast<unsigned*>(smem + FW_OFF_C);
  unsigned* BM = reinterpret_cast<unsigned*>(smem + FW_OFF_BM);
  unsigned short* L = reinterpret_cast<unsigned short*>(smem + FW_OFF_LIST);
  unsigned* sh = reinterpret_cast<unsigned*>(smem + FW_OFF_SH);
  const int lane = threadIdx.x;
  struct { int64_t b, lo, size; } it = {b, lo, size};
    // ---- the bucket is in the table
    if (nb == 1) {                                       // a bucket of one key: it lies where it belongs already
      if (PROBE) return;
      if (lane == 0) { loose_counts[lo] = 1; Dv[b] = 1; }
      return;
    }
    if (nb == 0) {                                       // empty, or a heavy-hitter bucket counted by the caller beforehand
      if (PROBE) return;
      unsigned D = 0;
      if (it.size > 0) {
        int lo_i = 0, hi_i = n_big;
        while (lo_i < hi_i) {
          const int mid = (lo_i + hi_i) >> 1;
          if (big_table[3 * mid] < b) lo_i = mid + 1; else hi_i = mid;
        }
        if (lo_i < n_big && big_table[3 * lo_i] == b) {
          D = (unsigned)fw_uniform(big_table[3 * lo_i + 1]);
          const int64_t src = fw_uniform(big_table[3 * lo_i + 2]);
          uint64_t* ko = A + src_pos;
          int64_t* co = loose_counts + it.lo;
          for (unsigned i = (unsigned)lane; i < D; i += 64) {
            ko[i] = big_keys[src + i];
            co[i] = big_counts[src + i];
          }
        } else if (lane == 0) {
          atomicOr(&header[FS_FLAGS], 1ull);
        }
      }
      if (lane == 0) Dv[b] = D;
      return;
    }
    const unsigned D = (unsigned)__builtin_amdgcn_readfirstlane((int)sh[0]);      // used slots = distinct keys
    const bool bad = gave_up || D > (unsigned)FW_LIST;
    if (PROBE) {
      if (bad) { for (int i = lane; i < FW_TS; i += 64) { T[i] = FW_EMPTY; C[i] = 0; } }
      else for (unsigned j = (unsigned)lane; j < D; j += 64) { const int s = (int)L[j]; T[s] = FW_EMPTY; C[s] = 0; }
      if (lane < FW_BM_WORDS) BM[lane] = 0;
      if (lane == 0) {
        sh[0] = 0;
        if (bad && !gave_up) atomicAdd(&header[FS_PROBE_GIVEUP], (unsigned long long)nb);   // (more distinct keys than the list holds)
        atomicAdd(&header[FS_PROBE_KEYS], (unsigned long long)nb);
        atomicAdd(&header[bad ? FS_PROBE_BAD : FS_PROBE_DISTINCT], bad ? 1ull : (unsigned long long)D);
      }
      return;
    }
    if (!bad) {
      uint64_t* ko = A + src_pos;                          // scalar bases, 32-bit lane offsets
      int64_t* co = loose_counts + it.lo;
      // used slots before a slot: lane l holds bitmap word l and the used slots before it
      const unsigned word = lane < FW_BM_WORDS ? BM[lane] : 0u;
      const unsigned pc = (unsigned)__popc(word);
      const unsigned before = wave_inclusive_scan(pc) - pc;
      for (unsigned j0 = 0; j0 < D; j0 += 64) {
        const bool used = j0 + (unsigned)lane < D;
        const int s = used ? (int)L[j0 + (unsigned)lane] : 0;
        const unsigned long long x = T[s], y = T[s - 1], z = T[s + 1];
        const unsigned cnt = C[s];
        const unsigned w = (unsigned)s >> 5;
        unsigned place = (unsigned)__shfl((int)before, (int)w, 64) +
                         (unsigned)__popc((unsigned)__shfl((int)word, (int)w, 64) & ((1u << ((unsigned)s & 31u)) - 1u));
        // inside its cluster: larger entries on its left come after it, smaller ones on its right before it
        bool go_l = used && y != FW_EMPTY, go_r = used && z != FW_EMPTY;
        unsigned down = (go_l && y > x) ? 1u : 0u, up = (go_r && z < x) ? 1u : 0u;
        for (int d = 2; __any(go_l || go_r); ++d) {
          unsigned long long yy = FW_EMPTY, zz = FW_EMPTY;
          if (go_l) yy = T[s - d];
          if (go_r) zz = T[s + d];
          go_l = go_l && yy != FW_EMPTY;
          go_r = go_r && zz != FW_EMPTY;
          down += (go_l && yy > x) ? 1u : 0u;
          up += (go_r && zz < x) ? 1u : 0u;
        }
        place = place - down + up;
        if (used) {
          ko[place] = x;
          co[place] = (int64_t)cnt;
        }
      }
      for (unsigned j0 = 0; j0 < D; j0 += 64) {          // (the LDS is in order: every neighbour has been read)
        if (j0 + (unsigned)lane < D) {
          const int s = (int)L[j0 + (unsigned)lane];
          T[s] = FW_EMPTY;
          C[s] = 0;
        }
      }
      if (lane < FW_BM_WORDS) BM[lane] = 0;
      if (lane == 0) { Dv[b] = D; sh[0] = 0; }
    } else {
      // not this kernel's bucket: its keys are untouched, the next kernel finds it in the list
      for (int i = lane; i < FW_TS; i += 64) { T[i] = FW_EMPTY; C[i] = 0; }
      if (lane < FW_BM_WORDS) BM[lane] = 0;
      if (lane == 0) {
        sh[0] = 0;
        todo_ids[atomicAdd(&header[FS_TODO], 1ull)] = (unsigned)b;
      }
    }
}

// HI: the home bits lie in the key's high word (sshift >= 32); otherwise they straddle the words (one V_ALIGNBIT)
// PROBE: the buckets 0, stride, 2 stride, ... only, and no output but header[FS_PROBE_BAD] / header[FS_PROBE_KEYS]
template <bool HI, bool PROBE>
__global__ __launch_bounds__(64) void finish_wave_kernel(
    uint64_t* A, int64_t n_total, const int64_t* __restrict__ bucket_off, int64_t n_buckets, int64_t stride, int sshift, int sbits,
    unsigned long long* __restrict__ header, int64_t* __restrict__ Dv, unsigned* __restrict__ todo_ids,
    int64_t* __restrict__ loose_counts, const int64_t* __restrict__ big_table, int n_big,
    const uint64_t* __restrict__ big_keys, const int64_t* __restrict__ big_counts, int64_t pstride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* T = reinterpret_cast<unsigned long long*>(smem + FW_OFF_T);
  unsigned* C = reinterpret_cast<unsigned*>(smem + FW_OFF_C);
  unsigned* BM = reinterpret_cast<unsigned*>(smem + FW_OFF_BM);
  unsigned short* L = reinterpret_cast<unsigned short*>(smem + FW_OFF_LIST);
  unsigned* sh = reinterpret_cast<unsigned*>(smem + FW_OFF_SH);
  const int lane = threadIdx.x;
  const unsigned hmask = (1u << sbits) - 1u;
  const unsigned hshift = (unsigned)(HI ? sshift - 32 : sshift);
  const int64_t G = (int64_t)gridDim.x * stride;

  for (int i = lane - FW_GUARD; i < FW_TS; i += 64) T[i] = FW_EMPTY;
  for (int i = lane; i < FW_TS; i += 64) C[i] = 0;
  if (lane < FW_BM_WORDS) BM[lane] = 0;
  if (lane < 2) sh[lane] = 0;

  // ---- the request stream: buckets blockIdx, blockIdx + gridDim, ... (times the stride); every bucket in requests of FW_ITEM keys (an empty or
  // pre-counted bucket is one request without keys); the offsets of the bucket after the current one are on their way
  int64_t sb = (int64_t)blockIdx.x * stride, s_lo = 0, s_size = 0, p_lo = 0, p_hi = 0;
  int s_nb = 0, s_c = 0;
  auto fetch = [&](int64_t bb, int64_t& o0, int64_t& o1) {
    o0 = 0; o1 = 0;
    if (bb < n_buckets) { o0 = bucket_off[bb]; o1 = bucket_off[bb + 1]; }
  };
  auto open = [&](int64_t o0, int64_t o1) {
    s_lo = fw_uniform(o0);
    s_size = fw_uniform(o1) - s_lo;
    s_nb = s_size > FINISH_CAP ? 0 : (int)s_size;
    s_c = 0;
  };
  fetch(sb, p_lo, p_hi);
  open(p_lo, p_hi);
  fetch(sb + G, p_lo, p_hi);
  auto next_item = [&]() -> fw_item {
    fw_item it = {sb, s_lo, s_size, s_nb, s_c, pstride ? sb * pstride : s_lo};
    s_c += FW_ITEM;
    if (s_c >= s_nb) {                                   // (uniform) the bucket is exhausted
      sb += G;
      open(p_lo, p_hi);
      fetch(sb + G, p_lo, p_hi);
    }
    return it;
  };
  // Two 16-byte loads per lane and request, ALWAYS issued (the wait for a request counts the loads behind it): lane l
  // holds the keys c + 128 u + 2 l and + 1; addresses past the bucket's end are clamped to its last pair, where the
  // bucket's last key — at an odd position — arrives as the second element.  Past the last bucket: the array's start.
  auto issue = [&](const fw_item& it, fw_v2 (&r)[2]) {
    const bool in = it.b < n_buckets;
    const uint64_t* base = A + (in ? (it.nb >= 2 ? it.src : std::min<int64_t>(it.src, n_total - 2)) : 0);
    const unsigned span = (in && it.nb >= 2) ? (unsigned)(it.nb - 2) : 0u;
#if FW_ABL & 1
    if (it.b >= 3 * (int64_t)gridDim.x) return;          // (no loads after the first buckets: the registers keep their keys)
#endif
#pragma unroll
    for (int u = 0; u < 2; ++u)
      r[u] = __builtin_nontemporal_load(reinterpret_cast<const fw_v2*>(base + min((unsigned)(it.c + 128 * u + 2 * lane), span)));
  };
  // byte offset of a key's home: a PAIR of slots at an even position (its leading free bits, scaled to the pairs)
  auto home16 = [&](uint64_t key) -> unsigned {
    const unsigned lo = (unsigned)key, hi = (unsigned)(key >> 32);
    const unsigned v = (HI ? hi >> hshift : __builtin_amdgcn_alignbit(hi, lo, hshift)) & hmask;
    return (__umul24(v, (unsigned)FW_HOMES) >> sbits) << 4;
  };
  unsigned char* Tb = reinterpret_cast<unsigned char*>(T);
  unsigned char* Cb = reinterpret_cast<unsigned char*>(C);
#ifdef FW_PHASES
  unsigned long long ph_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_last = __builtin_readcyclecounter();
#define FW_MARK(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long now__ = __builtin_readcyclecounter(); ph_t[i] += now__ - ph_last; ph_last = now__; __builtin_amdgcn_sched_barrier(0); }
#else
#define FW_MARK(i)
#endif
  bool gave_up = false;
  // Four keys per lane go into the table.  Two 16-byte reads show a key's home pair and the pair behind it: a copy of a
  // key that sits there already — every copy but the first, unless five keys met on two pairs — needs no
  // compare-and-swap and no second round trip, just its count.  (With single home slots, one key in seven of a bucket's
  // ~95 lives next to its home and each of its copies probed again; with one pair, one key per bucket still did, and
  // half of all groups of 64 keys held one of its copies and waited for it: 62 % of the kernel's time.)
  // What is not there yet is claimed by compare-and-swaps, the four items of a lane probing in step: one round trip per
  // probe step, as at the start of every bucket, when all 256 keys of the first request are new.
  auto insert4 = [&](const unsigned long long (&key)[4], const bool (&act)[4]) {
    unsigned a8[4];
    uint4 w[4][2];
    bool ok[4], pend[4], fresh[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a8[u] = home16(key[u]);
    FW_MARK(0)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      w[u][0] = *reinterpret_cast<const uint4*>(Tb + a8[u]);
      w[u][1] = *reinterpret_cast<const uint4*>(Tb + a8[u] + 16);
    }
#ifdef FW_PHASES
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    FW_MARK(1)
    bool any_pend = false;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned long long s0 = ((unsigned long long)w[u][0].y << 32) | w[u][0].x, s1 = ((unsigned long long)w[u][0].w << 32) | w[u][0].z;
      const unsigned long long s2 = ((unsigned long long)w[u][1].y << 32) | w[u][1].x, s3 = ((unsigned long long)w[u][1].w << 32) | w[u][1].z;
      ok[u] = !(FW_ABL & 2) && act[u];
      const bool h0 = s0 == key[u], h1 = s1 == key[u], h2 = s2 == key[u], h3 = s3 == key[u];
      pend[u] = ok[u] && !(h0 || h1 || h2 || h3);
      fresh[u] = false;
      a8[u] += h1 ? 8u : h2 ? 16u : h3 ? 24u : 0u;
      any_pend = any_pend || pend[u];
    }
    if (__any(any_pend)) {                               // (uniform) somebody's key is not there (yet)
      for (int probes = 0; probes <= FW_PROBES; ++probes) {
        unsigned long long o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (pend[u]) o[u] = atomicCAS(reinterpret_cast<unsigned long long*>(Tb + a8[u]), FW_EMPTY, key[u]);
        bool more = false;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (pend[u]) {
            if (o[u] == FW_EMPTY) { fresh[u] = true; pend[u] = false; }
            else if (o[u] == key[u]) pend[u] = false;
            else { a8[u] += 8; more = true; }
          }
        }
        if (!__any(more)) break;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (pend[u]) { gave_up = true; ok[u] = false; }  // FW_PROBES slots on and still another key's: the next kernel's bucket
        if (fresh[u]) {                                  // a new entry: its bit in the bitmap, its slot in the list
          const unsigned p = a8[u] >> 3;
          atomicOr(&BM[p >> 5], 1u << (p & 31u));
          const unsigned at = atomicAdd(&sh[0], 1u);
          if (at < (unsigned)FW_LIST) L[at] = (unsigned short)p;
        }
      }
    }
    FW_MARK(2)
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (!(FW_ABL & 4) && ok[u]) atomicAdd(reinterpret_cast<unsigned*>(Cb + (a8[u] >> 1)), 1u);
  };

  // ---- a request's keys arrive: insert them; after a bucket's last request the table is ranked, emitted and cleared by
  // fw_finalize — ONE copy of that code (a real call): sixteen wavefronts per CU run this kernel each at a place of its
  // own, and what they execute has to stay in the instruction cache
  auto consume = [&](const fw_item& it, const fw_v2 (&r)[2]) {
    const int nb = it.nb;
    if (nb >= 2 && !__any(gave_up)) {                      // (uniform; a bucket the table could not hold is not probed further)
      const unsigned long long key[4] = {r[0].x, r[0].y, r[1].x, r[1].y};
      const int i0 = it.c + 2 * lane, i1 = i0 + 128;
      const bool act[4] = {i0 + 1 < nb, i0 < nb, i1 + 1 < nb, i1 < nb};
      insert4(key, act);
      // (the probe notes how many keys of the bucket it took to overflow the table: ~700 when they are all distinct, the
      // more the more they repeat — what tells the caller which kernel such buckets belong to)
      if (PROBE && __any(gave_up) && lane == 0)
        atomicAdd(&header[FS_PROBE_GIVEUP], (unsigned long long)min(it.c + FW_ITEM, nb));
    }
    FW_MARK(3)
    if (it.c + FW_ITEM < nb) return;                     // (uniform) more requests of this bucket follow
    fw_finalize<PROBE>(A, it.b, it.lo, it.src, it.size, nb, __any(gave_up), header, Dv, todo_ids, loose_counts, big_table, n_big, big_keys,
                       big_counts);
    gave_up = false;
    FW_MARK(4)
  };

  fw_item desc[FW_RING];
  fw_v2 ring[FW_RING][2];
#pragma unroll
  for (int r = 0; r < FW_RING; ++r) {
    desc[r] = next_item();
    issue(desc[r], ring[r]);
  }
  for (bool more = true; more;) {
#pragma unroll
    for (int r = 0; r < FW_RING; ++r) {
      if (desc[r].b >= n_buckets) { more = false; break; }
      consume(desc[r], ring[r]);
      desc[r] = next_item();
      issue(desc[r], ring[r]);
      FW_MARK(5)
    }
  }
#ifdef FW_PHASES
  if (lane == 0 && !PROBE) for (int i = 0; i < 8; ++i) atomicAdd(header + FS_LOG + 8 + i, ph_t[i]);      // (experiment builds only)
#endif
}

}  // namespace

// probe == false: every bucket the table of one wavefront can hold is finished the loose way (see finish_dup.hip) and gets
// its Dv[b]; the others are listed in todo_ids / header[FS_TODO], their keys untouched.
// probe == true: ~`probe_buckets` evenly spaced buckets go through the table and nothing is written but header[FS_PROBE_BAD]
// (sampled buckets the table could not hold), [FS_PROBE_DISTINCT] (distinct keys of the others), [FS_PROBE_KEYS].
int bnpk_finish_wave_launch(bnpk_ctx* ctx, bool probe, int64_t probe_buckets, uint64_t* part, int64_t n, const int64_t* bucket_off,
                            int64_t n_buckets, int low_bits, unsigned long long* header, int64_t* Dv, unsigned* todo_ids,
                            int64_t* loose_counts, const int64_t* big_table, int n_big, const uint64_t* big_keys,
                            const int64_t* big_counts, int64_t pstride, hipStream_t s) {
  if (n < 2) return BNPK_ERR_ARG;                        // (the loads are pairs of keys)
  if (!ctx->finish_wave_ready) {
    int per_cu = 1 << 30;
    const void* kernels[4] = {(const void*)finish_wave_kernel<false, false>, (const void*)finish_wave_kernel<true, false>,
                              (const void*)finish_wave_kernel<false, true>, (const void*)finish_wave_kernel<true, true>};
    for (const void* f : kernels) {
      BNPK_HIP(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FW_LDS));
      int c = 0;
      BNPK_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&c, f, 64, FW_LDS));
      per_cu = std::min(per_cu, c);
    }
    // (four per SIMD; a seventeenth wavefront would start when another has finished, and double the run time)
    ctx->finish_wave_grid = ctx->compute_units * std::max(1, std::min(per_cu, 16));
    if (const char* e = getenv("BNPK_WAVE_PER_CU")) {      // (experiments)
      fprintf(stderr, "finish_wave: %d wavefronts per CU by the occupancy query, %s requested\n", per_cu, e);
      ctx->finish_wave_grid = ctx->compute_units * std::max(1, atoi(e));
    }
    ctx->finish_wave_ready = true;
  }
  const int sbits = std::min(low_bits, FW_HBITS), sshift = low_bits - sbits;
  const int64_t stride = probe ? std::max<int64_t>(1, n_buckets / std::max<int64_t>(probe_buckets, 1)) : 1;
  const unsigned grid = (unsigned)std::min<int64_t>(ceil_div(n_buckets, stride), (int64_t)ctx->finish_wave_grid);
#define FW_LAUNCH(HI, PROBE)                                                                                                  \
  hipLaunchKernelGGL((finish_wave_kernel<HI, PROBE>), dim3(grid), dim3(64), FW_LDS, s, part, n, bucket_off, n_buckets, stride, \
                     sshift, sbits, header, Dv, todo_ids, loose_counts, big_table, n_big, big_keys, big_counts, pstride)
  if (sshift >= 32) { if (probe) FW_LAUNCH(true, true); else FW_LAUNCH(true, false); }
  else { if (probe) FW_LAUNCH(false, true); else FW_LAUNCH(false, false); }
#undef FW_LAUNCH
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}
