// Duplicate-aware finishing kernel of the sparse k-mer histogram (A9 for k > 13): the path of buckets whose keys repeat
// — the k-mers of reads that cover a genome many times over, SURVEY §8(d)'s S-genome: 6e9 keys, 1e8 distinct.
//
// The fast kernel (finish.hip) sorts a bucket's keys and finds every key distinct; the general kernel sorts them and
// pays for every copy of a key as for a key of its own (a triangular pass over the copies inside a bin).  Here a bucket
// is never sorted.  Its keys are INSERTED, straight from the registers they were loaded into, into an open-addressing
// table in LDS whose home slot is a monotone function of the key (the bucket's leading free bits, scaled to the table):
// one 64-bit compare-and-swap claims an empty slot or meets the key's first copy, one 16-bit add counts the copy.  The
// cost per key does not depend on its multiplicity, and what is left to sort are the table's entries: with linear
// probing and monotone homes an entry's home lies inside its cluster (maximal run of used slots), so the clusters
// ascend, and inside a cluster — a handful of slots while the table is sparse — every entry ranks itself against its
// neighbours: place = entries before its slot - larger entries on its left + smaller entries on its right.  Slot
// owners again (lane l of a wavefront reads slot 64 c + l: conflict-free), no atomics, no bin boundaries.
//
// Output positions: a bucket does not know how many distinct keys the buckets before it hold, and nobody waits for
// anybody (no look-back, no spinning): the distinct keys go back over the bucket's own keys (all of them are in
// registers by then; D <= n), the counts to the same positions of the future key array, D[b] to the state, and after
// one scan over D two copies move the runs to their final places (bnpk_finish_compact_launch).  For duplicate-heavy
// keys the runs are a few per cent of the input, so the second pass costs next to nothing.
//
// A bucket the table cannot hold (more distinct keys than slots, a probe sequence over FD_PROBES — skewed low bits) is
// handed to the general kernel through the redo list, which finishes it the same loose way.
#include <algorithm>

#include "finish.h"
#include "scan.h"

namespace {

constexpr int FD_THREADS = 512;
constexpr int FD_WAVES = FD_THREADS / 64;
constexpr int FD_ITEMS = FINISH_CAP / FD_THREADS;        // keys of a bucket per lane (16)
#ifndef FD_GROUP_N
#define FD_GROUP_N 4
#endif
constexpr int FD_GROUP = FD_GROUP_N;                     // ... inserted together (independent LDS round trips)
constexpr int FD_CHUNKS = 12;                            // chunks of 64 slots owned by one wavefront (dense tables)
constexpr int FD_WG = 4;                                 // ... read together
constexpr int FD_REGION = 64 * FD_CHUNKS;
constexpr int FD_TS = FD_WAVES * FD_REGION;              // table slots (6144)
constexpr int FD_PROBES = 48;                            // longest probe sequence; beyond it the bucket is handed back
constexpr int FD_HOME = FD_TS - FD_PROBES - 16;          // homes: [0, FD_HOME) — the last slot of the table stays empty
constexpr int FD_GUARD = 8;                              // empty slots in front of slot 0 (the left walks end there)
constexpr int FD_HBITS = 16;                             // leading free key bits that make up the home slot
constexpr int FD_BM_WORDS = FD_TS / 32;                  // bitmap of the used slots (192 words)
constexpr int FD_BM_LANES = FD_BM_WORDS / 4;             // ... four words per lane of a wavefront (48 lanes)
constexpr int FD_LIST = 3 * FD_THREADS;                  // used slots listed in the order they were claimed (sparse tables)
constexpr unsigned long long FD_EMPTY = ~0ull;           // (keys are < 2^63)
static_assert(FD_ITEMS % FD_GROUP == 0 && FD_CHUNKS % FD_WG == 0, "whole groups");
static_assert(FD_HOME < (1 << 24) && FD_HBITS <= 16, "the home slot is a 24-bit product");
static_assert(FD_BM_WORDS % 4 == 0 && FD_BM_LANES <= 64, "the bitmap prefix is one wavefront scan");
constexpr size_t FD_OFF_T = (size_t)FD_GUARD * 8;
constexpr size_t FD_OFF_C = FD_OFF_T + (size_t)FD_TS * 8;               // 32-bit counts, one per slot
constexpr size_t FD_OFF_BM = FD_OFF_C + (size_t)FD_TS * 4;
constexpr size_t FD_OFF_LIST = FD_OFF_BM + (size_t)FD_BM_WORDS * 4;
constexpr size_t FD_OFF_SH = FD_OFF_LIST + (size_t)FD_LIST * 2;         // [0] used slots, [1] give-up flag
constexpr size_t FD_LDS = FD_OFF_SH + 64;
static_assert(2 * FD_LDS <= 160 * 1024, "two workgroups per CU");

__device__ __forceinline__ int64_t fd_uniform(int64_t v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uint64_t)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}
template <int N> struct fd_int { static constexpr int value = N; };
__device__ __forceinline__ unsigned long long fd_lane(unsigned long long v, int l) {   // lane l's value, in scalar registers
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
  return ((unsigned long long)hi << 32) | lo;
}
// a wave shift by one lane (CTRL: 0x138 = from the lane below, 0x130 = from the lane above); the lane without a source
// gets `bound`
template <typename CTRL>
__device__ __forceinline__ unsigned long long fd_shift(unsigned long long v, unsigned long long bound, CTRL) {
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)bound, (int)(unsigned)v, CTRL::value, 0xf, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(bound >> 32), (int)(unsigned)(v >> 32), CTRL::value, 0xf, 0xf, false);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ int fd_fresh(int x) {         // opaque to the optimiser: recomputed where used, not kept live
  asm volatile("" : "+v"(x));
  return x;
}

// HI: the home bits lie in the key's high word (sshift >= 32); otherwise they straddle the words (one V_ALIGNBIT)
template <bool HI>
__global__ __launch_bounds__(FD_THREADS, 4) void finish_dup_kernel(
    uint64_t* A, const int64_t* __restrict__ bucket_off, int64_t n_buckets, int sshift, int sbits,
    unsigned long long* __restrict__ header, int64_t* __restrict__ Dv, unsigned* __restrict__ redo_ids,
    int64_t* __restrict__ loose_counts, const int64_t* __restrict__ big_table, int n_big,
    const uint64_t* __restrict__ big_keys, const int64_t* __restrict__ big_counts, const unsigned* __restrict__ todo_ids,
    int64_t pstride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // the buckets to finish: all of them, or (todo_ids) the ones the wavefront kernel listed — the length of its list is
  // read here, on the device
  const int64_t n_items = todo_ids ? fd_uniform((int64_t)header[FS_TODO]) : n_buckets;
  unsigned long long* T = reinterpret_cast<unsigned long long*>(smem + FD_OFF_T);
  unsigned* C = reinterpret_cast<unsigned*>(smem + FD_OFF_C);
  unsigned* BM = reinterpret_cast<unsigned*>(smem + FD_OFF_BM);
  unsigned short* L = reinterpret_cast<unsigned short*>(smem + FD_OFF_LIST);
  unsigned* sh = reinterpret_cast<unsigned*>(smem + FD_OFF_SH);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned hmask = (1u << sbits) - 1u;
  const unsigned hshift = (unsigned)(HI ? sshift - 32 : sshift);
  const int64_t G = gridDim.x;

  for (int i = tid - FD_GUARD; i < FD_TS; i += FD_THREADS) T[i] = FD_EMPTY;
  for (int i = tid; i < FD_TS; i += FD_THREADS) C[i] = 0;
  if (tid < FD_BM_WORDS) BM[tid] = 0;
  if (tid < 16) sh[tid] = 0;
  __syncthreads();

  struct bucket_t { int64_t id, lo, size; int nb; int64_t src; };   // src: where the bucket's keys lie in A (lo, or id * pstride)
  auto fetch_offsets = [&](int64_t j, int64_t& o0, int64_t& o1, int64_t& id) {   // (scalar loads; consumed an iteration later)
    o0 = 0; o1 = 0; id = 0;
    if (j < n_items) {
      id = todo_ids ? (int64_t)todo_ids[j] : j;
      o0 = bucket_off[id];
      o1 = bucket_off[id + 1];
    }
  };
  auto open_bucket = [&](int64_t o0, int64_t o1, int64_t id) {
    bucket_t x;
    x.id = fd_uniform(id);
    x.lo = fd_uniform(o0);
    x.size = fd_uniform(o1) - x.lo;
    x.nb = x.size > FINISH_CAP ? 0 : (int)x.size;
    x.src = pstride ? x.id * pstride : x.lo;
    return x;
  };
  uint64_t k[FD_ITEMS];
  auto load_keys = [&](const bucket_t& x) {              // k[q] = key tid + 512 q of the bucket (clamped: branch-free)
    const uint64_t* Ab = A + x.src;
    const int t = fd_fresh(tid);
#pragma unroll
    for (int q0 = 0; q0 < FD_ITEMS; q0 += FD_GROUP) {
      if (q0 * FD_THREADS < x.nb) {                      // uniform
#pragma unroll
        for (int u = 0; u < FD_GROUP; ++u)
          k[q0 + u] = __builtin_nontemporal_load(&Ab[(unsigned)min(t + (q0 + u) * FD_THREADS, x.nb - 1)]);
      }
    }
  };
  // byte offset of a key's home slot: its leading free bits, scaled to [0, FD_HOME)
  auto home8 = [&](uint64_t key) -> unsigned {
    const unsigned lo = (unsigned)key, hi = (unsigned)(key >> 32);
    const unsigned v = (HI ? hi >> hshift : __builtin_amdgcn_alignbit(hi, lo, hshift)) & hmask;
    return (__umul24(v, (unsigned)FD_HOME) >> sbits) << 3;
  };
  // One group of keys goes into the table: all compare-and-swaps first, then (the LDS answers in order: nothing that
  // waits may sit between them) the adds.  FULL: every lane holds a key of the bucket in every item of the group.
  bool gave_up = false;
  auto insert_group = [&](auto q0_tag, auto full_tag, int nb) {
    constexpr int q0 = decltype(q0_tag)::value;
    constexpr bool FULL = decltype(full_tag)::value != 0;
    const int t0 = fd_fresh(tid);
    unsigned a8[FD_GROUP];
    unsigned long long old[FD_GROUP];
    bool act[FD_GROUP];
#pragma unroll
    for (int u = 0; u < FD_GROUP; ++u) {
      a8[u] = home8(k[q0 + u]);
      act[u] = FULL || t0 + (q0 + u) * FD_THREADS < nb;
      old[u] = k[q0 + u];
    }
    unsigned char* Tb = reinterpret_cast<unsigned char*>(T);
#pragma unroll
    for (int u = 0; u < FD_GROUP; ++u)
      if (act[u]) old[u] = atomicCAS(reinterpret_cast<unsigned long long*>(Tb + a8[u]), FD_EMPTY, (unsigned long long)k[q0 + u]);
#pragma unroll
    for (int u = 0; u < FD_GROUP; ++u) {
      bool fresh = old[u] == FD_EMPTY;                   // (lanes without a key: old == key)
      if (!fresh && old[u] != k[q0 + u]) {               // another key lives there: probe on (rare while the table is sparse)
        const unsigned long long key = k[q0 + u];
        unsigned long long o = old[u];
        unsigned p8 = a8[u];
        for (int probes = 0; o != FD_EMPTY && o != key && probes < FD_PROBES; ++probes) {
          p8 += 8;
          o = atomicCAS(reinterpret_cast<unsigned long long*>(Tb + p8), FD_EMPTY, key);
        }
        a8[u] = p8;
        fresh = o == FD_EMPTY;
        if (!fresh && o != key) { gave_up = true; act[u] = false; }
      }
      if (fresh) {                                       // a new entry: its bit in the bitmap, its slot in the list
        const unsigned p = a8[u] >> 3;
        atomicOr(&BM[p >> 5], 1u << (p & 31u));
        const unsigned at = atomicAdd(&sh[0], 1u);
        if (at < (unsigned)FD_LIST) L[at] = (unsigned short)p;
      }
    }
    unsigned char* Cb = reinterpret_cast<unsigned char*>(C);
#pragma unroll
    for (int u = 0; u < FD_GROUP; ++u)
      if (act[u]) atomicAdd(reinterpret_cast<unsigned*>(Cb + (a8[u] >> 1)), 1u);
  };
#pragma unroll
  for (int q = 0; q < FD_ITEMS; ++q) k[q] = 0;
  int64_t f0, f1, fi;
  fetch_offsets((int64_t)blockIdx.x, f0, f1, fi);
  bucket_t cur = open_bucket(f0, f1, fi);
  load_keys(cur);
  fetch_offsets((int64_t)blockIdx.x + G, f0, f1, fi);

#ifdef FD_PHASES
  unsigned long long ph_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_last = __builtin_readcyclecounter();
#define FD_MARK(i) { const unsigned long long now__ = __builtin_readcyclecounter(); ph_t[i] += now__ - ph_last; ph_last = now__; }
#else
#define FD_MARK(i)
#endif
  for (int64_t j = blockIdx.x; j < n_items; j += G) {
    const bucket_t nxt = open_bucket(f0, f1, fi);        // item j + G (its offsets were fetched an iteration ago)
    fetch_offsets(j + 2 * G, f0, f1, fi);
    const int64_t b = cur.id;
    const int nb = cur.nb;
    if (nb == 0) {                                       // empty, or a heavy-hitter bucket counted by the caller beforehand
      unsigned D = 0;
      if (cur.size > 0) {
        int lo_i = 0, hi_i = n_big;
        while (lo_i < hi_i) {
          const int mid = (lo_i + hi_i) >> 1;
          if (big_table[3 * mid] < b) lo_i = mid + 1; else hi_i = mid;
        }
        if (lo_i < n_big && big_table[3 * lo_i] == b) {
          D = (unsigned)fd_uniform(big_table[3 * lo_i + 1]);
          const int64_t src = fd_uniform(big_table[3 * lo_i + 2]);
          uint64_t* ko = A + cur.src;
          int64_t* co = loose_counts + cur.lo;
          for (unsigned i = (unsigned)fd_fresh(tid); i < D; i += FD_THREADS) {
            ko[i] = big_keys[src + i];
            co[i] = big_counts[src + i];
          }
        } else if (tid == 0) {
          atomicOr(&header[FS_FLAGS], 1ull);
        }
      }
      if (tid == 0) Dv[b] = D;
      load_keys(nxt);
    } else {
      // ---- insert: every key claims the first free slot at or after its home, or meets its first copy there
      gave_up = false;
      {
        auto step = [&](auto q0_tag) {
          constexpr int q0 = decltype(q0_tag)::value;
          if (q0 * FD_THREADS < nb) {                    // uniform
            if ((q0 + FD_GROUP) * FD_THREADS <= nb) insert_group(q0_tag, fd_int<1>(), nb);
            else insert_group(q0_tag, fd_int<0>(), nb);
          }
        };
        step(fd_int<0>());
        if (FD_GROUP < FD_ITEMS) step(fd_int<FD_GROUP % FD_ITEMS>());
        if (2 * FD_GROUP < FD_ITEMS) step(fd_int<2 * FD_GROUP % FD_ITEMS>());
        if (3 * FD_GROUP < FD_ITEMS) step(fd_int<3 * FD_GROUP % FD_ITEMS>());
        static_assert(4 * FD_GROUP >= FD_ITEMS, "four steps cover the items");
      }
      if (gave_up) sh[1] = 1u;
      FD_MARK(0)
      __syncthreads();                                   // (1) the table holds the bucket
      FD_MARK(1)
      load_keys(nxt);                                    // k[] is free: the next bucket's keys, in flight until the next iteration
      const unsigned D = (unsigned)__builtin_amdgcn_readfirstlane((int)sh[0]);     // used slots = distinct keys
      const bool bad = __builtin_amdgcn_readfirstlane((int)sh[1]) != 0;
      uint64_t* ko = A + cur.src;                         // scalar bases, 32-bit lane offsets
      int64_t* co = loose_counts + cur.lo;
      const int l3 = fd_fresh(lane);
      // used slots before a slot: every wavefront scans the bitmap's popcounts in its own registers — lane l holds the
      // words 4 l .. 4 l + 3, the used slots before them and (one byte each) before the second, third and fourth
      unsigned bm_before = 0, bm_within = 0;
      if (!bad) {
        uint4 w4 = make_uint4(0, 0, 0, 0);
        if (l3 < FD_BM_LANES) w4 = reinterpret_cast<const uint4*>(BM)[l3];
        const unsigned p0 = __popc(w4.x), p1 = p0 + __popc(w4.y), p2 = p1 + __popc(w4.z), p3 = p2 + __popc(w4.w);
        bm_before = wave_inclusive_scan(p3) - p3;
        bm_within = (p0 << 8) | (p1 << 16) | (p2 << 24);
      }
      auto rank_of = [&](unsigned s) -> unsigned {       // used slots before slot s (all lanes must call)
        const unsigned w = s >> 5;
        const unsigned e = (unsigned)__shfl((int)bm_before, (int)(w >> 2), 64), in4 = (unsigned)__shfl((int)bm_within, (int)(w >> 2), 64);
        return e + ((in4 >> ((w & 3u) * 8u)) & 0xffu) + (unsigned)__popc(BM[w] & ((1u << (s & 31u)) - 1u));
      };
      // an entry's place: used slots before it, minus the larger entries on its left in its cluster (they come after
      // it), plus the smaller ones on its right
      auto walk = [&](bool used, int s, unsigned long long x, unsigned long long y, unsigned long long z, unsigned idx) -> unsigned {
        bool go_l = used && y != FD_EMPTY, go_r = used && z != FD_EMPTY;
        unsigned down = (go_l && y > x) ? 1u : 0u, up = (go_r && z < x) ? 1u : 0u;
        for (int d = 2; __any(go_l || go_r); ++d) {
          unsigned long long yy = FD_EMPTY, zz = FD_EMPTY;
          if (go_l) yy = T[s - d];
          if (go_r) zz = T[s + d];
          go_l = go_l && yy != FD_EMPTY;
          go_r = go_r && zz != FD_EMPTY;
          down += (go_l && yy > x) ? 1u : 0u;
          up += (go_r && zz < x) ? 1u : 0u;
        }
        return idx - down + up;
      };
      unsigned dirty = 0;                                // dense tables: chunks in which this lane's slot is used
      int mine[FD_LIST / FD_THREADS];                    // sparse tables: the slots this lane emitted (-1: none)
#pragma unroll
      for (int r = 0; r < FD_LIST / FD_THREADS; ++r) mine[r] = -1;
      const bool sparse = D <= (unsigned)FD_LIST;        // (uniform)
      if (bad) {
        dirty = (1u << FD_CHUNKS) - 1u;
      } else if (sparse) {
        // ---- sparse table: one lane per listed slot
#pragma unroll
        for (int r = 0; r < FD_LIST / FD_THREADS; ++r) {
          const unsigned j0 = (unsigned)(r * FD_THREADS + wave * 64);
          if (j0 < D) {                                  // uniform
            const bool used = j0 + (unsigned)l3 < D;
            const int s = used ? (int)L[j0 + (unsigned)l3] : 0;
            const unsigned long long x = T[s], y = T[s - 1], z = T[s + 1];
            const unsigned cnt = C[s];
            const unsigned place = walk(used, s, x, y, z, rank_of((unsigned)s));
            if (used) {
              ko[place] = x;
              co[place] = (int64_t)cnt;
              mine[r] = s;
            }
          }
        }
      } else {
        // ---- dense table: slot owners — lane l of a wavefront reads slot 64 c + l of its region
        unsigned running = (unsigned)__builtin_amdgcn_readlane((int)bm_before, wave * (FD_REGION / 128));
        const uint64_t lt = (1ull << l3) - 1ull;
#pragma unroll
        for (int c0 = 0; c0 < FD_CHUNKS; c0 += FD_WG) {
          const int s0 = wave * FD_REGION + c0 * 64 + l3;
          // a slot and its count: all reads of the group in flight together; the two neighbours of a slot are the
          // neighbouring lanes' slots (DPP wave shifts), at the ends of a chunk the next chunk's — or, at the ends of
          // the group, one more LDS word (the same for all lanes)
          unsigned long long x[FD_WG];
          unsigned cnt[FD_WG];
#pragma unroll
          for (int u = 0; u < FD_WG; ++u) {
            x[u] = T[s0 + 64 * u];
            cnt[u] = C[s0 + 64 * u];
          }
          const int g0 = wave * FD_REGION + c0 * 64;
          const unsigned long long before_group = T[g0 - 1], after_group = T[g0 + 64 * FD_WG];
#pragma unroll
          for (int u = 0; u < FD_WG; ++u) {
            const bool used = x[u] != FD_EMPTY;
            const uint64_t m = __ballot(used);
            if (m != 0) {                                // uniform
              const unsigned long long lb = u == 0 ? before_group : fd_lane(x[u > 0 ? u - 1 : 0], 63);
              const unsigned long long rb = u == FD_WG - 1 ? after_group : fd_lane(x[u < FD_WG - 1 ? u + 1 : u], 0);
              const unsigned long long y = fd_shift(x[u], lb, fd_int<0x138>());       // lane l: the slot of lane l - 1
              const unsigned long long z = fd_shift(x[u], rb, fd_int<0x130>());       // lane l: the slot of lane l + 1
              const unsigned idx = running + (unsigned)__popcll(m & lt);
              running += (unsigned)__popcll(m);
              const unsigned place = walk(used, s0 + 64 * u, x[u], y, z, idx);
              if (used) {
                ko[place] = x[u];
                co[place] = (int64_t)cnt[u];
                dirty |= 1u << (c0 + u);
              }
            }
          }
        }
      }
      if (tid == 0) {
        if (bad) {
          const unsigned long long at = atomicAdd(&header[FS_REDO], 1ull);
          redo_ids[at] = (unsigned)b;
          Dv[b] = 0;
        } else {
          Dv[b] = D;
        }
      }
      FD_MARK(2)
      __syncthreads();                                   // (2) every neighbour has been read
      FD_MARK(3)
      if (!bad && sparse) {                              // (uniform)
#pragma unroll
        for (int r = 0; r < FD_LIST / FD_THREADS; ++r) {
          if (mine[r] >= 0) {
            T[mine[r]] = FD_EMPTY;
            C[mine[r]] = 0;
            BM[mine[r] >> 5] = 0;
          }
        }
      } else {
        const int l4 = fd_fresh(lane);
#pragma unroll
        for (int c = 0; c < FD_CHUNKS; ++c) {
          if ((dirty >> c) & 1u) {
            const int s = wave * FD_REGION + c * 64 + l4;
            T[s] = FD_EMPTY;
            C[s] = 0;
          }
        }
        if (tid < FD_BM_WORDS) BM[tid] = 0;
      }
      if (tid < 2) sh[tid] = 0;
      __syncthreads();                                   // (3) the table is empty again
      FD_MARK(4)
    }
    cur = nxt;
  }
#ifdef FD_PHASES
  if (tid == 64) for (int i = 0; i < 8; ++i) atomicAdd(header + FS_LOG + i, ph_t[i]);      // (experiment builds only)
#endif
}

// dst[T[b] + i] = src[bucket_off[b] + i]: one wavefront per bucket
__global__ __launch_bounds__(256) void finish_compact_kernel(const int64_t* __restrict__ src, int64_t* __restrict__ dst,
                                                             const int64_t* __restrict__ bucket_off,
                                                             const int64_t* __restrict__ T, int64_t n_buckets,
                                                             unsigned long long* __restrict__ header, int64_t pstride) {
  const int lane = threadIdx.x & 63;
  const int64_t n_waves = (int64_t)gridDim.x * (256 / 64);
  if (blockIdx.x == 0 && threadIdx.x == 0) header[FS_UNIQUE] = (unsigned long long)T[n_buckets];
  for (int64_t b = (int64_t)blockIdx.x * (256 / 64) + (threadIdx.x >> 6); b < n_buckets; b += n_waves) {
    const int64_t t0 = fd_uniform(T[b]), len = fd_uniform(T[b + 1]) - t0, lo = pstride ? b * pstride : fd_uniform(bucket_off[b]);
    const int64_t* sp = src + lo;
    int64_t* dp = dst + t0;
    for (int64_t i = lane; i < len; i += 64) dp[i] = sp[i];
  }
}

}  // namespace

int bnpk_finish_dup_launch(bnpk_ctx* ctx, uint64_t* part, const int64_t* bucket_off, int64_t n_buckets, int low_bits,
                           unsigned long long* header, int64_t* Dv, unsigned* redo_ids, int64_t* loose_counts,
                           const int64_t* big_table, int n_big, const uint64_t* big_keys, const int64_t* big_counts,
                           const unsigned* todo_ids, int64_t pstride, hipStream_t s) {
  if (!ctx->finish_dup_ready) {
    BNPK_HIP(ctx, hipFuncSetAttribute((const void*)finish_dup_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FD_LDS));
    BNPK_HIP(ctx, hipFuncSetAttribute((const void*)finish_dup_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FD_LDS));
    int per_cu = 0, per_cu_hi = 0;
    BNPK_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)finish_dup_kernel<false>, FD_THREADS, FD_LDS));
    BNPK_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_hi, (const void*)finish_dup_kernel<true>, FD_THREADS, FD_LDS));
    ctx->finish_dup_grid = ctx->compute_units * std::max(1, std::min(per_cu, per_cu_hi));
    ctx->finish_dup_ready = true;
  }
  const int sbits = std::min(low_bits, FD_HBITS), sshift = low_bits - sbits;
  const unsigned grid = (unsigned)std::min<int64_t>(n_buckets, (int64_t)ctx->finish_dup_grid);
  if (sshift >= 32)
    hipLaunchKernelGGL(finish_dup_kernel<true>, dim3(grid), dim3(FD_THREADS), FD_LDS, s, part, bucket_off, n_buckets, sshift, sbits,
                       header, Dv, redo_ids, loose_counts, big_table, n_big, big_keys, big_counts, todo_ids, pstride);
  else
    hipLaunchKernelGGL(finish_dup_kernel<false>, dim3(grid), dim3(FD_THREADS), FD_LDS, s, part, bucket_off, n_buckets, sshift, sbits,
                       header, Dv, redo_ids, loose_counts, big_table, n_big, big_keys, big_counts, todo_ids, pstride);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

int bnpk_finish_compact_launch(bnpk_ctx* ctx, const int64_t* src, int64_t* dst, const int64_t* bucket_off, const int64_t* T,
                               int64_t n_buckets, unsigned long long* header, int64_t pstride, hipStream_t s) {
  const unsigned grid = grid_for(std::min<int64_t>(ceil_div(n_buckets, 4), (int64_t)ctx->compute_units * 16));
  hipLaunchKernelGGL(finish_compact_kernel, dim3(grid), dim3(256), 0, s, src, dst, bucket_off, T, n_buckets, header, pstride);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}
