// Finishing kernels of the sparse k-mer histogram (A9 for k > 13): every bucket of the MSD-partitioned keys (radix.hip)
// — equal top bits, at most bnpk_finish_capacity() keys, any order inside — is sorted in LDS, its duplicates are
// counted and the distinct (key, count) pairs are written at their final sorted position in one pass over HBM.
// Two kernels share the work: finish_fast_kernel (duplicate-free buckets: the common case for k = 31) and
// finish_sorted_kernel (any multiplicities; also works off the fast kernel's redo list).
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "finish.h"
#include "scan.h"

namespace {

// ===================================================================================================================
// Finishing kernel: every bucket of the partitioned keys (equal top bits, <= FN_CAP keys, arbitrary order inside)
// is sorted in LDS, its duplicates are counted and the distinct (key, count) pairs are written in sorted order.
// Single pass: buckets are handed out in ticket order; the output offset of a bucket (= number of distinct keys
// in all earlier buckets) comes from a decoupled look-back over one 64-bit {flag, value} word per bucket, walked
// by wavefront 0 (64 predecessors per poll) while the other wavefronts rank the bucket's keys, so its latency
// is hidden.  The next bucket's keys are loaded while the current one is processed.
constexpr int FN_THREADS = 1024;
constexpr int FN_CAP = FINISH_CAP;
constexpr int FN_ITEMS = FN_CAP / FN_THREADS;
constexpr int FN_MAXBITS = 12;
constexpr int FN_MAXBINS = 1 << FN_MAXBITS;
constexpr int FN_WORDS = FN_CAP / 64;                // first-occurrence mask words
constexpr int FN_WPL = FN_WORDS / 64;                // ... per lane of a wavefront
constexpr int FN_BINS_PER_LANE = FN_MAXBINS / FN_THREADS;
static_assert(FN_WPL == 1 || FN_WPL == 2, "the mask-prefix code below keeps one or two mask words per lane");
// d_state words (finish.h): [0] error flags (1 = bucket over capacity, 2 = look-back gave up), [1] ticket counter,
// [2] number of distinct keys, [8 + b] status word of bucket b
constexpr int FS_BUCKETS = 8;
constexpr unsigned long long FN_AGG = 1ull << 62, FN_INC = 2ull << 62, FN_VALUE = (1ull << 62) - 1;
constexpr unsigned FN_SPIN_LIMIT = 1u << 22;
#ifndef FN_SLEEP
#define FN_SLEEP 1
#endif

#define FN_SLOT(w) ((w) & 0x1fffu)
#define FN_RANK(w) (((w) >> 13) & 0x1fffu)
#define FN_LESS(w) ((w) >> 26)

constexpr size_t FN_BINS_BYTES = (size_t)(FN_MAXBINS + 4) * 4;
constexpr size_t FN_OFF_BINS = (size_t)FN_CAP * 8;                      // two bin arrays (ping-pong between buckets)
constexpr size_t FN_OFF_AUX = FN_OFF_BINS + 2 * FN_BINS_BYTES;          // per slot {rank increments : 16 | duplicates seen : 16}
constexpr size_t FN_OFF_MASK = FN_OFF_AUX + (size_t)FN_CAP * 4;
constexpr size_t FN_OFF_LIST = FN_OFF_MASK + (size_t)FN_WORDS * 8;       // per wavefront: slots of the keys with long walks
constexpr size_t FN_OFF_WSUM = FN_OFF_LIST + (size_t)FN_CAP * 2;
constexpr size_t FN_OFF_SH = FN_OFF_WSUM + 32 * 4;
constexpr size_t FN_LDS = FN_OFF_SH + 8 * 8;

// a value every lane holds identically -> scalar registers (the compiler cannot prove that what was read from LDS /
// global memory is wave-uniform and would keep it in vector registers)
__device__ __forceinline__ int64_t fn_uniform(int64_t v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uint64_t)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

// the same value, but opaque to the optimiser: what is derived from it is recomputed where it is used (two or three
// VALU instructions) instead of being hoisted out of the bucket loop and held in — or spilled from — registers
__device__ __forceinline__ int fn_fresh(int x) {
  asm volatile("" : "+v"(x));
  return x;
}

struct fn_bucket {
  int64_t b, lo, t;   // t: the ticket the bucket was handed out under
  int64_t src;        // where the bucket's keys lie in the partitioned array: lo, or b * pstride for buckets of fixed stride
  int nb;          // keys in the bucket; 0 = nothing to sort (empty, past the end, or over capacity)
  bool over;
};

// bucket b from its two offsets (loaded one iteration earlier, so nothing waits on them here)
__device__ __forceinline__ fn_bucket fn_open(int64_t n_buckets, int64_t b, int64_t lo, int64_t hi, int64_t t, int64_t pstride) {
  fn_bucket x;
  x.b = b;
  x.t = t;
  x.lo = 0;
  x.src = 0;
  x.nb = 0;
  x.over = false;
  if (b < n_buckets) {
    x.lo = lo;
    x.src = pstride ? b * pstride : lo;
    const int64_t m = hi - lo;
    x.over = m > FN_CAP;
    x.nb = x.over ? 0 : (int)m;
  }
  return x;
}

// MODE 0: every bucket, output positions by look-back.  MODE 1 (redo): the buckets of a list, at the positions given with
// it.  MODE 2 (loose): the buckets of a list whose length is read from the state on the device (the duplicate-aware
// kernel's hand-backs); every bucket's distinct keys go back over the bucket's own keys (keys_out == A), the counts to
// the same positions of counts_out, its number of distinct keys to loose_D — finish_dup.hip's convention.
template <int MODE>
__global__ __launch_bounds__(FN_THREADS) void finish_sorted_kernel(const uint64_t* __restrict__ A,
                                                                   const int64_t* __restrict__ bucket_off,
                                                                   int64_t n_buckets, int sshift, int sbits,
                                                                   unsigned long long* __restrict__ state,
                                                                   uint64_t* __restrict__ keys_out,
                                                                   int64_t* __restrict__ counts_out,
                                                                   const int64_t* __restrict__ big_table, int n_big,
                                                                   const uint64_t* __restrict__ big_keys,
                                                                   const int64_t* __restrict__ big_counts,
                                                                   const unsigned* __restrict__ redo_ids,
                                                                   const int64_t* __restrict__ redo_bases,
                                                                   int64_t n_redo_arg, int64_t* __restrict__ loose_D,
                                                                   int64_t pstride) {
  // pstride != 0: bucket b's keys lie at A + b * pstride (radix.hip: buckets of fixed stride, claimed line by line);
  // bucket_off then only says how many they are and where the bucket's output goes
  constexpr bool REDO = MODE != 0, LOOSE = MODE == 2;
  const int64_t n_redo = LOOSE ? fn_uniform((int64_t)state[FS_REDO]) : n_redo_arg;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* stage = reinterpret_cast<uint64_t*>(smem);
  unsigned* bins = reinterpret_cast<unsigned*>(smem + FN_OFF_BINS);            // bins of the bucket being sorted
  unsigned* bins_next = reinterpret_cast<unsigned*>(smem + FN_OFF_BINS + FN_BINS_BYTES);   // ... of the one after it
  unsigned* aux = reinterpret_cast<unsigned*>(smem + FN_OFF_AUX);
  unsigned long long* fmask = reinterpret_cast<unsigned long long*>(smem + FN_OFF_MASK);
  unsigned short* wlist = reinterpret_cast<unsigned short*>(smem + FN_OFF_LIST) + (threadIdx.x >> 6) * (FN_ITEMS * 64);
  unsigned* wsum = reinterpret_cast<unsigned*>(smem + FN_OFF_WSUM);
  long long* sh = reinterpret_cast<long long*>(smem + FN_OFF_SH);       // [0] next ticket, [1] output base, [2] 2nd ticket
  unsigned* sh_dups = reinterpret_cast<unsigned*>(sh + 4);              // duplicate counters, alternating between buckets
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave: scalar
  const unsigned SB = 1u << sbits;

  unsigned* fmask32 = reinterpret_cast<unsigned*>(fmask);

  // ---- decoupled look-back, off the critical path.  A bucket's distinct count is published as soon as it is known;
  // the walk over the predecessors' status words (wavefront 0) starts at the end of the iteration and its first poll
  // (the 64 nearest words) stays in flight, in a register, across the barrier until the start of the next
  // iteration, so the latency of the device-scope loads is hidden.  The result is needed when the NEXT bucket is
  // about to be placed in the stage; until then the sorted keys wait in LDS.  A poll uses every word up to the first
  // one that has not been published yet (nearest predecessor first); 0.6 further, blocking polls per bucket are
  // what a late predecessor costs today.
  // (Measured per 3e9 keys: blocking walk before the final placement 36.0 ms, this 23 ms, no waiting at all 20;
  // every wavefront polling for itself: 52 ms — the status words are a hot spot.)
  auto lb_poll = [&](int64_t top) -> unsigned long long {          // lane l: the status word at distance l behind `top`
    const unsigned long long* first = state + FS_BUCKETS + (top - 63);   // (scalar; only dereferenced where it is valid)
    const bool in_range = top >= 63 || lane <= (int)top;
    return in_range ? __hip_atomic_load(first + (63 - lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : FN_INC;
  };
  unsigned long long lb_v = FN_INC;                    // the poll in flight
  auto lb_resolve = [&](int64_t b) -> long long {      // distinct keys in all buckets before b
    long long base = 0;
    int64_t top = b - 1;
    unsigned long long v = lb_v;
    unsigned spins = 0;
    while (true) {
      const uint64_t incm = __ballot((v & ~FN_VALUE) == FN_INC);
      const uint64_t badm = __ballot((v & ~FN_VALUE) == 0);
      const int first_inc = incm ? __ffsll((long long)incm) - 1 : 64;
      const int first_bad = badm ? __ffsll((long long)badm) - 1 : 64;
      const int use = first_inc < first_bad ? first_inc + 1 : first_bad;    // words usable, nearest first
      long long contrib = lane < use ? (long long)(v & FN_VALUE) : 0ll;
      contrib = wave_reduce_sum(contrib);
      base += fn_uniform(__shfl(contrib, 0, 64));
      if (first_inc < first_bad) break;
      top -= use;
      if (use == 0) {
        if (++spins > FN_SPIN_LIMIT) { if (lane == 0) atomicOr(&state[FS_FLAGS], 2ull); break; }
        __builtin_amdgcn_s_sleep(FN_SLEEP);
      }
      v = lb_poll(top);
    }
    return base;
  };

  // a bucket's own distinct count, published as soon as it is known (nobody waits for the look-back of another bucket)
  auto publish_count = [&](int64_t b, unsigned D) {
    if (!REDO && b > 0 && tid == 0)
      __hip_atomic_store(&state[FS_BUCKETS + b], FN_AGG | (unsigned long long)D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

  for (unsigned i = tid; i <= SB; i += FN_THREADS) { bins[i] = 0; bins_next[i] = 0; }
  if (tid < FN_WORDS) fmask[tid] = 0;
  if (tid < 2) sh_dups[tid] = 0;
  // Software pipeline over tickets: while bucket `cur` is sorted, the keys of the next one are in flight (and get
  // their bin ranks at the end of the iteration) and the offsets of the one after that are being loaded.
  // Tickets are taken ONE PER ITERATION at a fixed phase (the second one only after the first bucket's keys have
  // arrived), so that the i-th buckets of all workgroups form one "round" of consecutive tickets.  Taking two
  // tickets back to back at the start interleaves the rounds: a workgroup's first bucket then waits for its
  // neighbour's second one and the launch degenerates into a staircase (measured: 46 vs 37 ms per 3e9 keys).
  // REDO: the tickets index a list of buckets whose output positions are already known (redo_bases): the buckets the
  // fast kernel found duplicates in.  No look-back, nothing is published.
  auto bucket_of = [&](int64_t t) -> int64_t {
    if (REDO) return t < n_redo ? (int64_t)redo_ids[t] : n_buckets;
    return t;
  };
  if (tid == 0) sh[0] = (long long)atomicAdd(&state[FS_TICKET], 1ull);
  __syncthreads();
  fn_bucket cur;
  {
    const int64_t t0 = fn_uniform(sh[0]);
    const int64_t b0 = fn_uniform(bucket_of(t0));
    cur = fn_open(n_buckets, b0, b0 < n_buckets ? fn_uniform(bucket_off[b0]) : 0, b0 < n_buckets ? fn_uniform(bucket_off[b0 + 1]) : 0, t0, pstride);
  }
  uint64_t k[FN_ITEMS];
  unsigned r[FN_ITEMS];
  unsigned valid = 0;
#pragma unroll
  for (int q = 0; q < FN_ITEMS; ++q) {
    const int i = tid + q * FN_THREADS;
    if (i < cur.nb) {
      k[q] = (A + cur.src)[(unsigned)i];
      r[q] = atomicAdd(&bins[(unsigned)(k[q] >> sshift) & (SB - 1)], 1u);
      valid |= 1u << q;
    }
  }
  __syncthreads();                                    // the ranks are taken
  if (tid == 0) sh[2] = (long long)atomicAdd(&state[FS_TICKET], 1ull);
  __syncthreads();
  int64_t nn_t = fn_uniform(sh[2]);
  int64_t nn_b = fn_uniform(bucket_of(nn_t)), nn_lo = 0, nn_hi = 0;          // the bucket after `cur`: ticket + offsets
  if (nn_b < n_buckets) { nn_lo = bucket_off[nn_b]; nn_hi = bucket_off[nn_b + 1]; }
  unsigned parity = 0;
  // The sorted keys of a bucket stay in LDS until the NEXT bucket is about to be placed there: its look-back runs
  // at the start of the following iteration, so the predecessors have had the rest of an iteration to publish
  // their counts, and the workgroups no longer wait for the slowest one of every round.
  bool have_prev = false, prev_one = true;
  int64_t prev_b = 0, prev_t = 0, prev_big = -1;
  unsigned prev_D = 0;
  auto resolve_prev = [&]() {                          // wavefront 0: where the previous bucket's output goes
    if (LOOSE) {
      if (lane == 0) { sh[1] = bucket_off[prev_b]; sh[3] = pstride ? prev_b * pstride : bucket_off[prev_b]; loose_D[prev_b] = prev_D; }
      return;
    }
    if (REDO) {
      if (lane == 0) sh[1] = redo_bases[prev_t];
      return;
    }
    const long long base = lb_resolve(prev_b);
    if (lane == 0) {
      sh[1] = base;
      __hip_atomic_store(&state[FS_BUCKETS + prev_b], FN_INC | (unsigned long long)(base + prev_D), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      if (prev_b == n_buckets - 1) state[FS_UNIQUE] = (unsigned long long)(base + prev_D);
    }
  };
  auto emit_prev = [&]() {
    const int64_t base = fn_uniform(sh[1]);
    uint64_t* ko = keys_out + (LOOSE ? fn_uniform(sh[3]) : base);   // scalar bases, 32-bit lane offsets (loose: back over the bucket's own keys)
    int64_t* co = counts_out + base;
    const unsigned t0 = (unsigned)fn_fresh(tid);
    if (prev_big >= 0) {                               // pre-counted bucket: copy its (key, count) pairs into place
      const uint64_t* bk = big_keys + prev_big;
      const int64_t* bc = big_counts + prev_big;
      for (unsigned i = t0; i < prev_D; i += FN_THREADS) {
        ko[i] = bk[i];
        co[i] = bc[i];
      }
    } else if (prev_one) {
      for (unsigned i = t0; i < prev_D; i += FN_THREADS) {
        __builtin_nontemporal_store(stage[i], &ko[i]);
        __builtin_nontemporal_store((int64_t)1, &co[i]);
      }
    } else {
      for (unsigned i = t0; i < prev_D; i += FN_THREADS) {
        __builtin_nontemporal_store(stage[i], &ko[i]);
        __builtin_nontemporal_store((int64_t)aux[i], &co[i]);
      }
    }
  };

  while (cur.b < n_buckets) {
    const int nb = cur.nb;
    // A bucket over capacity (heavy-hitter k-mers) has been counted by the caller beforehand: big_table holds
    // {bucket, distinct keys, offset into big_keys / big_counts} triples sorted by bucket.
    int64_t big_src = -1;
    unsigned big_D = 0;
    if (cur.over) {                                    // uniform
      int lo_i = 0, hi_i = n_big;
      while (lo_i < hi_i) {
        const int mid = (lo_i + hi_i) >> 1;
        if (big_table[3 * mid] < cur.b) lo_i = mid + 1; else hi_i = mid;
      }
      if (lo_i < n_big && big_table[3 * lo_i] == cur.b) {
        big_D = (unsigned)fn_uniform(big_table[3 * lo_i + 1]);
        big_src = fn_uniform(big_table[3 * lo_i + 2]);
      } else if (tid == 0) {
        atomicOr(&state[FS_FLAGS], 1ull);
      }
    }

    // (before this wavefront's loads of the next keys: the memory counter is in-order, younger loads would be waited for)
    if (have_prev && wave == 0) resolve_prev();
    // the next bucket: its offsets arrived during the previous iteration; start the loads of its keys now
    const fn_bucket nxt = fn_open(n_buckets, nn_b, fn_uniform(nn_lo), fn_uniform(nn_hi), nn_t, pstride);
    uint64_t kn[FN_ITEMS];
    {
      const int t = fn_fresh(tid);
#pragma unroll
      for (int q = 0; q < FN_ITEMS; ++q) {
        const int i = t + q * FN_THREADS;
        if (i < nxt.nb) kn[q] = __builtin_nontemporal_load(&(A + nxt.src)[(unsigned)i]);   // scalar base + 32-bit lane offset: no per-lane 64-bit addresses
      }
    }
    unsigned D = 0;
    bool all_one = true;                               // every multiplicity of the bucket is 1
    if (nb == 0) {                                     // empty (or over-capacity) bucket: only its place in the chain
      D = big_D;
      publish_count(cur.b, D);
      __syncthreads();                                 // keeps the reads of the ticket word a barrier away from its next write
      if (have_prev) emit_prev();
    } else {
      // counting sort on the next sbits bits: exclusive scan of the bin counts, keys to their bins
      {
        unsigned c[FN_BINS_PER_LANE], sum = 0;
        const int t = fn_fresh(tid);
#pragma unroll
        for (int j = 0; j < FN_BINS_PER_LANE; ++j) {
          const unsigned bi = t * FN_BINS_PER_LANE + j;
          c[j] = (bi < SB) ? bins[bi] : 0;
          sum += c[j];
        }
        const unsigned inc = wave_inclusive_scan(sum);
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        unsigned run = inc - sum;
        for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
        for (int j = 0; j < FN_BINS_PER_LANE; ++j) {
          const unsigned bi = t * FN_BINS_PER_LANE + j;
          if (bi < SB) bins[bi] = run;
          run += c[j];
        }
        if (tid == 0) bins[SB] = (unsigned)nb;
      }
      if (have_prev) emit_prev();                      // the stage is free for this bucket after the next barrier
      __syncthreads();
      // From here on r[q] packs what the ranking needs about a key, in ONE register (the kernel sits at the 128-VGPR
      // limit of 1024-thread workgroups and every spilled value costs a wait for all loads in flight):
      // {slot in the stage : 13 | rank in its bin : 13 | earlier keys that are smaller (dense steps) : 6}
      static_assert(FN_CAP <= 8192, "slot and rank are packed in 13 bits each");
#pragma unroll
      for (int q = 0; q < FN_ITEMS; ++q) {
        if ((valid >> q) & 1u) {
          const unsigned slot = bins[(unsigned)(k[q] >> sshift) & (SB - 1)] + r[q];
          stage[slot] = k[q];
          aux[slot] = 0;
          r[q] = slot | (r[q] << 13);
        }
      }
      __syncthreads();
      // Triangular pass over the (tiny) bins: every key meets the keys in EARLIER slots of its bin exactly once.
      // An earlier key that is smaller adds to this key's rank; one that is larger gets its own rank bumped (LDS
      // atomic); an equal one — the first hit of the ascending walk is that key's first occurrence — makes this
      // key a duplicate: it adds itself to the first occurrence's counter and drops out.  The eight keys of a lane
      // advance together (eight independent LDS reads per step instead of eight latency-bound loops).
      unsigned active = 0, dup = 0;
#pragma unroll
      for (int q = 0; q < FN_ITEMS; ++q)
        if (((valid >> q) & 1u) && FN_RANK(r[q]) > 0) active |= 1u << q;
      // Two dense steps (most walks are that short) ...
      for (unsigned step = 0; step < 2 && __any(active != 0); ++step) {
#pragma unroll
        for (int q = 0; q < FN_ITEMS; ++q) {
          if ((active >> q) & 1u) {
            const unsigned j = FN_SLOT(r[q]) - FN_RANK(r[q]) + step;
            const uint64_t y = stage[j];
            if (y == k[q]) { atomicAdd(&aux[j], 0x10000u); dup |= 1u << q; active &= ~(1u << q); }
            else {
              if (y < k[q]) r[q] += 1u << 26; else atomicAdd(&aux[j], 1u);
              if (step + 1 >= FN_RANK(r[q])) active &= ~(1u << q);
            }
          }
        }
      }
      {
        const unsigned nd = wave_sum((unsigned)__popc(dup));
        if (lane == 0 && nd) atomicAdd(&sh_dups[parity], nd);
      }
      // ... then the few keys with longer walks (~10 %) are compacted into a list private to the wavefront, ONE
      // per lane, instead of sweeping all eight register slots of every lane for a handful of stragglers.  Their
      // results travel through LDS: rank increments in aux (low half), "I am a duplicate" in bit 31.
      if (__any(active != 0)) {                        // wave-uniform
        unsigned n_items = 0;
#pragma unroll
        for (int q = 0; q < FN_ITEMS; ++q) {
          const bool a = (active >> q) & 1u;
          const uint64_t m = __ballot(a);
          if (a) wlist[n_items + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)FN_SLOT(r[q]);
          n_items += (unsigned)__popcll(m);
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (unsigned i0 = 0; i0 < n_items; i0 += 64) {
          if (i0 + lane < n_items) {
            const unsigned slot = wlist[i0 + lane];
            const uint64_t x = stage[slot];
            const unsigned b0 = bins[(unsigned)(x >> sshift) & (SB - 1)];
            for (unsigned j = b0 + 2; j < slot; ++j) {
              const uint64_t y = stage[j];
              if (y == x) {
                atomicAdd(&aux[j], 0x10000u);
                atomicOr(&aux[slot], 0x80000000u);
                atomicAdd(&sh_dups[parity], 1u);
                break;
              }
              atomicAdd(&aux[y < x ? slot : j], 1u);
            }
          }
        }
      }
      __syncthreads();
      const unsigned n_dups = (unsigned)__builtin_amdgcn_readfirstlane((int)sh_dups[parity]);
      D = (unsigned)nb - n_dups;                       // distinct keys of the bucket
      if (n_dups) {                                    // uniform: duplicates found by the list walkers above
#pragma unroll
        for (int q = 0; q < FN_ITEMS; ++q)
          if (((valid >> q) & 1u) && (aux[FN_SLOT(r[q])] >> 31)) dup |= 1u << q;
      }
      const unsigned first_bits = valid & ~dup;
      unsigned idx[FN_ITEMS];
      publish_count(cur.b, D);
      if (n_dups == 0) {                               // uniform: the common case for well-spread k-mers
#pragma unroll
        for (int q = 0; q < FN_ITEMS; ++q)
          if ((valid >> q) & 1u) idx[q] = FN_SLOT(r[q]) - FN_RANK(r[q]) + FN_LESS(r[q]) + (aux[FN_SLOT(r[q])] & 0xffffu);
#pragma unroll
        for (int q = 0; q < FN_ITEMS; ++q)
          if ((valid >> q) & 1u) stage[idx[q]] = k[q];
      } else {
        // Buckets with duplicates: the first occurrences (bit mask + popcount prefix) are compacted to the front of
        // the stage in slot order (bins stay contiguous) and ranked among themselves, so the work per key does not
        // grow with the multiplicities; a first occurrence's multiplicity is 1 + the duplicates that found it above.
        all_one = false;
        unsigned bs[FN_ITEMS], lt[FN_ITEMS];           // (this rarely taken branch works on the unpacked fields)
#pragma unroll
        for (int q = 0; q < FN_ITEMS; ++q) {
          const unsigned slot = FN_SLOT(r[q]);
          r[q] = FN_RANK(r[q]);
          bs[q] = slot - r[q];
          lt[q] = 0;
          if ((first_bits >> q) & 1u) atomicOr(&fmask32[slot >> 5], 1u << (slot & 31));
        }
        __syncthreads();
        // every wavefront scans the popcounts of the mask words in its own registers (lane l: words FN_WPL*l ..)
        const unsigned c0 = __popcll(fmask[FN_WPL * lane]);
        const unsigned c1 = FN_WPL == 2 ? __popcll(fmask[FN_WPL * lane + 1]) : 0u;
        const unsigned pinc = wave_inclusive_scan(c0 + c1);
        const unsigned pex = pinc - c0 - c1;
        auto distinct_before = [&](unsigned x) -> unsigned {   // first occurrences in slots < x (all lanes must call)
          const unsigned w = min(x >> 6, (unsigned)FN_WORDS - 1);
          const unsigned pw = __shfl(pex, w / FN_WPL, 64), cw = __shfl(c0, w / FN_WPL, 64);
          const uint64_t below = x >= (unsigned)FN_CAP ? ~0ull : ((1ull << (x & 63)) - 1ull);
          return pw + ((FN_WPL == 2 && (w & 1)) ? cw : 0u) + __popcll(fmask[w] & below);
        };
        unsigned todo = 0;                             // idx[q] = compact start of the bin, r[q] = first occurrences in it,
#pragma unroll                                         // lt[q] = compact slot | multiplicity << 16
        for (int q = 0; q < FN_ITEMS; ++q) {
          const bool is_first = (first_bits >> q) & 1u;
          unsigned e = bs[q];
          if (is_first) e = bins[((unsigned)(k[q] >> sshift) & (SB - 1)) + 1];
          const unsigned slot = bs[q] + r[q];
          const unsigned cs = distinct_before(bs[q]), ce = distinct_before(e), c = distinct_before(slot);
          const unsigned m = is_first ? 1u + ((aux[slot] >> 16) & 0x7fffu) : 0u;
          idx[q] = cs;
          r[q] = is_first ? ce - cs : 0u;
          lt[q] = c | (m << 16);
          if (is_first && ce - cs > 1) todo |= 1u << q;
        }
        __syncthreads();                               // every lane has read its slot's counter
#pragma unroll
        for (int q = 0; q < FN_ITEMS; ++q)
          if ((first_bits >> q) & 1u) stage[lt[q] & 0xffffu] = k[q];
        __syncthreads();
        for (unsigned step = 0; __any(todo != 0); ++step) {            // rank inside the compacted bin, in r[q] >> 16
#pragma unroll
          for (int q = 0; q < FN_ITEMS; ++q) {
            if ((todo >> q) & 1u) {
              r[q] += (stage[idx[q] + step] < k[q]) ? 0x10000u : 0u;
              if (step + 1 >= (r[q] & 0xffffu)) todo &= ~(1u << q);
            }
          }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < FN_ITEMS; ++q) {
          if ((first_bits >> q) & 1u) {
            stage[idx[q] + (r[q] >> 16)] = k[q];
            aux[idx[q] + (r[q] >> 16)] = lt[q] >> 16;
          }
        }
        if (tid < FN_WORDS) fmask[tid] = 0;
        // (the prefetched keys of the next bucket are loaded again here, so that their registers are free
        // throughout this rarely taken branch)
#pragma unroll
        for (int q = 0; q < FN_ITEMS; ++q) {
          const int i = tid + q * FN_THREADS;
          if (i < nxt.nb) kn[q] = (A + nxt.src)[(unsigned)i];
        }
      }
    }
    // ---- tail: this bucket's bins are free; the next bucket takes its ranks in the other bin array (zeroed one
    // iteration ago), so its counting sort can start right after the output below
    const int tt = fn_fresh(tid);
    for (unsigned i = tt; i <= SB; i += FN_THREADS) bins[i] = 0;
    if (tid == 0) sh_dups[parity ^ 1] = 0;
    valid = 0;
#pragma unroll
    for (int q = 0; q < FN_ITEMS; ++q) {
      const int i = tt + q * FN_THREADS;
      if (i < nxt.nb) {
        k[q] = kn[q];
        r[q] = atomicAdd(&bins_next[(unsigned)(k[q] >> sshift) & (SB - 1)], 1u);
        valid |= 1u << q;
      }
    }
    // This iteration's ticket (the bucket after the next one).  Taken by wavefront 0 AFTER its look-back: the
    // order of the tickets then follows the order in which the buckets complete, which keeps the rounds intact
    // (taken by another wavefront, or earlier in the iteration, the launch becomes unstable: 35-44 / 57-67 ms
    // instead of 37.4 per 3e9 keys).
    // The walk over the predecessors' counts starts here, as late as possible: a predecessor that has not published
    // yet costs a second, blocking poll at the top of the next iteration (measured: 0.61 extra polls per bucket
    // from here, 0.82 when the poll is issued before the ranks above).
    if (!REDO && wave == 0) lb_v = lb_poll(cur.b - 1);
    if (tid == 0) sh[0] = (long long)atomicAdd(&state[FS_TICKET], 1ull);
    __syncthreads();
    nn_t = fn_uniform(sh[0]);
    nn_b = REDO ? fn_uniform(bucket_of(nn_t)) : nn_t;
    if (nn_b < n_buckets) { nn_lo = bucket_off[nn_b]; nn_hi = bucket_off[nn_b + 1]; }   // consumed (made scalar) next iteration
    have_prev = true;
    prev_b = cur.b;
    prev_t = cur.t;
    prev_D = D;
    prev_one = all_one;
    prev_big = big_src;
    unsigned* t = bins; bins = bins_next; bins_next = t;
    parity ^= 1;
    cur = nxt;
  }
  if (have_prev) {                                     // the last bucket of this workgroup
    if (wave == 0) resolve_prev();
    __syncthreads();
    emit_prev();
  }
}
// ===================================================================================================================
// Fast finishing kernel — the path (nearly) duplicate-free buckets take, i.e. the k-mers of S-uniform-like reads.
//
// What the counters of the general kernel above said (profiles/r01_sq_counters.json): 58 % of its wave cycles wait,
// 55 % of its LDS cycles are bank conflicts, and every bucket pays ~6 us of fixed latency (ticket atomic, chained
// look-back, eight barriers) with ONE 1024-thread workgroup per CU to hide it behind.  A first version of this kernel
// kept a look-back (chain-free: every workgroup summed the distinct counts of the G buckets in flight) and still ran
// in lock step: whoever needs the counts of buckets that are being sorted right now waits for the slowest of them,
// every round, and the chip's reads and writes arrive in bursts.  So this kernel waits for NOBODY:
//   * the output position of bucket b is bucket_off[b] minus the duplicates found in earlier buckets SO FAR.  Buckets
//     with a repeated key are rare here (six in 2^20 for 6e9 random 31-mers); each one is announced in a small log
//     {bucket, duplicates} that every workgroup re-reads once per bucket.  A bucket emitted before an earlier
//     bucket's announcement arrived sits a few slots too far right.  That is found afterwards — finish_check_kernel
//     compares the position every bucket used with the exclusive scan of the distinct counts — and such buckets,
//     the buckets with duplicates themselves (which this kernel cannot emit: it does not count multiplicities) and
//     the neighbours their stray writes touched are redone by finish_sorted_kernel<REDO> at their true positions:
//     a few thousand buckets per million.  More than FF_LOG announcements (duplicate-heavy keys) abort the kernel
//     within one bucket per workgroup and the general kernel takes everything;
//   * 512 threads, <= 80 KiB of LDS (the stage of 64-bit keys + 16-bit bin offsets, packed two per word): two
//     workgroups share a CU and overlap each other's barriers; buckets are dealt round robin (no ticket atomic) and a
//     workgroup's next keys are loaded while the current bucket is ranked;
//   * the ranking inside the (tiny) bins is done by SLOT OWNERS: after the counting sort has grouped the keys by bin,
//     lane l of a wavefront owns slot s = 64 c + l of the stage and compares its key with the neighbours s -/+ d
//     inside its bin — consecutive lanes read consecutive LDS words (no bank conflicts, no atomics, no per-slot
//     counters), and a key's final place is bin start + number of smaller keys in the bin.  Keys of one bin differ
//     only below bit `sshift`: for sshift <= 32 (NARROW) only the low words are read and compared;
//   * a key leaves for HBM straight from its owner's registers, at base + place: the stores of a wavefront cover
//     the same lines as 64 consecutive slots, permuted inside the bins.
constexpr int FF_THREADS = 512;
constexpr int FF_WAVES = FF_THREADS / 64;
constexpr int FF_ITEMS = 15;
constexpr int FF_CAP = FF_THREADS * FF_ITEMS;            // 7680 keys
constexpr int FF_SLICE = 64 * FF_ITEMS;                  // slots owned by one wavefront
#ifndef FF_BITS
#define FF_BITS 13
#endif
constexpr int FF_MAXBITS = FF_BITS;
constexpr int FF_MAXBINS = 1 << FF_MAXBITS;
constexpr int FF_SCAN_DW = FF_MAXBINS / 2 / FF_THREADS;  // packed bin words scanned by one lane (8)
constexpr int FF_NEAR = 64;                              // guard slots around the stage: neighbour reads are not clamped
constexpr int FF_LOG = 64;                               // announcements of buckets with duplicates (one per lane of the reader)
#ifndef FF_UNROLL
#define FF_UNROLL 4
#endif
#ifndef FF_WG
#define FF_WG 3                                          // chunks of 64 slots whose neighbour walks advance together
#endif
static_assert(FF_ITEMS % FF_WG == 0, "whole groups");
constexpr size_t FF_OFF_STAGE = (size_t)FF_NEAR * 8;                               // FF_NEAR guard slots in front of the stage ...
constexpr size_t FF_OFF_P = FF_OFF_STAGE + (size_t)(FF_CAP + FF_NEAR) * 8;          // ... and behind it
constexpr size_t FF_OFF_WSUM = FF_OFF_P + (((size_t)(FF_MAXBINS + 2) * 2 + 15) & ~(size_t)15);
constexpr size_t FF_OFF_SH = FF_OFF_WSUM + 3 * FF_WAVES * 4;
constexpr size_t FF_LDS = FF_OFF_SH + 4 * 8;
static_assert(2 * FF_LDS <= 160 * 1024, "two workgroups per CU");
static_assert(FF_CAP <= (1 << 13), "slot / rank are packed in 13 bits");
// d_state words (fast path): [0] flags (1 = over-capacity bucket without a pre-counted entry, 2 = general kernel's
// look-back gave up, 4 = too many buckets with duplicates), [2] distinct keys, [3] redo list length, [4] announcements,
// [8, 8 + FF_LOG) the announcements {valid : 1 | bucket : 31 | duplicates : 32}, then per bucket: distinct count
// (int64, scanned in place afterwards), {not emitted : 1 | duplicates known when emitted : 31}, redo mark, and the
// redo list (ids, bases).
// The fast kernel's ticket counter has a 128-byte line of its own ([96, 112)): every workgroup hits it once per bucket,
// and the flags / announcements, which every workgroup reads once per bucket, must not share a line with it.
constexpr int FS_SPARE [[maybe_unused]] = FS_LOG + FF_LOG;
constexpr unsigned FF_BAD = 0x80000000u;

template <int N> struct ff_int { static constexpr int value = N; };

template <bool NARROW>
__global__ __launch_bounds__(FF_THREADS, 4) void finish_fast_kernel(
    const uint64_t* __restrict__ A, const int64_t* __restrict__ bucket_off, const int64_t* __restrict__ out_off,
    int64_t n_buckets, int sshift, int sbits, unsigned long long* __restrict__ header, int64_t* __restrict__ Dv,
    unsigned* __restrict__ meta, uint64_t* __restrict__ keys_out, int64_t* __restrict__ counts_out,
    const int64_t* __restrict__ big_table, int n_big, const uint64_t* __restrict__ big_keys,
    const int64_t* __restrict__ big_counts, int64_t pstride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* stage = reinterpret_cast<uint64_t*>(smem + FF_OFF_STAGE);
  unsigned* P32 = reinterpret_cast<unsigned*>(smem + FF_OFF_P);                 // bins: counts, then exclusive offsets
  const unsigned short* P16 = reinterpret_cast<const unsigned short*>(smem + FF_OFF_P);
  unsigned* wsum = reinterpret_cast<unsigned*>(smem + FF_OFF_WSUM);             // [0..7] scan, [8..15] duplicates, [16..23] long bins
  long long* sh = reinterpret_cast<long long*>(smem + FF_OFF_SH);               // [0] duplicates known before this bucket, [1] abort, [2] next ticket
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned SB = 1u << sbits;
  const unsigned n_dw = SB > 1 ? SB >> 1 : 1u;           // words holding the bins (+ one for the end offset P[SB])
  const unsigned dwl = n_dw >= FF_THREADS ? n_dw / FF_THREADS : 1u;   // ... scanned by one lane
  const int64_t G = gridDim.x;
  unsigned long long* dup_log = header + FS_LOG;

  for (unsigned i = tid; i <= n_dw; i += FF_THREADS) P32[i] = 0;
  if (tid < FF_NEAR) stage[tid - FF_NEAR] = ~0ull;       // guard keys in front of slot 0
  __syncthreads();

#ifdef FF_PHASES
  unsigned long long ph_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_last = __builtin_readcyclecounter();
#define FF_MARK(i) { const unsigned long long now__ = __builtin_readcyclecounter(); ph_t[i] += now__ - ph_last; ph_last = now__; }
#else
#define FF_MARK(i)
#endif
  // Software pipeline: the keys of bucket b + G are loaded into k[] as soon as the keys of bucket b have been placed
  // in the stage (the registers are free from then on) and stay in flight through the rest of the iteration; the
  // offsets of bucket b + 2 G are fetched an iteration before that.
  uint64_t k[FF_ITEMS];
  struct bucket_t { int64_t lo, out; int nb; int64_t size; int64_t src; };   // src: where the keys lie (lo, or b * pstride)
  auto fetch_offsets = [&](int64_t bb, int64_t& o0, int64_t& o1, int64_t& oo) {      // (scalar loads; consumed an iteration later)
    o0 = 0; o1 = 0; oo = 0;
    if (bb < n_buckets) { o0 = bucket_off[bb]; o1 = bucket_off[bb + 1]; oo = out_off[bb]; }
  };
  auto open_bucket = [&](int64_t bb, int64_t o0, int64_t o1, int64_t oo) {
    bucket_t x;
    x.lo = fn_uniform(o0);
    x.size = fn_uniform(o1) - x.lo;
    x.out = fn_uniform(oo);
    x.src = pstride ? bb * pstride : x.lo;
    x.nb = x.size > FF_CAP ? 0 : (int)x.size;
    return x;
  };
  // The usual bucket (n / 2^bits keys, <= 6000) fills 12 of the 15 items: the last three are only touched for the
  // larger ones (uniform branches around whole groups of instructions, none inside).
  constexpr int FF_USUAL = 12;
  auto load_keys = [&](const bucket_t& x) {              // k[q] = key tid + 512 q of the bucket (clamped: branch-free)
    const uint64_t* Ab = A + x.src;                      // scalar base + 32-bit lane offsets
    const int t = fn_fresh(tid);
    if (x.nb > 0) {
#pragma unroll
      for (int q = 0; q < FF_USUAL; ++q) k[q] = __builtin_nontemporal_load(&Ab[(unsigned)min(t + q * FF_THREADS, x.nb - 1)]);
      if (x.nb > FF_USUAL * FF_THREADS) {
#pragma unroll
        for (int q = FF_USUAL; q < FF_ITEMS; ++q) k[q] = __builtin_nontemporal_load(&Ab[(unsigned)min(t + q * FF_THREADS, x.nb - 1)]);
      }
    }
  };
#pragma unroll
  for (int q = 0; q < FF_ITEMS; ++q) k[q] = 0;
  int64_t f0, f1, fo;
  fetch_offsets((int64_t)blockIdx.x, f0, f1, fo);
  bucket_t cur = open_bucket((int64_t)blockIdx.x, f0, f1, fo);
  load_keys(cur);
  fetch_offsets((int64_t)blockIdx.x + G, f0, f1, fo);
  // The first two buckets of a workgroup are blockIdx and blockIdx + G; after that the buckets are handed out by a
  // ticket counter, three iterations ahead (ticket -> offsets -> keys -> sort), so that the buckets in flight stay
  // within a few rounds of each other however unevenly the workgroups run: a bucket emitted before an EARLIER
  // bucket's announcement has to be redone, and without this the fast workgroups run tens of rounds ahead.
  int64_t b = blockIdx.x, b_nxt = b + G, b_n2 = b_nxt + G;
  if (tid == 0) b_n2 = 2 * G + (int64_t)atomicAdd(&header[FS_FTICKET], 1ull);
  if (tid == 0) sh[2] = b_n2;
  __syncthreads();
  b_n2 = fn_uniform(sh[2]);
  for (; b < n_buckets;) {
    const bucket_t nxt = open_bucket(b_nxt, f0, f1, fo); // bucket b_nxt (its offsets were fetched an iteration ago)
    fetch_offsets(b_n2, f0, f1, fo);
    unsigned long long tk = 0;                           // the ticket after b_n2: in flight until the keys are placed
    if (tid == 0) tk = atomicAdd(&header[FS_FTICKET], 1ull);
    int64_t b_n3_latched = 0;
    auto b_nxt_shift = [&](int64_t v) { b_n3_latched = v; };
    const int nb = cur.nb;
    // wavefront 0: the announcements so far (in flight until the keys are placed)
    unsigned long long lv = 0, fl = 0;
    if (wave == 0) {
      lv = __hip_atomic_load(dup_log + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      fl = __hip_atomic_load(header + FS_FLAGS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    auto publish_known = [&]() {                         // wavefront 0, before a barrier
      const bool mine = (lv >> 63) && (int64_t)((lv >> 32) & 0x7fffffffull) < b;
      const unsigned known = wave_sum(mine ? (unsigned)lv : 0u);
      if (lane == 0) { sh[0] = (long long)known; sh[1] = (long long)(fl & 4ull); }
    };
    unsigned D = 0, bad = 0;
    if (nb == 0) {                                       // empty, or a heavy-hitter bucket counted by the caller beforehand
      if (wave == 0) publish_known();
      if (tid == 0) sh[2] = 2 * G + (long long)tk;
      __syncthreads();
      const unsigned known = (unsigned)fn_uniform(sh[0]);
      const int64_t b_n3 = fn_uniform(sh[2]);
      if (fn_uniform(sh[1])) return;
      b_nxt_shift(b_n3);
      if (cur.size > 0) {
        int lo_i = 0, hi_i = n_big;
        while (lo_i < hi_i) {
          const int mid = (lo_i + hi_i) >> 1;
          if (big_table[3 * mid] < b) lo_i = mid + 1; else hi_i = mid;
        }
        if (lo_i < n_big && big_table[3 * lo_i] == b) {
          D = (unsigned)fn_uniform(big_table[3 * lo_i + 1]);
          const int64_t src = fn_uniform(big_table[3 * lo_i + 2]);
          uint64_t* ko = keys_out + (cur.out - known);
          int64_t* co = counts_out + (cur.out - known);
          for (unsigned i = (unsigned)fn_fresh(tid); i < D; i += FF_THREADS) {
            ko[i] = big_keys[src + i];
            co[i] = big_counts[src + i];
          }
        } else if (tid == 0) {
          atomicOr(&header[FS_FLAGS], 1ull);
        }
      }
      if (tid == 0) { Dv[b] = D; meta[b] = known; }
      load_keys(nxt);
      __syncthreads();
    } else {
      // ---- counting sort on the next sbits bits: ranks from LDS atomics on the packed 16-bit bin counters
      // (branch-free: the items past the end add 0 to a bin, so that all fifteen atomics are in flight together)
      unsigned rb[FF_ITEMS];                             // {bin : 13 | rank in the bin : 13}
      const bool large = nb > FF_USUAL * FF_THREADS;     // (uniform)
      {
        const int t0 = fn_fresh(tid);
        unsigned old[FF_ITEMS];
#pragma unroll
        for (int q = 0; q < FF_USUAL; ++q) {
          const unsigned bin = (unsigned)(k[q] >> sshift) & (SB - 1);
          rb[q] = bin;
          old[q] = atomicAdd(&P32[bin >> 1], (t0 + q * FF_THREADS < nb ? 1u : 0u) << ((bin & 1u) * 16u));
        }
        if (large) {
#pragma unroll
          for (int q = FF_USUAL; q < FF_ITEMS; ++q) {
            const unsigned bin = (unsigned)(k[q] >> sshift) & (SB - 1);
            rb[q] = bin;
            old[q] = atomicAdd(&P32[bin >> 1], (t0 + q * FF_THREADS < nb ? 1u : 0u) << ((bin & 1u) * 16u));
          }
        } else {
#pragma unroll
          for (int q = FF_USUAL; q < FF_ITEMS; ++q) { rb[q] = 0; old[q] = 0; }
        }
#pragma unroll
        for (int q = 0; q < FF_ITEMS; ++q) rb[q] |= __builtin_amdgcn_ubfe(old[q], (rb[q] & 1u) * 16u, 16u) << 13;
      }
      __syncthreads();                                   // (1) every rank is taken
      FF_MARK(0)
      bool short_bins;                                   // no bin is longer than a chunk
      int t_walk;                                        // longest bin - 1: how far a key's bin can reach on either side
      {
        unsigned c[FF_SCAN_DW], sum = 0, longest = 0;
        const unsigned t1 = (unsigned)fn_fresh(tid);
#pragma unroll
        for (int j = 0; j < FF_SCAN_DW; ++j) {
          const unsigned w = t1 * dwl + j;
          c[j] = ((unsigned)j < dwl && w < n_dw) ? P32[w] : 0u;
          sum += (c[j] & 0xffffu) + (c[j] >> 16);
          longest = max(longest, max(c[j] & 0xffffu, c[j] >> 16));
        }
        const unsigned inc = wave_inclusive_scan(sum);
        longest = wave_max(longest);
        if (lane == 63) { wsum[wave] = inc; wsum[2 * FF_WAVES + wave] = longest; }
        __syncthreads();                                 // (2)
        FF_MARK(1)
        unsigned run = inc - sum, longs = 0;
#pragma unroll
        for (int w = 0; w < FF_WAVES; ++w) {
          run += w < wave ? wsum[w] : 0u;
          longs = max(longs, wsum[2 * FF_WAVES + w]);
        }
        t_walk = __builtin_amdgcn_readfirstlane((int)longs) - 1;
        short_bins = t_walk < FF_NEAR;
#pragma unroll
        for (int j = 0; j < FF_SCAN_DW; ++j) {
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
      FF_MARK(2)
      {
        const int t2 = fn_fresh(tid);
        unsigned slot[FF_ITEMS];
#pragma unroll
        for (int q = 0; q < FF_USUAL; ++q) slot[q] = P16[rb[q] & 0x1fffu] + (rb[q] >> 13);
#pragma unroll
        for (int q = 0; q < FF_USUAL; ++q)
          if (t2 + q * FF_THREADS < nb) stage[slot[q]] = k[q];
        if (large) {
#pragma unroll
          for (int q = FF_USUAL; q < FF_ITEMS; ++q) slot[q] = P16[rb[q] & 0x1fffu] + (rb[q] >> 13);
#pragma unroll
          for (int q = FF_USUAL; q < FF_ITEMS; ++q)
            if (t2 + q * FF_THREADS < nb) stage[slot[q]] = k[q];
        }
      }
      if (wave == FF_WAVES - 1) stage[nb + fn_fresh(lane)] = ~0ull;   // guard keys behind the bucket (nobody places a key there)
      if (tid == 0) sh[2] = 2 * G + (long long)tk;       // (before the loads below: older than them in the memory counter)
      load_keys(nxt);                                    // k[] is free: the next bucket's keys, in flight until the next iteration
      if (wave == 0) publish_known();
      __syncthreads();                                   // (4) the keys are grouped by bin
      FF_MARK(3)
      const unsigned known = (unsigned)fn_uniform(sh[0]);
      const int64_t b_n3 = fn_uniform(sh[2]);
      if (fn_uniform(sh[1])) return;                     // (uniform) too many buckets with duplicates: the general kernel takes over
      b_nxt_shift(b_n3);
      uint64_t* ko = keys_out + (cur.out - known);       // scalar bases, 32-bit lane offsets
      int64_t* co = counts_out + (cur.out - known);
      // ---- slot owners.  The keys are grouped by bin, and the bins ascend: among the t keys on either side of a slot
      // (t = longest bin - 1) every key of an earlier bin is smaller and every key of a later bin larger, so
      //   place = s - min(t, s) + #{d <= t: stage[s - d] < x} + #{d <= t: stage[s + d] < x}
      // with no need to know where the bin starts or ends: two LDS reads, two compares and two adds per step, the same
      // for all lanes (all-ones guard keys lie in front of slot 0 and behind slot nb - 1).  An equal key in an EARLIER
      // slot makes a key a duplicate.
      unsigned ndup = 0;
      const int l3 = fn_fresh(lane);
      const int slice0 = wave * FF_SLICE;
      if (short_bins) {                                  // uniform: t < 64
        // (the bins are not needed any more: cleared for the next bucket's ranks, which start after barrier 5)
        for (unsigned i = (unsigned)fn_fresh(tid); i <= n_dw; i += FF_THREADS) P32[i] = 0;
        // out, first half: the counts are all 1 (a bucket where they are not is redone) and their places do not depend on
        // the ranking — a third of the kernel's traffic leaves NOW, sixteen bytes per lane, and drains while the walks
        // below compute (issued per slot behind the walks, 8 bytes per lane, it left in the same burst as the keys)
        {
          typedef long long i64x2 __attribute__((ext_vector_type(2)));
          const unsigned head = (unsigned)((reinterpret_cast<uintptr_t>(co) >> 3) & 1u);      // (nb >= 1 here)
          if (head && tid == 0) __builtin_nontemporal_store((int64_t)1, &co[0]);
          const unsigned pairs = ((unsigned)nb - head) >> 1;
          i64x2 ones;
          ones.x = 1;
          ones.y = 1;
          for (unsigned p = (unsigned)fn_fresh(tid); p < pairs; p += FF_THREADS)
            __builtin_nontemporal_store(ones, reinterpret_cast<i64x2*>(co + head + 2 * p));
          if ((((unsigned)nb - head) & 1u) && tid == 0) __builtin_nontemporal_store((int64_t)1, &co[nb - 1]);
        }
#pragma unroll
        for (int c0 = 0; c0 < FF_ITEMS; c0 += FF_WG) {
          if (slice0 + c0 * 64 < nb) {                   // uniform: the group holds keys
            uint64_t x[FF_WG];
            unsigned cnt[FF_WG];
            const int sl0 = slice0 + c0 * 64 + l3;
            const uint64_t* mid = stage + sl0;
#pragma unroll
            for (int u = 0; u < FF_WG; ++u) { x[u] = mid[64 * u]; cnt[u] = 0; }
            bool dup = false;
#pragma unroll FF_UNROLL
            for (int d = 1; d <= t_walk; ++d) {
              uint64_t y[FF_WG], z[FF_WG];
#pragma unroll
              for (int u = 0; u < FF_WG; ++u) { y[u] = mid[64 * u - d]; z[u] = mid[64 * u + d]; }
#pragma unroll
              for (int u = 0; u < FF_WG; ++u) {
                cnt[u] += (y[u] < x[u] ? 1u : 0u) + (z[u] < x[u] ? 1u : 0u);        // (as doubles, V_CMP_*_F64: 29.8 vs 28.1 ms)
                dup |= y[u] == x[u];
              }
            }
            // (a key may have several equal neighbours: with a duplicate in the group, count the duplicates exactly)
            if (__any(dup)) {
#pragma unroll
              for (int u = 0; u < FF_WG; ++u) {
                bool is_dup = false;
                for (int d = 1; d <= t_walk; ++d) is_dup |= mid[64 * u - d] == x[u];
                ndup += (is_dup && sl0 + 64 * u < nb) ? 1u : 0u;
              }
            }
            // out, second half: every key at its place
#pragma unroll
            for (int u = 0; u < FF_WG; ++u) {
              const int s = sl0 + 64 * u;
              if (s < nb) __builtin_nontemporal_store(x[u], &ko[(unsigned)(s - min(t_walk, s)) + cnt[u]]);
            }
          }
        }
      } else {
        // a bin longer than a chunk (skewed keys): only the exact number of duplicates is taken here — the bucket is
        // redone.  A key is a duplicate iff an equal key sits in an earlier slot of its bin; the walk stops at the first
        // one, so long runs of one key cost one step each.
        bad = FF_BAD;
#pragma unroll 1
        for (int c = 0; c < FF_ITEMS; ++c) {
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
      }
      ndup = wave_sum(ndup);
      if (lane == 0) wsum[FF_WAVES + wave] = ndup;
      __syncthreads();                                   // (5) every neighbour has been read
      FF_MARK(4)
      unsigned dups = 0;
#pragma unroll
      for (int w = 0; w < FF_WAVES; ++w) dups += wsum[FF_WAVES + w];
      dups = (unsigned)__builtin_amdgcn_readfirstlane((int)dups);
      D = (unsigned)nb - dups;
      if (dups) bad = FF_BAD;
      if (tid == 0) {
        Dv[b] = D;
        meta[b] = known | bad;
        if (dups) {                                      // announce: later buckets start `dups` slots further left
          const unsigned long long at = atomicAdd(&header[FS_NLOG], 1ull);
          if (at < (unsigned long long)FF_LOG)
            __hip_atomic_store(dup_log + at, (1ull << 63) | ((unsigned long long)b << 32) | dups, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
          else
            atomicOr(&header[FS_FLAGS], 4ull);
        }
      }
      if (!short_bins) {                                 // (uniform) the long-bin walk read the bins until barrier 5
        for (unsigned i = (unsigned)fn_fresh(tid); i <= n_dw; i += FF_THREADS) P32[i] = 0;
        __syncthreads();
      }
      FF_MARK(5)
    }
    cur = nxt;
    b = b_nxt;
    b_nxt = b_n2;
    b_n2 = b_n3_latched;
  }
#ifdef FF_PHASES
  if (tid == 64) for (int i = 0; i < 8; ++i) atomicAdd(header + FS_SPARE + i, ph_t[i]);      // (experiment builds only)
#endif
}

// ---- after the fast kernel: which buckets have to be redone -----------------------------------------------------------
// T = exclusive scan of the distinct counts (T[b] = true output position of bucket b, T[n] = distinct keys).  A bucket
// is redone if it was not emitted properly (duplicates / long bins), if it was emitted at another position than T[b],
// or if the stray writes of such a bucket reached into its region.
__global__ void finish_check_kernel(const int64_t* __restrict__ T, const unsigned* __restrict__ meta,
                                    const int64_t* __restrict__ bucket_off, const int64_t* __restrict__ out_off,
                                    int64_t n_buckets, unsigned* __restrict__ marks, const unsigned long long* __restrict__ header) {
  BNPK_VGPR_FLOOR_32();                                   // (24 VGPRs otherwise: see common.h)
  // The fast kernel gave up (keys that repeat: every real read set): most distinct counts were never written, T is the scan of
  // whatever the memory held, and the walk over "every bucket the stray writes touched" below can then run over a million
  // buckets per thread — 145 ms on the first call of a process, over fresh memory (rocprofv3, round 6: the maximum of finish_check
  // in profiles/r06_k21_kernel_stats.txt); later calls found the previous call's counts there and were quick, by luck.
  if (header[FS_FLAGS] & 4ull) return;
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; b < n_buckets; b += stride) {
    const unsigned m = meta[b];
    const int64_t used = out_off[b] - (int64_t)(m & ~FF_BAD), size = bucket_off[b + 1] - bucket_off[b];
    const int64_t D = T[b + 1] - T[b];
    if (!(m & FF_BAD) && used == T[b]) continue;
    marks[b] = 1;
    // what it wrote: [used, used + size) (a pre-counted bucket: its D pairs); every other bucket whose region that touches
    const int64_t w_lo = used, w_hi = used + (size > FF_CAP ? D : size);
    int64_t lo = 0, hi = n_buckets;                      // first c with T[c + 1] > w_lo
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (T[mid + 1] > w_lo) hi = mid; else lo = mid + 1;
    }
    for (int64_t c = lo; c < n_buckets && T[c] < w_hi; ++c)
      if (T[c + 1] > T[c]) marks[c] = 1;
  }
}

__global__ void finish_collect_kernel(const int64_t* __restrict__ T, const unsigned* __restrict__ marks, int64_t n_buckets,
                                      unsigned long long* __restrict__ header, unsigned* __restrict__ redo_ids,
                                      int64_t* __restrict__ redo_bases) {
  if (header[FS_FLAGS] & 4ull) return;                    // (the fast kernel gave up: nothing here is looked at)
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (b == 0) header[FS_UNIQUE] = (unsigned long long)T[n_buckets];
  for (; b < n_buckets; b += stride) {
    if (marks[b]) {
      const unsigned long long at = atomicAdd(&header[FS_REDO], 1ull);
      redo_ids[at] = (unsigned)b;
      redo_bases[at] = T[b];
    }
  }
}

// What the host decides on before the finishing call, in one reduction over the offsets: out[0] = keys of the largest bucket,
// out[1] = buckets over `cap`, then {bucket, first key, keys} of up to max_list of those IN NO PARTICULAR ORDER (whoever
// finds one takes the next row; the caller sorts the few rows).  (A second kernel that listed them in order with one
// wavefront walking all the offsets cost 5.3 ms per million buckets — on every call, with nothing to list.)
__global__ __launch_bounds__(256) void bucket_census_kernel(const int64_t* __restrict__ bucket_off, int64_t n_buckets, int64_t cap,
                                                            int max_list, unsigned long long* __restrict__ out) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned long long largest = 0;
  for (; b < n_buckets; b += stride) {
    const int64_t lo = bucket_off[b];
    const unsigned long long m = (unsigned long long)(bucket_off[b + 1] - lo);
    largest = max(largest, m);
    if (m > (unsigned long long)cap) {
      const unsigned long long at = atomicAdd(&out[1], 1ull);
      if (at < (unsigned long long)max_list) {
        out[2 + 3 * at] = (unsigned long long)b;
        out[3 + 3 * at] = (unsigned long long)lo;
        out[4 + 3 * at] = m;
      }
    }
  }
  largest = wave_reduce_max(largest);
  if ((threadIdx.x & 63) == 0) atomicMax(&out[0], largest);
}

// How many buckets are too large for the fast kernel but not for the general one (the caller pre-counts only buckets
// over the general kernel's capacity): with any of them the general kernel takes the call.
__global__ void finish_fit_kernel(const int64_t* __restrict__ bucket_off, int64_t n_buckets, unsigned long long* __restrict__ header) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned misfits = 0;
  for (; b < n_buckets; b += stride) {
    const int64_t m = bucket_off[b + 1] - bucket_off[b];
    misfits += (m > FF_CAP && m <= FN_CAP) ? 1u : 0u;
  }
  if (__any(misfits != 0) && (threadIdx.x & 63) == 0) atomicAdd(&header[FS_MISFIT], 1ull);
}

// out_off[b] = bucket_off[b] - (keys - distinct keys) of the pre-counted buckets before b: where bucket b starts in the
// output if no other bucket holds a duplicate.  big_table: {bucket, distinct keys, offset} triples sorted by bucket.
constexpr int FF_MAXBIG = 1024;
__global__ __launch_bounds__(256) void finish_out_offsets_kernel(const int64_t* __restrict__ bucket_off, int64_t n_buckets,
                                                                 const int64_t* __restrict__ big_table, int n_big,
                                                                 int64_t* __restrict__ out_off) {
  __shared__ int64_t ids[FF_MAXBIG], cum[FF_MAXBIG + 1];
  if (threadIdx.x == 0) {
    int64_t run = 0;
    for (int i = 0; i < n_big; ++i) {
      const int64_t c = big_table[3 * i];
      ids[i] = c;
      cum[i] = run;
      run += bucket_off[c + 1] - bucket_off[c] - big_table[3 * i + 1];
    }
    cum[n_big] = run;
  }
  __syncthreads();
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; b <= n_buckets; b += stride) {
    int lo = 0, hi = n_big;                              // number of pre-counted buckets before b
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (ids[mid] < b) lo = mid + 1; else hi = mid;
    }
    out_off[b] = bucket_off[b] - cum[lo];
  }
}


}  // namespace

extern "C" {

int64_t bnpk_finish_capacity(void) { return FN_CAP; }

int bnpk_bucket_census(bnpk_ctx* ctx, const int64_t* d_bucket_offsets, int64_t n_buckets, int64_t cap, int max_list, int64_t* d_out,
                       void* stream) {
  if (!ctx || !d_bucket_offsets || !d_out || n_buckets < 1 || cap < 0 || max_list < 0 || max_list > 4096) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  bnpk_timer t(ctx, "bucket_census", s);
  BNPK_HIP(ctx, hipMemsetAsync(d_out, 0, (size_t)(2 + 3 * max_list) * 8, s));
  hipLaunchKernelGGL(bucket_census_kernel, dim3(grid_for(std::min<int64_t>(ceil_div(n_buckets, 256), 1024))), dim3(256), 0, s,
                     d_bucket_offsets, n_buckets, cap, max_list, reinterpret_cast<unsigned long long*>(d_out));
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

// d_state: the header words; then either the general kernel's 64-bit status words, or the fast path's bookkeeping
// (announcements, distinct counts, per-bucket notes, redo marks); the redo list; the adjusted output offsets.
static int64_t fs_half(int64_t n) { return (n + 1) / 2 + 1; }
int64_t bnpk_finish_state_words(int64_t n_buckets) {
  const int64_t n = std::max<int64_t>(n_buckets, 0);
  return FS_FAST + (n + 1) + 3 * fs_half(n) + n + (n + 1) + 8;
}

int bnpk_finish_sorted(bnpk_ctx* ctx, int64_t* d_part, int64_t n, const int64_t* d_bucket_offsets,
                       int64_t n_buckets, int low_bits, int64_t* d_keys_out, int64_t* d_counts_out, int64_t* d_state,
                       const int64_t* d_big_table, int n_big, const int64_t* d_big_keys, const int64_t* d_big_counts,
                       int64_t* h_n_unique, int* h_overflow, void* stream) {
  return bnpk_finish_sorted_strided(ctx, d_part, n, 0, d_bucket_offsets, n_buckets, low_bits, d_keys_out, d_counts_out, d_state,
                                    d_big_table, n_big, d_big_keys, d_big_counts, h_n_unique, h_overflow, stream);
}

// part_stride != 0: bucket b's keys lie at d_part + b * part_stride (bnpk_radix_partition_claimed + bnpk_claimed_finalize);
// d_bucket_offsets says how many they are and where they would lie in a dense array (the output positions derive from that)
int bnpk_finish_sorted_strided(bnpk_ctx* ctx, int64_t* d_part, int64_t n, int64_t part_stride, const int64_t* d_bucket_offsets,
                               int64_t n_buckets, int low_bits, int64_t* d_keys_out, int64_t* d_counts_out, int64_t* d_state,
                               const int64_t* d_big_table, int n_big, const int64_t* d_big_keys, const int64_t* d_big_counts,
                               int64_t* h_n_unique, int* h_overflow, void* stream) {
  const int64_t pstride = part_stride;
  const int64_t n_slots = pstride ? n_buckets * pstride : n;   // elements of d_part
  if (pstride < 0) return BNPK_ERR_ARG;
  if (!ctx || n < 0 || n_buckets < 1 || low_bits < 0 || low_bits > 63 || !h_n_unique || !h_overflow || !d_state ||
      !d_bucket_offsets || n_big < 0 || (n_big > 0 && (!d_big_table || !d_big_keys || !d_big_counts)))
    return BNPK_ERR_ARG;
  *h_n_unique = 0;
  *h_overflow = 0;
  if (n == 0) return BNPK_OK;
  if (!d_part || !d_keys_out || !d_counts_out || d_keys_out == d_part || d_counts_out == d_part || d_counts_out == d_keys_out)
    return BNPK_ERR_ARG;
  if (n_buckets >= (1ll << 31)) return BNPK_ERR_RANGE;
  hipStream_t s = (hipStream_t)stream;
  if (!ctx->finish_ready) {
    BNPK_HIP(ctx, hipFuncSetAttribute((const void*)finish_sorted_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)FN_LDS));
    BNPK_HIP(ctx, hipFuncSetAttribute((const void*)finish_sorted_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)FN_LDS));
    BNPK_HIP(ctx, hipFuncSetAttribute((const void*)finish_sorted_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)FN_LDS));
    BNPK_HIP(ctx, hipFuncSetAttribute((const void*)finish_fast_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)FF_LDS));
    BNPK_HIP(ctx, hipFuncSetAttribute((const void*)finish_fast_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)FF_LDS));
    int per_cu = 0, per_cu_narrow = 0;
    BNPK_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)finish_fast_kernel<false>, FF_THREADS, FF_LDS));
    BNPK_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_narrow, (const void*)finish_fast_kernel<true>, FF_THREADS, FF_LDS));
    // persistent workgroups, dealt the buckets round robin: as many as are resident at once
    ctx->finish_fast_grid = ctx->compute_units * std::max(1, std::min(per_cu, per_cu_narrow));
    ctx->finish_ready = true;
  }
  unsigned long long* state = reinterpret_cast<unsigned long long*>(d_state);
  int64_t* Dv = d_state + FS_FAST;
  unsigned* meta = reinterpret_cast<unsigned*>(Dv + n_buckets + 1);
  unsigned* marks = meta + 2 * fs_half(n_buckets);
  unsigned* redo_ids = marks + 2 * fs_half(n_buckets);
  int64_t* redo_bases = reinterpret_cast<int64_t*>(redo_ids + 2 * fs_half(n_buckets));
  int64_t* out_off_buf = redo_bases + n_buckets;
  uint64_t* part = reinterpret_cast<uint64_t*>(d_part);
  uint64_t* keys_out = reinterpret_cast<uint64_t*>(d_keys_out);
  const uint64_t* big_keys = reinterpret_cast<const uint64_t*>(d_big_keys);
  int64_t host[6] = {0, 0, 0, 0, 0, 0};
  auto read_header = [&]() -> int {
    BNPK_HIP(ctx, hipMemcpyAsync(host, d_state, sizeof(host), hipMemcpyDeviceToHost, s));
    BNPK_HIP(ctx, hipStreamSynchronize(s));
    return BNPK_OK;
  };
  auto general = [&](bool redo, int64_t n_redo) -> int {
    const int sbits = std::min(low_bits, FN_MAXBITS), sshift = low_bits - sbits;
    // one workgroup per CU fits (LDS); the ticket order keeps the look-back deadlock-free for any grid size
    const unsigned grid = (unsigned)std::min<int64_t>(redo ? n_redo : n_buckets, (int64_t)ctx->compute_units);
    if (redo) {
      BNPK_HIP(ctx, hipMemsetAsync(state + FS_TICKET, 0, 8, s));
      hipLaunchKernelGGL(finish_sorted_kernel<1>, dim3(grid), dim3(FN_THREADS), FN_LDS, s, part, d_bucket_offsets,
                         n_buckets, sshift, sbits, state, keys_out, d_counts_out, d_big_table, n_big, big_keys,
                         d_big_counts, (const unsigned*)redo_ids, (const int64_t*)redo_bases, n_redo, (int64_t*)nullptr, pstride);
    } else {
      BNPK_HIP(ctx, hipMemsetAsync(d_state, 0, (size_t)(FS_BUCKETS + n_buckets + 1) * 8, s));
      hipLaunchKernelGGL(finish_sorted_kernel<0>, dim3(grid), dim3(FN_THREADS), FN_LDS, s, part, d_bucket_offsets,
                         n_buckets, sshift, sbits, state, keys_out, d_counts_out, d_big_table, n_big, big_keys,
                         d_big_counts, (const unsigned*)nullptr, (const int64_t*)nullptr, (int64_t)0, (int64_t*)nullptr, pstride);
    }
    BNPK_HIP(ctx, hipGetLastError());
    return BNPK_OK;
  };
  void* scan_scratch = nullptr;
  unsigned* todo_ids = marks;                            // (the marks are the fast path's)
  // The duplicate-aware path (finish_wave.hip, finish_dup.hip): every bucket's distinct keys back over its own keys, the
  // counts to the same positions of the key array.  A cascade of three kernels, each taking what the one before could not
  // hold, through lists whose lengths are read on the device (no host round trip): one wavefront per bucket with a
  // 704-slot table (only when the probe found the keys duplicate-heavy), one workgroup per bucket with 6144 slots, the
  // general kernel.  Then one scan over the distinct counts and two copies into place.
  auto duplicate_aware = [&](bool wave_first) -> int {
    BNPK_HIP(ctx, hipMemsetAsync(d_state, 0, (size_t)FS_FAST * 8, s));
    if (wave_first) {
      BNPK_CHECK(bnpk_finish_wave_launch(ctx, false, 0, part, n_slots, d_bucket_offsets, n_buckets, low_bits, state, Dv, todo_ids,
                                         d_keys_out, d_big_table, n_big, big_keys, d_big_counts, pstride, s));
    }
    BNPK_CHECK(bnpk_finish_dup_launch(ctx, part, d_bucket_offsets, n_buckets, low_bits, state, Dv, redo_ids, d_keys_out,
                                      d_big_table, n_big, big_keys, d_big_counts, wave_first ? todo_ids : nullptr, pstride, s));
    {
      const int sbits = std::min(low_bits, FN_MAXBITS), sshift = low_bits - sbits;
      const unsigned grid = (unsigned)std::min<int64_t>(n_buckets, (int64_t)ctx->compute_units);
      hipLaunchKernelGGL(finish_sorted_kernel<2>, dim3(grid), dim3(FN_THREADS), FN_LDS, s, (const uint64_t*)part,
                         d_bucket_offsets, n_buckets, sshift, sbits, state, part, d_keys_out, d_big_table, n_big, big_keys,
                         d_big_counts, (const unsigned*)redo_ids, (const int64_t*)nullptr, (int64_t)0, Dv, pstride);
      BNPK_HIP(ctx, hipGetLastError());
    }
    BNPK_CHECK(bnpk_scan_launch(ctx, Dv, n_buckets, 1, Dv, true, (int64_t*)scan_scratch, s));
    BNPK_CHECK(bnpk_finish_compact_launch(ctx, d_keys_out, d_counts_out, d_bucket_offsets, Dv, n_buckets, state, 0, s));
    BNPK_CHECK(bnpk_finish_compact_launch(ctx, d_part, d_keys_out, d_bucket_offsets, Dv, n_buckets, state, pstride, s));
    return BNPK_OK;
  };
  // finish_mode 0: a probe decides — a sample of the buckets goes through the wavefront kernel's table; if (nearly) all of
  // them fit, the keys are duplicate-heavy and the cascade above runs.  Otherwise the fast kernel, which gives up within
  // a bucket per workgroup when keys repeat after all; then the cascade without its first stage.
  // 1 = general kernel only; 2 = fast kernel + redo list, general kernel if it gives up; 3 = the workgroup kernel of the
  // cascade (+ general kernel); 4 = the whole cascade.
  // The multiplicity-counting fast kernel (finish_multi.hip): exact positions, every bucket emitted once; what it leaves
  // (a bin of more than 64 keys) is redone by the general kernel at the position the list carries.
  auto multi = [&]() -> int {
    BNPK_HIP(ctx, hipMemsetAsync(d_state, 0, (size_t)FS_FAST * 8, s));
    BNPK_HIP(ctx, hipMemsetAsync(meta, 0, (size_t)n_buckets * 4, s));
    BNPK_CHECK(bnpk_finish_multi_launch(ctx, part, d_bucket_offsets, n_buckets, low_bits, state, meta, keys_out, d_counts_out,
                                        d_big_table, n_big, big_keys, d_big_counts, redo_ids, redo_bases, pstride, s));
    BNPK_CHECK(read_header());
    if (host[FS_FLAGS] & 2) return BNPK_OK;               // (a wait gave up: the caller takes the general kernel)
    if (host[FS_REDO] > 0 && !(host[FS_FLAGS] & 1)) BNPK_CHECK(general(true, host[FS_REDO]));
    return BNPK_OK;
  };
  const int mode = ctx->finish_mode;
  const bool can_wave = n >= 2;
  // Few buckets (a histogram of ten million keys, not of billions: a chunk's k-mers, a KmerIndex): the general kernel takes
  // them in one launch, ~11 us per bucket and workgroup, and no answer from the device is needed before the end — the probe,
  // the fast kernel's attempt and what follows it are three round trips and a dozen launches, and the kernels built for a
  // million buckets (tickets three iterations ahead, parking rings, look-backs over thousands of status words) idle through
  // most of them: the sacCer3 index spent 3.6 of its 8.4 ms in one finish_multi call over 2048 buckets.
  const bool few = mode == 0 && n_buckets <= (int64_t)16 * ctx->compute_units;
  // A small histogram (up to 2^25 keys) in SMALL buckets (2048 keys on average or fewer: what extra levels over a genome's skewed
  // k-mers leave — sacCer3: 6608 buckets of ~750 keys in the batch of its over-full buckets): one workgroup per bucket sorts it
  // with a bitonic network (finish_small.hip), no probe, no answer from the device before the end, and no dependence on what the
  // keys look like.  The counting-sort kernels take 4.8 ms on 4096 even buckets of the yeast genome's 31-mers (long bins: k-mers
  // that share their next 13 bits) where they take 0.12 ms on random keys, and all of them pay 11-30 us per bucket on 16 K buckets
  // of 740 keys (round 6, scripts/exp/exp_index3.py: general 5.7 ms, fast + redo 5.9, workgroup table 3.8, multiplicities 3.9,
  // cascade 3.4).  The network's cost grows with P log^2 P: at 8192-key buckets it is 6x the general kernel on random keys
  // (0.78 against 0.12 ms per 12 M keys), which is why full-size buckets keep the kernels above.
  const bool small = mode == 0 && n <= (1ll << 25) && n <= 2048 * n_buckets;
  bool use_bitonic = mode == 6 || small;
  bool use_general = mode == 1 || (few && !small), use_dup = mode == 3 || (mode == 4 && !can_wave), use_wave = mode == 4 && can_wave;
  bool try_fast = (mode == 0 || mode == 2) && n_big <= FF_MAXBIG && !few && !small;
  bool use_multi = mode == 5;
  bool nearly_distinct = false;
  // (one arena: the scan partials of the fast / duplicate-aware paths, or the parking ring of the multiplicity kernel —
  // never both at a time; asked for together so that the arena does not move between them)
  if (!use_general)
    BNPK_CHECK(bnpk_scratch(ctx, std::max<size_t>(bnpk_scan_scratch_bytes(n_buckets), (size_t)bnpk_finish_multi_park_bytes(2 * ctx->compute_units)),
                            &scan_scratch, (hipStream_t)stream));
  {
    bnpk_timer t(ctx, "finish_sorted", s);
    if (!few && !small && (mode == 0 || try_fast || use_multi)) {
      bnpk_timer t_probe(ctx, "finish.probe", s);
      BNPK_HIP(ctx, hipMemsetAsync(d_state, 0, (size_t)FS_FAST * 8, s));
      hipLaunchKernelGGL(finish_fit_kernel, dim3(grid_for(std::min<int64_t>(ceil_div(n_buckets, 256), 1024))), dim3(256), 0, s,
                         d_bucket_offsets, n_buckets, state);
      const int64_t probe_buckets = 1024;
      if (mode == 0 && can_wave)
        BNPK_CHECK(bnpk_finish_wave_launch(ctx, true, probe_buckets, part, n_slots, d_bucket_offsets, n_buckets, low_bits, state, Dv,
                                           todo_ids, d_keys_out, d_big_table, n_big, big_keys, d_big_counts, pstride, s));
      int64_t probe[4] = {0, 0, 0, 0};
      BNPK_HIP(ctx, hipMemcpyAsync(probe, d_state + FS_PROBE_BAD, sizeof(probe), hipMemcpyDeviceToHost, s));
      BNPK_CHECK(read_header());
      if (host[FS_MISFIT] != 0) { try_fast = false; if (use_multi) { use_multi = false; use_general = true; } }
      if (mode == 0 && can_wave && probe[2] > 0) {
        const int64_t stride = std::max<int64_t>(1, n_buckets / probe_buckets), sampled = ceil_div(n_buckets, stride);
        if (getenv("BNPK_FINISH_DEBUG"))
          fprintf(stderr, "bnpk finish probe: sampled %lld bad %lld distinct(good) %lld keys %lld shown(bad) %lld\n", (long long)sampled,
                  (long long)probe[0], (long long)probe[1], (long long)probe[2], (long long)probe[3]);
        // (nearly) all sampled buckets fit the wavefront's table: duplicate-heavy keys.  The wavefront kernel pays while a
        // bucket's distinct keys fill a small part of its 704 slots; above ~150 of them the probe sequences grow and the
        // workgroup table alone is faster (round 6, reads of a genome, 5.7 K keys per bucket, scripts/exp/exp_finish_rules.sh:
        // 20x coverage = 358 distinct per bucket: cascade 72-83 ms, workgroup table 22; 40x = 180: 22-25 / 17.6; 60x = 120:
        // 12.7-14.5 / 16.3; 100x = 72: 10.3-11.8 / 15.3)
        if (probe[0] * 16 <= sampled) {
          try_fast = false;
          if (probe[1] <= 150 * (sampled - probe[0])) use_wave = true; else use_dup = true;
        }
        // (only if the whole-bucket probe below finds no bucket to sort: the sampled buckets overflowed the table's list of 448
        // distinct keys, and how many keys they had shown by then — 559 on average when all are distinct, 710 / 728 on reads at 4x /
        // 5x coverage — says how often keys repeat, as long as the copies of a key arrive spread out)
        nearly_distinct = probe[0] > 0 && probe[3] < 720 * probe[0];
      }
    }
    if (try_fast) {
      bnpk_timer t_fast(ctx, "finish.fast", s);
      const int sbits = std::min(low_bits, FF_MAXBITS), sshift = low_bits - sbits;
      BNPK_HIP(ctx, hipMemsetAsync(d_state, 0, (size_t)FS_FAST * 8, s));
      BNPK_HIP(ctx, hipMemsetAsync(marks, 0, (size_t)n_buckets * 4, s));
      const int64_t* out_off = d_bucket_offsets;
      if (n_big > 0) {
        hipLaunchKernelGGL(finish_out_offsets_kernel, dim3(grid_for(ceil_div(n_buckets + 1, 256))), dim3(256), 0, s,
                           d_bucket_offsets, n_buckets, d_big_table, n_big, out_off_buf);
        out_off = out_off_buf;
      }
      const unsigned grid = (unsigned)std::min<int64_t>(n_buckets, (int64_t)ctx->finish_fast_grid);
      // keys of one bin differ only below bit sshift: 32-bit compares when that is all inside the low word
      if (sshift <= 32)
        hipLaunchKernelGGL(finish_fast_kernel<true>, dim3(grid), dim3(FF_THREADS), FF_LDS, s, (const uint64_t*)part, d_bucket_offsets,
                           out_off, n_buckets, sshift, sbits, state, Dv, meta, keys_out, d_counts_out, d_big_table,
                           n_big, big_keys, d_big_counts, pstride);
      else
        hipLaunchKernelGGL(finish_fast_kernel<false>, dim3(grid), dim3(FF_THREADS), FF_LDS, s, (const uint64_t*)part, d_bucket_offsets,
                           out_off, n_buckets, sshift, sbits, state, Dv, meta, keys_out, d_counts_out, d_big_table,
                           n_big, big_keys, d_big_counts, pstride);
      BNPK_HIP(ctx, hipGetLastError());
      BNPK_CHECK(bnpk_scan_launch(ctx, Dv, n_buckets, 1, Dv, true, (int64_t*)scan_scratch, s));
      const unsigned cgrid = grid_for(std::min<int64_t>(ceil_div(n_buckets, 256), 2048));
      hipLaunchKernelGGL(finish_check_kernel, dim3(cgrid), dim3(256), 0, s, (const int64_t*)Dv, (const unsigned*)meta,
                         d_bucket_offsets, out_off, n_buckets, marks, (const unsigned long long*)state);
      hipLaunchKernelGGL(finish_collect_kernel, dim3(cgrid), dim3(256), 0, s, (const int64_t*)Dv, (const unsigned*)marks,
                         n_buckets, state, redo_ids, redo_bases);
      BNPK_HIP(ctx, hipGetLastError());
      BNPK_CHECK(read_header());
      if (host[FS_FLAGS] & 4) try_fast = false;           // duplicate-heavy keys: everything again, with another kernel
      else if (host[FS_REDO] > 0 && !(host[FS_FLAGS] & 1)) BNPK_CHECK(general(true, host[FS_REDO]));
    }
    if (!try_fast && !use_general && !use_dup && !use_wave && !use_multi && !use_bitonic) {
      // the fast kernel refused (repeats in more than FF_LOG buckets) or could not be tried
      if (mode == 2) use_general = true;
      else if (mode == 0 && host[FS_MISFIT] == 0) {
        // Keys that repeat in full-size buckets: the multiplicity kernel (its time grows with the longest run of equal keys)
        // or the workgroup table (its time grows with the bucket's distinct keys: the table's load).  On reads of a genome
        // they cross where a bucket holds ~1450 distinct keys — 4x-5x coverage at 5.7 K keys per bucket; DESIGN 4c has the
        // scan — and (k-mer, row) words of which a third occur twice (2500 distinct of 3800) belong to the multiplicity kernel
        // at half the table's time.  What the wavefront probe saw (the first few hundred keys of a bucket) says how many
        // distinct keys the whole bucket holds only if the copies of a key arrive spread out, so 256 buckets are sorted
        // whole (finish_small.hip in probe mode: ~50 us, one more answer from the device; not on the headline's path).
        bnpk_timer t_probe2(ctx, "finish.probe", s);
        BNPK_HIP(ctx, hipMemsetAsync(d_state, 0, (size_t)FS_FAST * 8, s));
        BNPK_CHECK(bnpk_finish_bitonic_probe_launch(ctx, part, d_bucket_offsets, n_buckets, 256, state, pstride, s));
        int64_t whole[3] = {0, 0, 0};                     // buckets sorted, their distinct keys, their keys
        BNPK_HIP(ctx, hipMemcpyAsync(whole, d_state + FS_PROBE_BAD, sizeof(whole), hipMemcpyDeviceToHost, s));
        BNPK_HIP(ctx, hipStreamSynchronize(s));
        if (getenv("BNPK_FINISH_DEBUG"))
          fprintf(stderr, "bnpk finish probe (whole buckets): %lld sorted, %lld distinct of %lld keys\n", (long long)whole[0],
                  (long long)whole[1], (long long)whole[2]);
        if (whole[0] > 0 ? whole[1] > 1450 * whole[0] : nearly_distinct) use_multi = true; else use_dup = true;
      }
      else use_dup = true;
    }
    if (use_multi) {
      bnpk_timer t_multi(ctx, "finish.multi", s);
      BNPK_CHECK(multi());
      if (host[FS_FLAGS] & 2) use_general = true;
    }
    if (use_bitonic) {
      bnpk_timer t_small(ctx, "finish.bitonic", s);
      BNPK_HIP(ctx, hipMemsetAsync(d_state, 0, (size_t)FS_FAST * 8, s));
      BNPK_CHECK(bnpk_finish_bitonic_launch(ctx, part, d_bucket_offsets, n_buckets, state, Dv, d_keys_out, d_big_table, n_big, big_keys,
                                            d_big_counts, pstride, s));
      BNPK_CHECK(bnpk_scan_launch(ctx, Dv, n_buckets, 1, Dv, true, (int64_t*)scan_scratch, s));
      BNPK_CHECK(bnpk_finish_compact_launch(ctx, d_keys_out, d_counts_out, d_bucket_offsets, Dv, n_buckets, state, 0, s));
      BNPK_CHECK(bnpk_finish_compact_launch(ctx, d_part, d_keys_out, d_bucket_offsets, Dv, n_buckets, state, pstride, s));
    }
    if (use_wave) { bnpk_timer t_wave(ctx, "finish.cascade", s); BNPK_CHECK(duplicate_aware(true)); }
    else if (use_dup) { bnpk_timer t_dup(ctx, "finish.dup", s); BNPK_CHECK(duplicate_aware(false)); }
    if (use_general) { bnpk_timer t_gen(ctx, "finish.general", s); BNPK_CHECK(general(false, 0)); }
  }
  BNPK_CHECK(read_header());
  *h_overflow = (int)(host[FS_FLAGS] & 3);              // 1: a bucket over the capacity without a pre-counted entry; 2: a wait gave up
  *h_n_unique = host[FS_UNIQUE];
  return BNPK_OK;
}

}  // extern "C"
