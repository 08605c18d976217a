// Finishing kernel of the sparse k-mer histogram for SMALL histograms (round 6): a comparison sort, because everything else
// here degrades on the k-mers of a real genome.  np.unique(return_counts=True) semantics per bucket
// (bionumpy/sequence/count_encoded.py:150-188 extended to k > 8).
//
// The other finishing kernels rank a bucket's keys by a counting sort over the next 13 bits and then inside the bins, or through
// hash tables of a few thousand slots.  On 6e9 random 31-mers that is the right thing (bins of 1.4 keys); on the 12 M 31-mers of a
// yeast genome thousands of keys of a bucket share those 13 bits (low-complexity sequence), the bins are long and the ranking
// inside them is quadratic: the general kernel takes 4.8 ms on 4096 even buckets of sacCer3 where it takes 0.12 ms on random
// keys, and every kernel pays 11-30 us of set-up per bucket on buckets a tenth of the size it was built for (NOTES round 6).
// A histogram of up to 2^25 keys has at most a few thousand buckets; one workgroup sorts a bucket in LDS with a bitonic network
// (up to 8192 keys: 91 compare-exchange steps, ~40 us, whatever the keys look like), marks the first occurrences, and writes the
// distinct keys back over the bucket and their counts beside them — the duplicate-aware kernels' convention (finish.h), so the
// scan of the distinct counts and the two compacting copies that follow are theirs.
#include <algorithm>

#include "finish.h"

namespace {

constexpr int FB_THREADS = 512;
constexpr int FB_CAP = FINISH_CAP;                           // 8192 keys
constexpr size_t FB_OFF_HEADS = (size_t)FB_CAP * 8;
constexpr size_t FB_LDS = FB_OFF_HEADS + (size_t)FB_CAP * 2;      // 80 KB exactly: the stage and one 16-bit position per distinct key
constexpr size_t FB_OFF_WSUM = FB_LDS - (FB_THREADS / 64) * 4;    // (the wave totals of the scan borrow the last head slots: see below)
static_assert(2 * FB_LDS <= 160 * 1024, "two workgroups per CU");

__device__ __forceinline__ int64_t fb_uniform(int64_t v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uint64_t)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

// PROBE (mode 0 of bnpk_finish_sorted, full-size buckets whose keys repeat): every probe_stride-th bucket is sorted and NOTHING is
// written but header[FS_PROBE_BAD] += 1, [FS_PROBE_DISTINCT] += its distinct keys, [FS_PROBE_KEYS] += its keys
// — the exact number the choice between the multiplicity kernel and the workgroup table depends on (the wavefront probe sees the first few hundred keys of a bucket: what they say about the rest
// depends on the order the keys arrived in).  Buckets over the capacity are skipped.
template <bool PROBE>
__global__ __launch_bounds__(FB_THREADS) void finish_bitonic_kernel(
    uint64_t* __restrict__ A, const int64_t* __restrict__ bucket_off, int64_t n_buckets, unsigned long long* __restrict__ header,
    int64_t* __restrict__ Dv, int64_t* __restrict__ loose_counts, const int64_t* __restrict__ big_table, int n_big,
    const uint64_t* __restrict__ big_keys, const int64_t* __restrict__ big_counts, int64_t pstride, int64_t probe_stride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* stage = reinterpret_cast<uint64_t*>(smem);
  unsigned short* heads = reinterpret_cast<unsigned short*>(smem + FB_OFF_HEADS);
  unsigned* wsum = reinterpret_cast<unsigned*>(smem + FB_OFF_WSUM);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Buckets are handed out by a ticket counter (header[FS_TICKET], zeroed by the launcher), not by blockIdx: a genome's dense
  // buckets have regular ids (the k-mers that end in AAAA...), a fixed stride would give a few workgroups all of them (sacCer3 at
  // 14 bits: the busiest of 512 workgroups would sort for 3.5 ms, the average one for 0.64).
  unsigned* ticket = reinterpret_cast<unsigned*>(heads);   // (the head positions are not in use between two buckets)
  while (true) {
    __syncthreads();                                         // everybody is done with the previous bucket — and with its ticket:
    if (tid == 0) ticket[0] = (unsigned)atomicAdd(&header[FS_TICKET], 1ull);   // (a branch without a barrier of its own would otherwise
    __syncthreads();                                         //  race with this write: a wavefront still looking at the old ticket)
    const int64_t b = (int64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)ticket[0]) * (PROBE ? probe_stride : 1);
    if (b >= n_buckets) break;
    const int64_t lo = fb_uniform(bucket_off[b]);
    const int64_t size = fb_uniform(bucket_off[b + 1]) - lo;
    const int64_t src = pstride ? b * pstride : lo;
    if (PROBE && (size < 2 || size > FB_CAP)) continue;
    if (size == 0) {
      if (tid == 0) Dv[b] = 0;
      continue;
    }
    if (size > FB_CAP) {                                     // a bucket the caller counted beforehand (sorted by bucket id), or a mistake
      int lo_i = 0, hi_i = n_big;
      while (lo_i < hi_i) {
        const int mid = (lo_i + hi_i) >> 1;
        if (big_table[3 * mid] < b) lo_i = mid + 1; else hi_i = mid;
      }
      unsigned D = 0;
      if (lo_i < n_big && big_table[3 * lo_i] == b) {
        D = (unsigned)fb_uniform(big_table[3 * lo_i + 1]);
        const int64_t from = fb_uniform(big_table[3 * lo_i + 2]);
        for (unsigned i = tid; i < D; i += FB_THREADS) {
          A[src + i] = big_keys[from + i];
          loose_counts[lo + i] = big_counts[from + i];
        }
      } else if (tid == 0) {
        atomicOr(&header[FS_FLAGS], 1ull);
      }
      if (tid == 0) Dv[b] = D;
      continue;
    }
    const int nb = (int)size;
    int P = 64;
    while (P < nb) P <<= 1;
    for (int i = tid; i < P; i += FB_THREADS) stage[i] = i < nb ? A[src + i] : ~0ull;     // (keys are < 2^63: the pad sorts last)
    __syncthreads();
    // bitonic network, ascending
    for (int k = 2; k <= P; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int t = tid; t < (P >> 1); t += FB_THREADS) {
          const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
          const int p = i | j;
          const uint64_t x = stage[i], y = stage[p];
          const bool up = (i & k) == 0;
          if ((x > y) == up) {
            stage[i] = y;
            stage[p] = x;
          }
        }
        __syncthreads();
      }
    }
    // first occurrences: a lane looks at a contiguous stretch, so that the distinct keys leave in order
    const int per = (P + FB_THREADS - 1) / FB_THREADS;
    const int first = tid * per;
    unsigned mine = 0;
    for (int u = 0; u < per; ++u) {
      const int i = first + u;
      if (i < nb && (i == 0 || stage[i] != stage[i - 1])) ++mine;
    }
    const unsigned inc = wave_inclusive_scan(mine);
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned base = inc - mine, D = 0;
#pragma unroll
    for (int w = 0; w < FB_THREADS / 64; ++w) {
      const unsigned s = wsum[w];
      if (w < wave) base += s;
      D += s;
    }
    __syncthreads();                                         // (the totals lie where the last head positions go)
    if (PROBE) {
      if (tid == 0) {
        atomicAdd(&header[FS_PROBE_BAD], 1ull);
        atomicAdd(&header[FS_PROBE_DISTINCT], (unsigned long long)D);
        atomicAdd(&header[FS_PROBE_KEYS], (unsigned long long)nb);
      }
      continue;
    }
    uint64_t* ko = A + src;
    for (int u = 0; u < per; ++u) {
      const int i = first + u;
      if (i < nb && (i == 0 || stage[i] != stage[i - 1])) {
        ko[base] = stage[i];
        heads[base] = (unsigned short)i;
        ++base;
      }
    }
    if (tid == 0) Dv[b] = D;
    __syncthreads();
    int64_t* co = loose_counts + lo;
    for (unsigned r = tid; r < D; r += FB_THREADS) co[r] = (int64_t)(r + 1 < D ? (int)heads[r + 1] : nb) - (int64_t)heads[r];
  }                                                          // (the barrier at the top hands the stage and the heads to the next bucket)
}

}  // namespace

int bnpk_finish_bitonic_launch(bnpk_ctx* ctx, uint64_t* part, const int64_t* bucket_off, int64_t n_buckets, unsigned long long* header,
                               int64_t* Dv, int64_t* loose_counts, const int64_t* big_table, int n_big, const uint64_t* big_keys,
                               const int64_t* big_counts, int64_t pstride, hipStream_t s) {
  if (!ctx->finish_small_ready) {
    BNPK_HIP(ctx, hipFuncSetAttribute((const void*)finish_bitonic_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FB_LDS));
    BNPK_HIP(ctx, hipFuncSetAttribute((const void*)finish_bitonic_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FB_LDS));
    ctx->finish_small_ready = true;
  }
  const unsigned grid = (unsigned)std::min<int64_t>(n_buckets, (int64_t)ctx->compute_units * 2);
  hipLaunchKernelGGL(finish_bitonic_kernel<false>, dim3(grid), dim3(FB_THREADS), FB_LDS, s, part, bucket_off, n_buckets, header, Dv, loose_counts,
                     big_table, n_big, big_keys, big_counts, pstride, (int64_t)1);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

// ~`probe_buckets` evenly spaced buckets sorted, nothing written but the header's probe words (zeroed by the caller, the ticket too)
int bnpk_finish_bitonic_probe_launch(bnpk_ctx* ctx, const uint64_t* part, const int64_t* bucket_off, int64_t n_buckets,
                                     int64_t probe_buckets, unsigned long long* header, int64_t pstride, hipStream_t s) {
  if (!ctx->finish_small_ready) {
    BNPK_HIP(ctx, hipFuncSetAttribute((const void*)finish_bitonic_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FB_LDS));
    BNPK_HIP(ctx, hipFuncSetAttribute((const void*)finish_bitonic_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FB_LDS));
    ctx->finish_small_ready = true;
  }
  const int64_t stride = std::max<int64_t>(1, n_buckets / probe_buckets), sampled = ceil_div(n_buckets, stride);
  const unsigned grid = (unsigned)std::min<int64_t>(sampled, (int64_t)ctx->compute_units * 2);
  hipLaunchKernelGGL(finish_bitonic_kernel<true>, dim3(grid), dim3(FB_THREADS), FB_LDS, s, const_cast<uint64_t*>(part), bucket_off, n_buckets, header,
                     (int64_t*)nullptr, (int64_t*)nullptr, (const int64_t*)nullptr, 0, (const uint64_t*)nullptr, (const int64_t*)nullptr, pstride,
                     stride);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}
