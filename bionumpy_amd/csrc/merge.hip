// A10 for sparse histograms: SparseKmerCounts.__add__ == EncodedCounts.__add__ (bionumpy/sequence/count_encoded.py:38-48)
// for (sorted distinct keys, counts) pairs — the sum that `streamable(sum)` folds over the chunks of a file
// (bionumpy/streams/..., sequence/kmers.py:129).  Two sorted lists of distinct keys are merged along the merge path:
// the merged sequence is cut into tiles of MG_TILE elements by one binary search per tile (diagonals of the merge
// matrix), a workgroup merges its tile from LDS, and because the keys of either list are distinct an output entry is
//   every element of A, with B's count added when B holds the same key (it is the very next element of the merge), and
//   every element of B whose key A does not hold (the element before it in the merge is not equal to it).
// Two passes over the tiles: count the outputs per tile, scan, write.  Nothing is sorted, nothing re-sorted: adding
// the histogram of a chunk to the running total reads both once and writes the sum once.
#include <algorithm>

#include "common.h"
#include "scan.h"

namespace {

constexpr int MG_THREADS = 256;
constexpr int MG_ITEMS = 8;
constexpr int MG_TILE = MG_THREADS * MG_ITEMS;

// number of elements of a among the first d elements of merge(a, b) (ties: a first)
template <typename GetA, typename GetB>
__device__ __forceinline__ int64_t merge_path(int64_t d, int64_t na, int64_t nb, GetA a, GetB b) {
  int64_t lo = d > nb ? d - nb : 0, hi = d < na ? d : na;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (a(mid) <= b(d - 1 - mid)) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__global__ void mg_split_kernel(const int64_t* __restrict__ a, int64_t na, const int64_t* __restrict__ b, int64_t nb,
                                int64_t n_tiles, int64_t* __restrict__ split_a) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; t <= n_tiles; t += stride) {
    const int64_t d = min(t * MG_TILE, na + nb);
    split_a[t] = merge_path(d, na, nb, [&](int64_t i) { return a[i]; }, [&](int64_t j) { return b[j]; });
  }
}

template <bool WRITE>
__global__ __launch_bounds__(MG_THREADS) void mg_tile_kernel(const int64_t* __restrict__ a, const int64_t* __restrict__ ca, int64_t na,
                                                             const int64_t* __restrict__ b, const int64_t* __restrict__ cb, int64_t nb,
                                                             const int64_t* __restrict__ split_a, int64_t* __restrict__ tile_out,
                                                             int64_t* __restrict__ out_keys, int64_t* __restrict__ out_counts) {
  __shared__ int64_t keys[MG_TILE];
  __shared__ int scan_smem[MG_THREADS / 64 + 1];
  const int64_t t = blockIdx.x;
  const int64_t a0 = split_a[t], a1 = split_a[t + 1];
  const int64_t d0 = t * MG_TILE, d1 = min(d0 + MG_TILE, na + nb);
  const int64_t b0 = d0 - a0, b1 = d1 - a1;
  const int la = (int)(a1 - a0), lb = (int)(b1 - b0);
  for (int i = threadIdx.x; i < la; i += MG_THREADS) keys[i] = a[a0 + i];
  for (int i = threadIdx.x; i < lb; i += MG_THREADS) keys[la + i] = b[b0 + i];
  __syncthreads();
  // this thread's MG_ITEMS consecutive elements of the tile's merge
  const int d = min((int)threadIdx.x * MG_ITEMS, la + lb);
  int ai = (int)merge_path(d, la, lb, [&](int64_t i) { return keys[i]; }, [&](int64_t j) { return keys[la + j]; });
  int bi = d - ai;
  // the key of b right behind the cursor may lie outside the tile
  auto key_b = [&](int j) -> int64_t { return j < lb ? keys[la + j] : (b0 + j < nb ? b[b0 + j] : INT64_MAX); };
  int64_t ok[MG_ITEMS], oc[MG_ITEMS];
  unsigned entries = 0;                                    // bit q: element q of this thread is an output entry
  int n_out = 0;
#pragma unroll
  for (int q = 0; q < MG_ITEMS; ++q) {
    ok[q] = 0;
    oc[q] = 0;
    if (ai + bi < la + lb) {
      const bool take_a = bi >= lb || (ai < la && keys[ai] <= keys[la + bi]);
      if (take_a) {                                        // every element of a is an entry; b may hold the same key next
        const int64_t key = keys[ai];
        ok[q] = key;
        if (WRITE) oc[q] = ca[a0 + ai] + (key_b(bi) == key ? cb[b0 + bi] : 0);
        entries |= 1u << q;
        ++ai;
      } else {                                             // an element of b is an entry unless a held its key just before
        const int64_t key = keys[la + bi];
        const int64_t prev = ai > 0 ? keys[ai - 1] : (a0 > 0 ? a[a0 - 1] : INT64_MIN);
        if (prev != key) {
          ok[q] = key;
          if (WRITE) oc[q] = cb[b0 + bi];
          entries |= 1u << q;
        }
        ++bi;
      }
    }
  }
  n_out = __popc(entries);
  int total;
  const int before = block_exclusive_scan(n_out, scan_smem, &total);
  if (!WRITE) {
    if (threadIdx.x == 0) tile_out[t] = total;
    return;
  }
  int64_t at = tile_out[t] + before;
#pragma unroll
  for (int q = 0; q < MG_ITEMS; ++q) {
    if ((entries >> q) & 1u) {
      out_keys[at] = ok[q];
      out_counts[at] = oc[q];
      ++at;
    }
  }
}

}  // namespace

extern "C" {

int bnpk_merge_add(bnpk_ctx* ctx, const int64_t* d_a_keys, const int64_t* d_a_counts, int64_t na, const int64_t* d_b_keys,
                   const int64_t* d_b_counts, int64_t nb, int64_t* d_out_keys, int64_t* d_out_counts, int64_t* h_n_out,
                   void* stream) {
  if (!ctx || na < 0 || nb < 0 || !h_n_out) return BNPK_ERR_ARG;
  *h_n_out = 0;
  const int64_t n = na + nb;
  if (n == 0) return BNPK_OK;
  if ((na > 0 && (!d_a_keys || !d_a_counts)) || (nb > 0 && (!d_b_keys || !d_b_counts)) || !d_out_keys || !d_out_counts)
    return BNPK_ERR_ARG;
  const int64_t n_tiles = ceil_div(n, MG_TILE);
  if (n_tiles > BNPK_MAX_BLOCKS) return BNPK_ERR_RANGE;
  hipStream_t s = (hipStream_t)stream;
  void* scratch = nullptr;
  const size_t words = (size_t)(n_tiles + 1) * 2;
  BNPK_CHECK(bnpk_scratch(ctx, words * 8 + bnpk_scan_scratch_bytes(n_tiles) + 64, &scratch, s));
  int64_t* split_a = (int64_t*)scratch;
  int64_t* tile_out = split_a + (n_tiles + 1);
  int64_t* partials = tile_out + (n_tiles + 1);
  {
    bnpk_timer t(ctx, "merge_add", s);
    hipLaunchKernelGGL(mg_split_kernel, dim3(grid_for(std::min<int64_t>(ceil_div(n_tiles + 1, 256), 4096))), dim3(256), 0, s,
                       d_a_keys, na, d_b_keys, nb, n_tiles, split_a);
    hipLaunchKernelGGL((mg_tile_kernel<false>), dim3((unsigned)n_tiles), dim3(MG_THREADS), 0, s, d_a_keys, d_a_counts, na,
                       d_b_keys, d_b_counts, nb, (const int64_t*)split_a, tile_out, d_out_keys, d_out_counts);
    BNPK_CHECK(bnpk_scan_launch(ctx, tile_out, n_tiles, 1, tile_out, true, partials, s));
    hipLaunchKernelGGL((mg_tile_kernel<true>), dim3((unsigned)n_tiles), dim3(MG_THREADS), 0, s, d_a_keys, d_a_counts, na,
                       d_b_keys, d_b_counts, nb, (const int64_t*)split_a, tile_out, d_out_keys, d_out_counts);
    BNPK_HIP(ctx, hipGetLastError());
  }
  BNPK_HIP(ctx, hipMemcpyAsync(h_n_out, tile_out + n_tiles, 8, hipMemcpyDeviceToHost, s));
  BNPK_HIP(ctx, hipStreamSynchronize(s));
  return BNPK_OK;
}

}  // extern "C"
