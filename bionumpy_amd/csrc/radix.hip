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
#include "rows.h"
#include "scan.h"

namespace {

constexpr int RP_THREADS = 1024;
constexpr int RP_MAXB = 1024;                         // buckets per level
constexpr int RP_STAGE = 16384;                       // keys staged in LDS (128 KiB)
constexpr int RP_TILE = 8192;                         // new keys per round (at most)
constexpr int RP_MAXITEMS = RP_TILE / RP_THREADS;     // 8
constexpr int RP_GRAN = 1024;                         // granularity of the tile -> row table of the fused source
constexpr uint64_t RP_PHANTOM = 1ull << 63;           // placeholder for the slots before a bucket's first key

// LDS carve-up of the partition kernels (dynamic, 16-byte aligned pieces)
constexpr size_t RP_OFF_META = (size_t)RP_STAGE * 8;
constexpr size_t RP_OFF_CNT = RP_OFF_META + (size_t)RP_MAXB * 8;
constexpr size_t RP_OFF_LINE = RP_OFF_CNT + (size_t)RP_MAXB * 4;
constexpr size_t RP_OFF_WSUM = RP_OFF_LINE + (size_t)RP_MAXB * 4;
constexpr size_t RP_OFF_SLAB = RP_OFF_WSUM + 32 * 4;
constexpr size_t RP_LDS = RP_OFF_SLAB + 8 * 8;
constexpr size_t RP_HIST_LDS = (size_t)RP_MAXB * 4 + 8 * 8;

struct slab_t {
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
  sl.local = (int64_t)blockIdx.x - sh[1];
  sl.nsl = sh[2];
  sl.hbase = sh[1] * B;
  sl.lo = sh[3] + sl.local * slab_keys;
  sl.hi = min(sl.lo + slab_keys, sh[4]);
  return true;
}

// ---- key sources ---------------------------------------------------------------------------------------------
// load(): up to `items` keys of the tile [t0, t0 + items*RP_THREADS) ∩ [.., hi) for this lane; returns how many.
struct mem_source {
  const uint64_t* __restrict__ keys;
  __device__ __forceinline__ int load(int64_t t0, int64_t hi, int items, uint64_t k[RP_MAXITEMS]) const {
    int cnt = 0;
#pragma unroll
    for (int q = 0; q < RP_MAXITEMS; ++q) {
      int64_t i = t0 + threadIdx.x + (int64_t)q * RP_THREADS;
      if (q < items && i < hi) { k[q] = keys[i]; ++cnt; }
    }
    return cnt;
  }
};

// the k-mer hashes of the ragged read set, generated on the fly from the packed 2-bit reads (A8); key index ==
// flat output index of bnpk_kmers
struct kmer_source {
  const uint64_t* __restrict__ W;
  const int64_t* __restrict__ in_off;
  const int64_t* __restrict__ out_off;
  const int64_t* __restrict__ tile_rows;   // row containing output t*RP_GRAN
  int64_t n_rows, n_tiles;
  uint64_t mask;
  __device__ __forceinline__ int load(int64_t t0, int64_t hi, int items, uint64_t k[RP_MAXITEMS]) const {
    const int64_t o = t0 + (int64_t)threadIdx.x * items;
    if (o >= hi) return 0;
    const int64_t t_first = t0 / RP_GRAN, t_last = (min(t0 + (int64_t)items * RP_THREADS, hi) - 1) / RP_GRAN;
    const int64_t rlo = tile_rows[t_first];
    const int64_t rhi = (t_last + 1 < n_tiles) ? tile_rows[t_last + 1] : n_rows - 1;
    row_cursor c = seek_row(in_off, out_off, rlo, rhi, o);
    word_window ww;
    int cnt = 0;
#pragma unroll
    for (int q = 0; q < RP_MAXITEMS; ++q) {
      const int64_t oo = o + q;
      if (q >= items || oo >= hi) break;
      if (q) next_output(c, in_off, out_off, oo);
      k[q] = bits_at(W, c.in_pos, ww) & mask;
      ++cnt;
    }
    return cnt;
  }
};

// ---- pass 1 of a level: digit counts per slab ------------------------------------------------------------------
template <typename Source>
__global__ __launch_bounds__(RP_THREADS) void rp_hist_kernel(Source src, const int64_t* __restrict__ seg_off,
                                                             const int64_t* __restrict__ seg_slabs, int64_t n_seg,
                                                             int64_t slab_keys, int shift, int bits,
                                                             int64_t* __restrict__ H) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned* h = reinterpret_cast<unsigned*>(smem);
  int64_t* sh = reinterpret_cast<int64_t*>(smem + (size_t)RP_MAXB * 4);
  const int B = 1 << bits;
  slab_t sl;
  if (!find_slab(seg_off, seg_slabs, n_seg, slab_keys, B, sh, sl)) return;
  if (threadIdx.x < B) h[threadIdx.x] = 0;
  __syncthreads();
  for (int64_t t0 = sl.lo; t0 < sl.hi; t0 += RP_TILE) {
    uint64_t k[RP_MAXITEMS];
    const int cnt = src.load(t0, sl.hi, RP_MAXITEMS, k);
#pragma unroll
    for (int q = 0; q < RP_MAXITEMS; ++q)
      if (q < cnt) atomicAdd(&h[(unsigned)(k[q] >> shift) & (B - 1)], 1u);
  }
  __syncthreads();
  if (threadIdx.x < B) H[sl.hbase + (int64_t)threadIdx.x * sl.nsl + sl.local] = h[threadIdx.x];
}

// ---- pass 2 of a level: write-combining scatter ------------------------------------------------------------------
// One round = one tile of new keys merged with the keys carried over from the previous round:
//   rank   every new key takes a rank inside its bucket from an LDS counter (done right after the tile is loaded,
//          i.e. at the end of the previous round, so the loads / the k-mer generation overlap the store drain)
//   layout per bucket: nfl = keys that complete whole 128-byte lines, the rest is carried; one packed scan gives
//          every bucket a slice of the FLUSH region (a multiple of 16 keys, 128-byte aligned in LDS) and a slice of
//          the CARRY region behind it
//   stage  carried + new keys are written to their slices
//   flush  the FLUSH region leaves the CU as aligned 16-byte-per-lane stores (whole lines only); the CARRY region is
//          read back into the owning lanes' registers
__device__ __forceinline__ unsigned rp_tile_size(unsigned carried) {
  return min((unsigned)RP_TILE, (RP_STAGE - carried) & ~(unsigned)(RP_THREADS - 1));     // >= RP_THREADS
}

template <typename Source>
__global__ __launch_bounds__(RP_THREADS) void rp_scatter_kernel(Source src, const int64_t* __restrict__ seg_off,
                                                                const int64_t* __restrict__ seg_slabs, int64_t n_seg,
                                                                int64_t slab_keys, int shift, int bits,
                                                                const int64_t* __restrict__ offs,
                                                                uint64_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* stage = reinterpret_cast<uint64_t*>(smem);
  uint64_t* meta = reinterpret_cast<uint64_t*>(smem + RP_OFF_META);   // {flush start:16 | nfl:16 | carry start:16 | rem:16}
  unsigned* newcnt = reinterpret_cast<unsigned*>(smem + RP_OFF_CNT);
  unsigned* line = reinterpret_cast<unsigned*>(smem + RP_OFF_LINE);   // write cursor of the bucket / 16
  unsigned* wsum = reinterpret_cast<unsigned*>(smem + RP_OFF_WSUM);
  int64_t* sh = reinterpret_cast<int64_t*>(smem + RP_OFF_SLAB);
  const int B = 1 << bits;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  slab_t sl;
  if (!find_slab(seg_off, seg_slabs, n_seg, slab_keys, B, sh, sl)) return;
  if (sl.lo >= sl.hi) return;

  // lane t owns bucket t: its write cursor (kept 16-key aligned; the slots between the aligned cursor and the
  // bucket's true first position are phantom keys that are staged like real ones but never stored), the
  // number of carried keys and those keys themselves.
  int64_t cursor = 0;
  unsigned rem = 0;
  uint64_t left[15];
  if (tid < B) {
    const int64_t c0 = offs[sl.hbase + (int64_t)tid * sl.nsl + sl.local];
    cursor = c0 & ~15ll;
    rem = (unsigned)(c0 & 15);
    newcnt[tid] = 0;
  }
  const uint64_t phantom = RP_PHANTOM | ((uint64_t)tid << shift);
#pragma unroll
  for (int j = 0; j < 15; ++j) left[j] = phantom;
  {
    unsigned s = wave_reduce_sum(rem);
    if (lane == 0) wsum[wave] = s;
  }
  __syncthreads();
  unsigned carried = 0;
#pragma unroll
  for (int w = 0; w < RP_THREADS / 64; ++w) carried += wsum[w];
  __syncthreads();

  int64_t t0 = sl.lo;
  unsigned T = rp_tile_size(carried);
  uint64_t k[RP_MAXITEMS];
  unsigned r[RP_MAXITEMS];
  int cnt = src.load(t0, sl.hi, (int)(T / RP_THREADS), k);
#pragma unroll
  for (int q = 0; q < RP_MAXITEMS; ++q)
    if (q < cnt) r[q] = atomicAdd(&newcnt[(unsigned)(k[q] >> shift) & (B - 1)], 1u);
  __syncthreads();

  while (true) {
    const bool last = t0 + T >= sl.hi;
    // layout of the round
    unsigned tot = 0;
    if (tid < B) { tot = rem + newcnt[tid]; newcnt[tid] = 0; }
    const unsigned nfl = last ? tot : (tot & ~15u);          // whole lines only, except in the slab's last round
    const unsigned nrem = tot - nfl;
    const unsigned packed = nfl | (nrem << 16);
    const unsigned inc = wave_inclusive_scan(packed);
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < RP_THREADS / 64; ++w) {
      const unsigned x = wsum[w];
      if (w < wave) wbase += x;
      total += x;
    }
    const unsigned ex = wbase + inc - packed;
    const unsigned total_f = total & 0xffffu;
    const unsigned fpos = ex & 0xffffu, cpos = total_f + (ex >> 16);
    if (tid < B) {
      meta[tid] = (uint64_t)fpos | ((uint64_t)nfl << 16) | ((uint64_t)cpos << 32) | ((uint64_t)rem << 48);
      line[tid] = (unsigned)(cursor >> 4);
      const unsigned base = nfl ? fpos : cpos;               // carried keys precede the new ones (rem < 16 <= nfl)
#pragma unroll
      for (int j = 0; j < 15; ++j)
        if (j < (int)rem) stage[base + j] = left[j];
    }
    __syncthreads();
    // stage the new keys behind the carried ones
#pragma unroll
    for (int q = 0; q < RP_MAXITEMS; ++q) {
      if (q < cnt) {
        const uint64_t m = meta[(unsigned)(k[q] >> shift) & (B - 1)];
        const unsigned j = (unsigned)(m >> 48) + r[q], f = (unsigned)(m >> 16) & 0xffffu;
        stage[j < f ? ((unsigned)m & 0xffffu) + j : ((unsigned)(m >> 32) & 0xffffu) + j - f] = k[q];
      }
    }
    __syncthreads();
    // flush: whole 128-byte lines, 16 bytes per lane
    if (!last) {
      for (unsigned i = 2 * tid; i < total_f; i += 2 * RP_THREADS) {
        const ulonglong2 kk = *reinterpret_cast<const ulonglong2*>(stage + i);
        const unsigned d = (unsigned)(kk.x >> shift) & (B - 1);
        const unsigned f = (unsigned)meta[d] & 0xffffu;
        uint64_t* dst = out + (((int64_t)line[d] << 4) + (i - f));
        if (!((kk.x | kk.y) >> 63)) {
          *reinterpret_cast<ulonglong2*>(dst) = kk;
        } else {                                             // phantom slots before the bucket's first key
          if (!(kk.x >> 63)) dst[0] = kk.x;
          if (!(kk.y >> 63)) dst[1] = kk.y;
        }
      }
    } else {
      for (unsigned i = tid; i < total_f; i += RP_THREADS) {
        const uint64_t key = stage[i];
        const unsigned d = (unsigned)(key >> shift) & (B - 1);
        if (!(key >> 63)) out[((int64_t)line[d] << 4) + (i - ((unsigned)meta[d] & 0xffffu))] = key;
      }
      break;
    }
    if (tid < B) {
#pragma unroll
      for (int j = 0; j < 15; ++j)
        if (j < (int)nrem) left[j] = stage[cpos + j];
      cursor += nfl;
      rem = nrem;
    }
    // next tile: load / generate and rank now, so that its latency overlaps the drain of the stores above
    t0 += T;
    T = rp_tile_size(total >> 16);
    cnt = src.load(t0, sl.hi, (int)(T / RP_THREADS), k);
#pragma unroll
    for (int q = 0; q < RP_MAXITEMS; ++q)
      if (q < cnt) r[q] = atomicAdd(&newcnt[(unsigned)(k[q] >> shift) & (B - 1)], 1u);
    __syncthreads();
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
                                        int64_t n_seg, int B, int64_t n, int64_t* __restrict__ child_off) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, m = n_seg * B;
  for (; i < m; i += stride) {
    const int64_t p = i / B, c = i - p * B;
    const int64_t first = seg_slabs[p], nsl = seg_slabs[p + 1] - first;
    child_off[i] = scanned[first * B + c * nsl];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) child_off[m] = n;
}

int64_t rp_slab_keys(int64_t n) {
  int64_t k = ceil_div(ceil_div(n, 4096), RP_TILE) * RP_TILE;
  return std::max<int64_t>(k, (int64_t)RP_TILE * 4);
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
  static bool attr_set[2] = {false, false};
  constexpr int which = std::is_same<Source, mem_source>::value ? 0 : 1;
  if (!attr_set[which]) {
    BNPK_HIP(ctx, hipFuncSetAttribute((const void*)rp_scatter_kernel<Source>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)RP_LDS));
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
    hipLaunchKernelGGL((rp_hist_kernel<Source>), dim3((unsigned)bound), dim3(RP_THREADS), RP_HIST_LDS, s, src, d_seg_off,
                       (const int64_t*)seg_slabs, n_seg, slab_keys, shift, bits, H);
    BNPK_HIP(ctx, hipGetLastError());
    BNPK_CHECK(bnpk_scan_launch(ctx, H, hn, 1, H, true, scan_scratch, s));
    if (d_child_off)
      hipLaunchKernelGGL(rp_child_offsets_kernel, dim3(grid_for(ceil_div(n_seg * B, 256))), dim3(256), 0, s,
                         (const int64_t*)H, (const int64_t*)seg_slabs, n_seg, B, n, d_child_off);
  }
  bnpk_timer t(ctx, scatter_name, s);
  hipLaunchKernelGGL((rp_scatter_kernel<Source>), dim3((unsigned)bound), dim3(RP_THREADS), RP_LDS, s, src, d_seg_off,
                     (const int64_t*)seg_slabs, n_seg, slab_keys, shift, bits, (const int64_t*)H,
                     reinterpret_cast<uint64_t*>(d_out));
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

// ===================================================================================================================
// Finishing kernel: every bucket of the partitioned keys (equal top bits, <= FN_CAP keys, arbitrary order inside)
// is sorted in LDS, its duplicates are counted and the distinct (key, count) pairs are written in sorted order.
// Two launches of the same kernel: COUNT stores the number of distinct keys per bucket, a device scan turns that
// into output offsets, WRITE repeats the LDS sort and stores at the final positions (no inter-workgroup waiting).
constexpr int FN_THREADS = 1024;
constexpr int FN_CAP = 8192;
constexpr int FN_ITEMS = FN_CAP / FN_THREADS;
constexpr int FN_MAXBITS = 12;
constexpr int FN_MAXBINS = 1 << FN_MAXBITS;
constexpr int FN_WORDS = FN_CAP / 64;
constexpr int FN_BINS_PER_LANE = FN_MAXBINS / FN_THREADS;
constexpr int FS_OVERFLOW = 0;                       // d_state words: [0] overflow flag, [8 ..] per-bucket counts -> offsets

constexpr size_t FN_OFF_BINS = (size_t)FN_CAP * 8;
constexpr size_t FN_OFF_CNT = FN_OFF_BINS + (size_t)(FN_MAXBINS + 4) * 4;
constexpr size_t FN_OFF_MASK = FN_OFF_CNT + (size_t)FN_CAP * 2;
constexpr size_t FN_OFF_PREFIX = FN_OFF_MASK + (size_t)FN_WORDS * 8;
constexpr size_t FN_OFF_WSUM = FN_OFF_PREFIX + (size_t)(FN_WORDS + 4) * 4;
constexpr size_t FN_LDS = FN_OFF_WSUM + 32 * 4;

template <bool WRITE>
__global__ __launch_bounds__(FN_THREADS) void finish_sorted_kernel(const uint64_t* __restrict__ A,
                                                                   const int64_t* __restrict__ bucket_off,
                                                                   int64_t n_buckets, int sshift, int sbits,
                                                                   int64_t* __restrict__ bucket_counts,
                                                                   uint64_t* __restrict__ keys_out,
                                                                   int64_t* __restrict__ counts_out,
                                                                   unsigned long long* __restrict__ flags) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* stage = reinterpret_cast<uint64_t*>(smem);
  unsigned* bins = reinterpret_cast<unsigned*>(smem + FN_OFF_BINS);
  unsigned short* cnt16 = reinterpret_cast<unsigned short*>(smem + FN_OFF_CNT);
  unsigned long long* fmask = reinterpret_cast<unsigned long long*>(smem + FN_OFF_MASK);
  unsigned* fprefix = reinterpret_cast<unsigned*>(smem + FN_OFF_PREFIX);
  unsigned* wsum = reinterpret_cast<unsigned*>(smem + FN_OFF_WSUM);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned SB = 1u << sbits;
  for (int64_t b = blockIdx.x; b < n_buckets; b += gridDim.x) {
    const int64_t lo = bucket_off[b], hi = bucket_off[b + 1];
    const int nb = (int)min(hi - lo, (int64_t)FN_CAP + 1);
    if (nb <= 0 || nb > FN_CAP) {                       // uniform per workgroup
      if (!WRITE && tid == 0) {
        bucket_counts[b] = 0;
        if (nb > FN_CAP) atomicOr(&flags[FS_OVERFLOW], 1ull);
      }
      continue;
    }
    for (unsigned i = tid; i <= SB; i += FN_THREADS) bins[i] = 0;
    if (tid < FN_WORDS) fmask[tid] = 0;
    __syncthreads();
    // counting sort on the next sbits bits: rank inside the bin from an LDS counter
    uint64_t k[FN_ITEMS];
    unsigned r[FN_ITEMS];
#pragma unroll
    for (int q = 0; q < FN_ITEMS; ++q) {
      const int i = tid + q * FN_THREADS;
      if (i < nb) {
        k[q] = A[lo + i];
        r[q] = atomicAdd(&bins[(unsigned)(k[q] >> sshift) & (SB - 1)], 1u);
      }
    }
    __syncthreads();
    {
      unsigned c[FN_BINS_PER_LANE], s = 0;
#pragma unroll
      for (int j = 0; j < FN_BINS_PER_LANE; ++j) {
        const unsigned bi = tid * FN_BINS_PER_LANE + j;
        c[j] = (bi < SB) ? bins[bi] : 0;
        s += c[j];
      }
      const unsigned inc = wave_inclusive_scan(s);
      if (lane == 63) wsum[wave] = inc;
      __syncthreads();
      unsigned run = inc - s;
      for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
      for (int j = 0; j < FN_BINS_PER_LANE; ++j) {
        const unsigned bi = tid * FN_BINS_PER_LANE + j;
        if (bi < SB) bins[bi] = run;
        run += c[j];
      }
      if (tid == 0) bins[SB] = (unsigned)nb;
    }
    __syncthreads();
    unsigned p[FN_ITEMS];
#pragma unroll
    for (int q = 0; q < FN_ITEMS; ++q) {
      const int i = tid + q * FN_THREADS;
      if (i < nb) {
        p[q] = bins[(unsigned)(k[q] >> sshift) & (SB - 1)] + r[q];
        stage[p[q]] = k[q];
      }
    }
    __syncthreads();
    // first occurrence of every distinct key inside its (tiny) bin
    unsigned first_bits = 0, dcount = 0;
#pragma unroll
    for (int q = 0; q < FN_ITEMS; ++q) {
      const int i = tid + q * FN_THREADS;
      if (i < nb) {
        const unsigned s = bins[(unsigned)(k[q] >> sshift) & (SB - 1)];
        bool first = true;
        for (unsigned j = s; j < p[q]; ++j)
          if (stage[j] == k[q]) { first = false; break; }
        if (first) {
          first_bits |= 1u << q;
          ++dcount;
          if (WRITE) atomicOr(&fmask[p[q] >> 6], 1ull << (p[q] & 63));
        }
      }
    }
    if (!WRITE) {
      const unsigned d = wave_reduce_sum(dcount);
      if (lane == 0) wsum[wave] = d;
      __syncthreads();
      if (tid == 0) {
        unsigned t = 0;
        for (int w = 0; w < FN_THREADS / 64; ++w) t += wsum[w];
        bucket_counts[b] = t;
      }
      __syncthreads();
      continue;
    }
    __syncthreads();
    if (tid < 64) {                                    // exclusive prefix of the popcounts of the mask words
      const unsigned c0 = __popcll(fmask[2 * tid]), c1 = __popcll(fmask[2 * tid + 1]);
      const unsigned inc = wave_inclusive_scan(c0 + c1);
      fprefix[2 * tid] = inc - c0 - c1;
      fprefix[2 * tid + 1] = inc - c1;
      if (tid == 63) fprefix[FN_WORDS] = inc;
    }
    __syncthreads();
    unsigned idx[FN_ITEMS], mult[FN_ITEMS];
#pragma unroll
    for (int q = 0; q < FN_ITEMS; ++q) {
      if (first_bits & (1u << q)) {
        const unsigned bin = (unsigned)(k[q] >> sshift) & (SB - 1);
        const unsigned s = bins[bin], e = bins[bin + 1];
        unsigned cnt = 0, rank = 0;
        for (unsigned j = s; j < e; ++j) {
          const uint64_t y = stage[j];
          cnt += (y == k[q]);
          rank += (y < k[q]) && ((fmask[j >> 6] >> (j & 63)) & 1ull);
        }
        mult[q] = cnt;
        idx[q] = fprefix[s >> 6] + __popcll(fmask[s >> 6] & ((1ull << (s & 63)) - 1ull)) + rank;
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < FN_ITEMS; ++q) {
      if (first_bits & (1u << q)) {
        stage[idx[q]] = k[q];
        cnt16[idx[q]] = (unsigned short)mult[q];
      }
    }
    __syncthreads();
    const unsigned D = fprefix[FN_WORDS];
    const int64_t base = bucket_counts[b];
    for (unsigned i = tid; i < D; i += FN_THREADS) {
      keys_out[base + i] = stage[i];
      counts_out[base + i] = cnt16[i];
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" {

int64_t bnpk_radix_max_bits(void) { return 10; }
int64_t bnpk_finish_capacity(void) { return FN_CAP; }

int bnpk_radix_partition(bnpk_ctx* ctx, const int64_t* d_keys, int64_t n, const int64_t* d_seg_offsets, int64_t n_seg,
                         int shift, int bits, int64_t* d_out, int64_t* d_child_offsets, void* stream) {
  if (!ctx || n < 0 || n_seg < 1 || bits < 0 || bits > 10 || shift < 0 || shift + bits > 63) return BNPK_ERR_ARG;
  if (n >= (1ll << 36)) return BNPK_ERR_RANGE;
  if (n > 0 && (!d_keys || !d_out || d_keys == d_out)) return BNPK_ERR_ARG;
  if (n_seg > 1 && !d_seg_offsets) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  void* scratch = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, rp_level_scratch(n, n_seg, bits), &scratch));
  mem_source src{reinterpret_cast<const uint64_t*>(d_keys)};
  return rp_level(ctx, src, n, d_seg_offsets, n_seg, shift, bits, d_out, d_child_offsets, (char*)scratch,
                  "radix_hist", "radix_scatter", s);
}

int bnpk_kmers_partition(bnpk_ctx* ctx, const uint64_t* d_packed, const int64_t* d_in_offsets,
                         const int64_t* d_out_offsets, int64_t n_rows, int64_t n_out, int k, int shift, int bits,
                         int64_t* d_out, int64_t* d_child_offsets, void* stream) {
  if (!ctx || k < 1 || k > 31 || n_rows < 0 || n_out < 0 || bits < 0 || bits > 10 || shift < 0 || shift + bits > 2 * k)
    return BNPK_ERR_ARG;
  if (n_out >= (1ll << 36)) return BNPK_ERR_RANGE;
  if (n_out > 0 && (!d_packed || !d_in_offsets || !d_out_offsets || !d_out || n_rows == 0)) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int64_t n_tiles = ceil_div(std::max<int64_t>(n_out, 1), RP_GRAN);
  const size_t table_bytes = align64(tile_rows_bytes(n_tiles));
  void* scratch = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, table_bytes + rp_level_scratch(n_out, 1, bits), &scratch));
  int64_t* table = (int64_t*)scratch;
  if (n_out > 0) BNPK_CHECK(build_tile_rows(ctx, d_out_offsets, n_rows, RP_GRAN, table, s));
  kmer_source src{d_packed, d_in_offsets, d_out_offsets, table, n_rows, n_tiles, (1ull << (2 * k)) - 1ull};
  return rp_level(ctx, src, n_out, nullptr, 1, shift, bits, d_out, d_child_offsets, (char*)scratch + table_bytes,
                  "kmers_partition_hist", "kmers_partition_scatter", s);
}

int64_t bnpk_finish_state_words(int64_t n_buckets) { return 8 + std::max<int64_t>(n_buckets, 0) + 1; }

int bnpk_finish_sorted(bnpk_ctx* ctx, const int64_t* d_part, int64_t n, const int64_t* d_bucket_offsets,
                       int64_t n_buckets, int low_bits, int64_t* d_keys_out, int64_t* d_counts_out, int64_t* d_state,
                       int64_t* h_n_unique, int* h_overflow, void* stream) {
  if (!ctx || n < 0 || n_buckets < 1 || low_bits < 0 || low_bits > 63 || !h_n_unique || !h_overflow || !d_state ||
      !d_bucket_offsets)
    return BNPK_ERR_ARG;
  *h_n_unique = 0;
  *h_overflow = 0;
  if (n == 0) return BNPK_OK;
  if (!d_part || !d_keys_out || !d_counts_out || d_keys_out == d_part) return BNPK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int sbits = std::min(low_bits, FN_MAXBITS);
  const int sshift = low_bits - sbits;
  unsigned long long* flags = reinterpret_cast<unsigned long long*>(d_state);
  int64_t* counts = d_state + 8;
  void* scratch = nullptr;
  BNPK_CHECK(bnpk_scratch(ctx, bnpk_scan_scratch_bytes(n_buckets), &scratch));
  static bool attr_set = false;
  if (!attr_set) {
    BNPK_HIP(ctx, hipFuncSetAttribute((const void*)finish_sorted_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)FN_LDS));
    BNPK_HIP(ctx, hipFuncSetAttribute((const void*)finish_sorted_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)FN_LDS));
    attr_set = true;
  }
  BNPK_HIP(ctx, hipMemsetAsync(d_state, 0, 8 * sizeof(int64_t), s));
  const unsigned grid = (unsigned)std::min<int64_t>(n_buckets, (int64_t)ctx->compute_units * 16);
  const uint64_t* A = reinterpret_cast<const uint64_t*>(d_part);
  {
    bnpk_timer t(ctx, "finish_sorted_count", s);
    hipLaunchKernelGGL((finish_sorted_kernel<false>), dim3(grid), dim3(FN_THREADS), FN_LDS, s, A, d_bucket_offsets,
                       n_buckets, sshift, sbits, counts, (uint64_t*)nullptr, (int64_t*)nullptr, flags);
    BNPK_HIP(ctx, hipGetLastError());
    BNPK_CHECK(bnpk_scan_launch(ctx, counts, n_buckets, 1, counts, true, (int64_t*)scratch, s));
  }
  int64_t host_flag = 0, total = 0;
  BNPK_HIP(ctx, hipMemcpyAsync(&host_flag, d_state, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  BNPK_HIP(ctx, hipMemcpyAsync(&total, counts + n_buckets, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  BNPK_HIP(ctx, hipStreamSynchronize(s));
  *h_overflow = host_flag != 0;
  if (*h_overflow) return BNPK_OK;
  *h_n_unique = total;
  bnpk_timer t(ctx, "finish_sorted_write", s);
  hipLaunchKernelGGL((finish_sorted_kernel<true>), dim3(grid), dim3(FN_THREADS), FN_LDS, s, A, d_bucket_offsets,
                     n_buckets, sshift, sbits, counts, reinterpret_cast<uint64_t*>(d_keys_out), d_counts_out, flags);
  BNPK_HIP(ctx, hipGetLastError());
  return BNPK_OK;
}

}  // extern "C"
