// Internal helpers shared by the gfx950 kernels: ctx, scratch arena, launch timers, block scans.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/bnpk.h"

// First statement of a kernel whose code would otherwise be given exactly 24 VGPRs (all three 8-register granules in use): the
// allocation becomes 32, nothing is emitted.  Round 5 found rc_packed_kernel's fully unrolled form — correct ISA by every check,
// byte-identical with and without this line — computing garbage in every workgroup that is not the first on its CU when it runs
// with a 24-register allocation, and never with 32 (NOTES.md "rc_packed: the cause"; csrc/isa_lint.py reports such kernels).
#define BNPK_VGPR_FLOOR_32() asm volatile("" ::: "v31")
#define BNPK_WAVE 64
#define BNPK_BLOCK 256

struct bnpk_prof_entry {
  std::string name;
  double total_ms = 0.0;
  int64_t launches = 0;
};

struct bnpk_pending_event {
  int entry;
  hipEvent_t start, stop;
};

struct bnpk_ctx {
  int device = 0;
  int compute_units = 256;
  void* scratch = nullptr;       // grow-only device arena for scan partials and small temporaries
  size_t scratch_bytes = 0;
  hipStream_t scratch_stream = nullptr;   // the stream of the last call that used the arena (see bnpk_scratch)
  hipEvent_t scratch_event = nullptr;
  bool scratch_in_use = false;
  bool launch_attr_set[8] = {false, false, false, false, false, false, false, false};   // per-device kernel attributes (radix.hip)
  bool prof = false;
  std::vector<bnpk_prof_entry> entries;
  std::vector<bnpk_pending_event> pending;
  std::vector<hipEvent_t> event_pool;
  hipError_t last_err = hipSuccess;
  // finishing kernels (finish.hip): launch attributes set / co-resident workgroups of the fast kernel / forced path
  bool finish_ready = false;
  int finish_fast_grid = 0;
  int finish_mode = 0;           // 0 = choose per call, 1 = general kernel only, 2 = fast kernel + redo list only, 3 = duplicate-aware only
  bool finish_dup_ready = false; // finish_dup.hip: launch attributes set / co-resident workgroups
  int finish_dup_grid = 0;
  bool finish_wave_ready = false;   // finish_wave.hip
  int finish_wave_grid = 0;
  bool finish_small_ready = false;  // finish_small.hip
  bool finish_multi_ready = false;  // finish_multi.hip
  int finish_multi_grid = 0;
  void* mailbox = nullptr;       // page-locked words the host scalars of a call come back through (bnpk_fetch_i64)
  int fastq_encoder = 1;         // fastq.hip: 1 = fast tile encoder + the general one for the tiles it hands back, 0 = general only
  int index_pairs = 1;           // sparse.hip: bnpk_index_build as one partition of (k-mer, row) words where the rows fit (0: always by ranks)
  int sparse_claim = 1;          // sparse.hip: may bnpk_count_sparse run its last level as the claiming level (0: never — tests, experiments)
  int l1_ring = 0;               // radix.hip: the fused first level's scatter — 0 = rp_scatter_kernel, 1 = rp_ring_kernel (one fixed line per bucket: round 6, measured slower)
};

#define BNPK_HIP(ctx, call)                          \
  do {                                               \
    hipError_t e__ = (call);                         \
    if (e__ != hipSuccess) {                         \
      if (ctx) (ctx)->last_err = e__;                \
      return BNPK_ERR_HIP;                           \
    }                                                \
  } while (0)

#define BNPK_CHECK(expr)                 \
  do {                                   \
    int s__ = (expr);                    \
    if (s__ != BNPK_OK) return s__;      \
  } while (0)

// Scratch arena: returns a device pointer to at least `bytes` bytes (valid until the next call that
// grows it; every entry point carves what it needs up front, so a single request per call).  The arena is shared by
// all entry points of the ctx: a call on another stream than the previous user's is ordered behind that user's work
// with an event (two streams never overwrite each other's tables); a ctx is for one host thread at a time.
int bnpk_scratch(bnpk_ctx* ctx, size_t bytes, void** out, hipStream_t stream);

// RAII hipEvent timer around a launch (or a group of launches) when profiling is enabled.
struct bnpk_timer {
  bnpk_ctx* ctx;
  hipStream_t stream;
  int entry = -1;
  hipEvent_t start = nullptr, stop = nullptr;
  bnpk_timer(bnpk_ctx* c, const char* name, hipStream_t s);
  ~bnpk_timer();
};

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// The AQL dispatch packet carries the grid size in WORK-ITEMS as a uint32, so gridDim.x * blockDim.x must
// stay below 2^32: element-wise kernels are grid-stride loops launched with at most BNPK_MAX_BLOCKS
// workgroups of 256 threads, tile kernels check their tile count against it.
#define BNPK_MAX_BLOCKS ((int64_t)((1ull << 32) / 256 - 1))
static inline unsigned grid_for(int64_t blocks) {
  if (blocks < 1) blocks = 1;
  return (unsigned)(blocks > BNPK_MAX_BLOCKS ? BNPK_MAX_BLOCKS : blocks);
}

// ------------------------------------------------------------------------------------------ device
#ifdef __HIPCC__

// the 2-bit groups of x in reverse order (group 0 <-> group 31): the reverse of a run of bases in the packed layout
__device__ __forceinline__ uint64_t reverse_2bit_groups(uint64_t x) {
  x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
  x = ((x >> 4) & 0x0f0f0f0f0f0f0f0full) | ((x & 0x0f0f0f0f0f0f0f0full) << 4);
  return __builtin_bswap64(x);
}

// bit j of the result = byte j of the sixteen bytes w[0..3] is NOT `rep`'s byte.  Per 32-bit word the classic exact
// zero-byte test leaves 0x7f in a matching byte and 0xff in any other; V_DOT4_U32_U8 against the weights 1,2,4,...,128 then
// gathers eight flags at a time: sum(w_i * g_i) = 0x7f * 255 + 0x80 * (mask of the non-matching bytes), and the
// constant goes into the accumulator.  (About 27 vector instructions; the shift-and-mask gather costs twice that, and a
// pass over text has ~100 instructions per sixteen bytes to spend before it, not HBM, sets the pace.)
__device__ __forceinline__ uint32_t nomatch16(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t rep) {
  const uint32_t w[4] = {w0, w1, w2, w3};
  uint32_t g[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t x = w[q] ^ rep;
    g[q] = ((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu;
  }
  const uint32_t bias = 0u - 0x7fu * 255u;
  const uint32_t a = __builtin_amdgcn_udot4(g[1], 0x80402010u, __builtin_amdgcn_udot4(g[0], 0x08040201u, bias, false), false);
  const uint32_t b = __builtin_amdgcn_udot4(g[3], 0x80402010u, __builtin_amdgcn_udot4(g[2], 0x08040201u, bias, false), false);
  return (a >> 7) | (b << 1);                                // a, b = 128 * (eight flags)
}
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// inclusive scan over the 64 lanes of a wavefront
template <typename T>
__device__ __forceinline__ T wave_inclusive_scan(T v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    T o = __shfl_up(v, d, 64);
    if (lane_id() >= d) v += o;
  }
  return v;
}

// Exclusive scan across a 256-thread block (4 wavefronts). `smem` needs 5 T's. Returns the
// exclusive prefix of this thread and writes the block total to *total.
template <typename T>
__device__ __forceinline__ T block_exclusive_scan(T v, T* smem, T* total) {
  T inc = wave_inclusive_scan(v);
  if (lane_id() == 63) smem[wave_id()] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    T run = 0;
#pragma unroll
    for (int w = 0; w < BNPK_BLOCK / 64; ++w) {
      T t = smem[w];
      smem[w] = run;
      run += t;
    }
    smem[BNPK_BLOCK / 64] = run;
  }
  __syncthreads();
  T base = smem[wave_id()];
  *total = smem[BNPK_BLOCK / 64];
  __syncthreads();
  return base + inc - v;
}

// 32-bit inclusive scan in six DPP moves (row_shr 1,2,4,8 inside the rows of 16 lanes, then row_bcast:15 / :31
// across rows) instead of six LDS-latency ds_bpermute round trips.  All 64 lanes must be active.
__device__ __forceinline__ unsigned wave_inclusive_scan(unsigned v) {
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
  return v;
}
__device__ __forceinline__ int wave_inclusive_scan(int v) { return (int)wave_inclusive_scan((unsigned)v); }
__device__ __forceinline__ unsigned wave_max(unsigned v) {          // largest of the 64 lanes, in every lane
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_sum(unsigned v) {          // total of the 64 lanes, in every lane
  return (unsigned)__builtin_amdgcn_readlane((int)wave_inclusive_scan(v), 63);
}

template <typename T>
__device__ __forceinline__ T wave_reduce_sum(T v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
  return v;
}

template <typename T>
__device__ __forceinline__ T wave_reduce_max(T v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const T o = __shfl_down(v, d, 64);
    v = o > v ? o : v;
  }
  return v;
}

// largest r in [lo, hi] with offsets[r] <= pos (offsets non-decreasing, offsets[lo] <= pos).
// With runs of equal offsets (empty rows) this returns the LAST such r, i.e. the non-empty row
// that actually contains pos.
__device__ __forceinline__ int64_t find_row(const int64_t* __restrict__ offsets, int64_t lo,
                                            int64_t hi, int64_t pos) {
  while (lo < hi) {
    int64_t mid = lo + ((hi - lo + 1) >> 1);
    if (offsets[mid] <= pos) lo = mid; else hi = mid - 1;
  }
  return lo;
}

#endif  // __HIPCC__
