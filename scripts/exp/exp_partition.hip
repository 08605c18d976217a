// Experiment (not product): throughput of a two-kernel LDS-staged radix partition on MI355X for different
// digit widths / tile sizes.  hipcc --offload-arch=gfx950 -O3 scripts/exp_partition.hip -o /tmp/exp_partition
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__global__ void gen(uint64_t* k, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, st = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) k[i] = mix64(i) >> 2;
}

template <int BITS, int THREADS>
__global__ __launch_bounds__(THREADS) void hist_kernel(const uint64_t* __restrict__ keys, int64_t n, int64_t slab, int shift,
                                                       unsigned* __restrict__ H, int nb) {
  constexpr int B = 1 << BITS;
  __shared__ unsigned h[B];
  for (int i = threadIdx.x; i < B; i += THREADS) h[i] = 0;
  __syncthreads();
  int64_t lo = (int64_t)blockIdx.x * slab, hi = min(lo + slab, n);
  for (int64_t i = lo + 2 * threadIdx.x; i < hi; i += 2 * THREADS) {
    if (i + 1 < hi) {
      ulonglong2 v = *reinterpret_cast<const ulonglong2*>(keys + i);
      atomicAdd(&h[(v.x >> shift) & (B - 1)], 1u);
      atomicAdd(&h[(v.y >> shift) & (B - 1)], 1u);
    } else atomicAdd(&h[(keys[i] >> shift) & (B - 1)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < B; i += THREADS) H[(int64_t)i * nb + blockIdx.x] = h[i];
}

template <int BITS, int THREADS, int ITEMS>
__global__ __launch_bounds__(THREADS) void scatter_kernel(const uint64_t* __restrict__ keys, int64_t n, int64_t slab, int shift,
                                                          const int64_t* __restrict__ offs, int nb, uint64_t* __restrict__ out) {
  constexpr int B = 1 << BITS;
  constexpr int TILE = THREADS * ITEMS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* stage = reinterpret_cast<uint64_t*>(smem);                 // TILE keys
  int64_t* cursor = reinterpret_cast<int64_t*>(stage + TILE);         // B
  unsigned* cnt = reinterpret_cast<unsigned*>(cursor + B);             // B
  unsigned* start = cnt + B;                                            // B
  __shared__ unsigned wsum[THREADS / 64 + 1];
  for (int i = threadIdx.x; i < B; i += THREADS) { cursor[i] = offs[(int64_t)i * nb + blockIdx.x]; cnt[i] = 0; }
  __syncthreads();
  int64_t lo = (int64_t)blockIdx.x * slab, hi = min(lo + slab, n);
  for (int64_t t0 = lo; t0 < hi; t0 += TILE) {
    uint64_t k[ITEMS]; unsigned r[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      int64_t i = t0 + threadIdx.x + (int64_t)j * THREADS;
      bool ok = i < hi;
      k[j] = ok ? keys[i] : ~0ull;
      r[j] = ok ? atomicAdd(&cnt[(k[j] >> shift) & (B - 1)], 1u) : 0;
    }
    __syncthreads();
    // exclusive scan of cnt[B] -> start[B]; each thread owns B/THREADS consecutive bins
    constexpr int PER = B / THREADS > 0 ? B / THREADS : 1;
    unsigned loc[PER]; unsigned s = 0;
    if (threadIdx.x * PER < B) {
#pragma unroll
      for (int q = 0; q < PER; ++q) { loc[q] = s; s += cnt[threadIdx.x * PER + q]; }
    }
    unsigned inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { unsigned o = __shfl_up(inc, d, 64); if ((threadIdx.x & 63) >= d) inc += o; }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    unsigned base = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wsum[w];
    if (threadIdx.x * PER < B) {
#pragma unroll
      for (int q = 0; q < PER; ++q) start[threadIdx.x * PER + q] = base + inc - s + loc[q];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      int64_t i = t0 + threadIdx.x + (int64_t)j * THREADS;
      if (i < hi) stage[start[(k[j] >> shift) & (B - 1)] + r[j]] = k[j];
    }
    __syncthreads();
    int m = (int)min((int64_t)TILE, hi - t0);
    for (int i = threadIdx.x; i < m; i += THREADS) {
      uint64_t key = stage[i];
      unsigned d = (key >> shift) & (B - 1);
      out[cursor[d] + (i - start[d])] = key;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < B; i += THREADS) { cursor[i] += cnt[i]; cnt[i] = 0; }
    __syncthreads();
  }
}

template <int BITS, int THREADS, int ITEMS>
void run(const uint64_t* keys, uint64_t* out, int64_t n, int nb, const char* label) {
  constexpr int B = 1 << BITS;
  int shift = 62 - BITS;
  int64_t slab = ((n + nb - 1) / nb + 1) & ~1ll;
  unsigned* H; int64_t* offs;
  CK(hipMalloc(&H, sizeof(unsigned) * B * (size_t)nb)); CK(hipMalloc(&offs, sizeof(int64_t) * B * (size_t)nb));
  std::vector<unsigned> h((size_t)B * nb); std::vector<int64_t> o((size_t)B * nb);
  hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  size_t lds = (size_t)THREADS * ITEMS * 8 + (size_t)B * 16;
  CK(hipFuncSetAttribute((const void*)scatter_kernel<BITS, THREADS, ITEMS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  float best_h = 1e9, best_s = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((hist_kernel<BITS, 256>), dim3(nb), dim3(256), 0, 0, keys, n, slab, shift, H, nb);
    CK(hipEventRecord(e1));
    CK(hipMemcpy(h.data(), H, h.size() * 4, hipMemcpyDeviceToHost));
    int64_t run = 0; for (size_t i = 0; i < h.size(); ++i) { o[i] = run; run += h[i]; }
    CK(hipMemcpy(offs, o.data(), o.size() * 8, hipMemcpyHostToDevice));
    CK(hipEventRecord(e1));
    hipLaunchKernelGGL((scatter_kernel<BITS, THREADS, ITEMS>), dim3(nb), dim3(THREADS), lds, 0, keys, n, slab, shift, offs, nb, out);
    CK(hipEventRecord(e2)); CK(hipEventSynchronize(e2));
    float th, ts; CK(hipEventElapsedTime(&ts, e1, e2));
    CK(hipEventRecord(e0)); hipLaunchKernelGGL((hist_kernel<BITS, 256>), dim3(nb), dim3(256), 0, 0, keys, n, slab, shift, H, nb);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&th, e0, e1));
    if (th < best_h) best_h = th; if (ts < best_s) best_s = ts;
  }
  // verify partition order on a sample
  std::vector<uint64_t> chk(1 << 20); CK(hipMemcpy(chk.data(), out + n / 3, chk.size() * 8, hipMemcpyDeviceToHost));
  bool ok = true; for (size_t i = 1; i < chk.size(); ++i) if ((chk[i] >> shift) < (chk[i - 1] >> shift)) ok = false;
  printf("%-34s nb=%5d  hist %.3f ms (%.0f GB/s)  scatter %.3f ms (%.0f GB/s r+w)  lds=%zu  %s\n", label, nb, best_h,
         n * 8.0 / best_h / 1e6, best_s, n * 16.0 / best_s / 1e6, lds, ok ? "ok" : "ORDER-BAD");
  CK(hipFree(H)); CK(hipFree(offs));
}

int main(int argc, char** argv) {
  int64_t n = argc > 1 ? atoll(argv[1]) : 1200000000ll;
  uint64_t *keys, *out;
  CK(hipMalloc(&keys, n * 8)); CK(hipMalloc(&out, n * 8));
  hipLaunchKernelGGL(gen, dim3(4096), dim3(256), 0, 0, keys, n);
  CK(hipDeviceSynchronize());
  for (int nb : {512, 1024, 2048, 4096, 16384}) {
    run<8, 256, 16>(keys, out, n, nb, "8 bits, 256 thr x16 (tile 4096)");
    run<8, 512, 16>(keys, out, n, nb, "8 bits, 512 thr x16 (tile 8192)");
    run<8, 1024, 8>(keys, out, n, nb, "8 bits, 1024 thr x8 (tile 8192)");
    run<8, 256, 8>(keys, out, n, nb, "8 bits, 256 thr x8 (tile 2048)");
    run<9, 512, 16>(keys, out, n, nb, "9 bits, 512 thr x16 (tile 8192)");
    run<11, 1024, 12>(keys, out, n, nb, "11 bits, 1024 thr x12 (tile 12288)");
  }
  return 0;
}
