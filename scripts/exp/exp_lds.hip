// Experiment (not product): LDS throughput on MI355X for the access patterns of the radix kernels:
// random-bin atomic add (with / without return), random 8-byte writes, random 8-byte reads.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
template <int MODE, int BINS>
__global__ __launch_bounds__(1024) void k(unsigned* sink, int iters) {
  __shared__ unsigned h[BINS];
  __shared__ uint64_t st[8192];
  for (int i = threadIdx.x; i < BINS; i += 1024) h[i] = 0;
  for (int i = threadIdx.x; i < 8192; i += 1024) st[i] = i;
  __syncthreads();
  uint64_t x = mix64(blockIdx.x * 1024 + threadIdx.x);
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    // 8 independent ops per iteration, addresses from a cheap LCG
    unsigned a[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { x = x * 6364136223846793005ull + 1442695040888963407ull; a[q] = (unsigned)(x >> 40); }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (MODE == 0) atomicAdd(&h[a[q] & (BINS - 1)], 1u);
      if (MODE == 1) acc += atomicAdd(&h[a[q] & (BINS - 1)], 1u);
      if (MODE == 2) st[a[q] & 8191] = x;
      if (MODE == 3) acc += (unsigned)st[a[q] & 8191];
      if (MODE == 4) acc += a[q];     // address generation only
    }
  }
  __syncthreads();
  if (acc == 0x12345 || h[threadIdx.x & (BINS - 1)] == 0xffffffffu) sink[0] = acc + (unsigned)st[threadIdx.x];
}
int main() {
  unsigned* sink; CK(hipMalloc(&sink, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2000, grid = 512;
  const char* names[] = {"ds_add (no return)", "ds_add_rtn", "ds_write_b64 random", "ds_read_b64 random", "address generation only"};
  auto run = [&](auto kern, const char* name, int bins) {
    float best = 1e9;
    for (int r = 0; r < 3; ++r) { CK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), 0, 0, sink, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
    double ops = (double)grid * 1024 * iters * 8;
    printf("%-26s bins=%5d: %.3f ms  %.1f G lane-ops/s  = %.2f lanes/clk/CU (2.4 GHz, 256 CUs)\n", name, bins, best, ops / best / 1e6, ops / best / 1e6 / 256 / 2.4);
  };
  run(k<4, 1024>, names[4], 0);
  run(k<0, 256>, names[0], 256); run(k<0, 1024>, names[0], 1024); run(k<0, 4096>, names[0], 4096);
  run(k<1, 256>, names[1], 256); run(k<1, 1024>, names[1], 1024); run(k<1, 4096>, names[1], 4096);
  run(k<2, 1024>, names[2], 0); run(k<3, 1024>, names[3], 0);
  return 0;
}
