// Experiment (not product): HBM write rate of the radix-scatter store pattern on MI355X as a function of the
// run length per (tile, bucket) and of the run alignment.  Write-only (keys come from registers).
// hipcc --offload-arch=gfx950 -O3 scripts/exp/exp_write.hip -o scripts/bin/exp_write
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void fill16(uint4* out, int64_t n16) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x, st = (int64_t)gridDim.x * 256;
  for (; i < n16; i += st) out[i] = make_uint4((unsigned)i, 1, 2, 3);
}
__global__ __launch_bounds__(256) void copy16(const uint4* __restrict__ in, uint4* __restrict__ out, int64_t n16) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x, st = (int64_t)gridDim.x * 256;
  for (; i < n16; i += st) out[i] = in[i];
}
__global__ __launch_bounds__(256) void read16(const uint4* __restrict__ in, unsigned* sink, int64_t n16) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x, st = (int64_t)gridDim.x * 256;
  unsigned a = 0;
  for (; i < n16; i += st) { uint4 v = in[i]; a ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (a == 0x12345) sink[0] = a;
}

// B buckets, each (bucket, workgroup) owns a contiguous segment; per tile every bucket receives a run of L keys
template <int THREADS, int WIDE>
__global__ __launch_bounds__(THREADS) void scatter_runs(uint64_t* __restrict__ out, int B, int L, int tiles, int64_t seg,
                                                        int phase) {
  const int nb = gridDim.x, w = blockIdx.x;
  const int tile_keys = B * L;
  for (int t = 0; t < tiles; ++t) {
    if (WIDE == 1) {
      for (int i = threadIdx.x; i < tile_keys; i += THREADS) {
        int d = i / L, j = i - d * L;
        int64_t at = ((int64_t)d * nb + w) * seg + phase + (int64_t)t * L + j;
        out[at] = ((uint64_t)d << 40) | (unsigned)i;
      }
    } else {   // 16-byte stores: two keys per lane (needs L even and even phase)
      for (int i = 2 * threadIdx.x; i < tile_keys; i += 2 * THREADS) {
        int d = i / L, j = i - d * L;
        int64_t at = ((int64_t)d * nb + w) * seg + phase + (int64_t)t * L + j;
        *reinterpret_cast<ulonglong2*>(out + at) = make_ulonglong2(((uint64_t)d << 40) | (unsigned)i, i);
      }
    }
  }
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  int64_t n = argc > 1 ? atoll(argv[1]) : 1200000000ll;
  uint64_t *a, *b; unsigned* sink;
  CK(hipMalloc(&a, n * 8 + (1 << 20))); CK(hipMalloc(&b, n * 8 + (1 << 20))); CK(hipMalloc(&sink, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](auto fn, int reps) { float best = 1e9; for (int r = 0; r < reps; ++r) { CK(hipEventRecord(e0)); fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; } return best; };
  for (int grid : {2048, 8192, 65536}) {
    float tf = timeit([&] { hipLaunchKernelGGL(fill16, dim3(grid), dim3(256), 0, 0, (uint4*)a, n / 2); }, 3);
    float tc = timeit([&] { hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, n / 2); }, 3);
    float tr = timeit([&] { hipLaunchKernelGGL(read16, dim3(grid), dim3(256), 0, 0, (const uint4*)a, sink, n / 2); }, 3);
    printf("grid %6d: fill %.3f ms (%.0f GB/s w)   copy %.3f ms (%.0f GB/s r+w)   read %.3f ms (%.0f GB/s r)\n", grid, tf, n * 8.0 / tf / 1e6, tc,
           n * 16.0 / tc / 1e6, tr, n * 8.0 / tr / 1e6);
  }
  for (int B : {256, 1024}) for (int nb : {512, 2048}) for (int L : {4, 8, 16, 32, 64, 128, 256}) {
    if ((int64_t)B * L > 65536) continue;
    int tiles = (int)((n / ((int64_t)nb * B) - 48) / L); if (tiles < 1) continue;
    int64_t seg = ((int64_t)tiles * L + 16 + 15) & ~15ll;   // segments start 128-B aligned; room for the phase
    if ((int64_t)B * nb * seg > n) { printf("skip\n"); continue; }
    double bytes = (double)nb * tiles * B * L * 8.0;
    for (int phase : {0, 2, 5}) {
      float t8 = timeit([&] { hipLaunchKernelGGL((scatter_runs<1024, 1>), dim3(nb), dim3(1024), 0, 0, a, B, L, tiles, seg, phase); }, 2);
      float t16 = -1;
      if (phase % 2 == 0 && L % 2 == 0) t16 = timeit([&] { hipLaunchKernelGGL((scatter_runs<1024, 2>), dim3(nb), dim3(1024), 0, 0, a, B, L, tiles, seg, phase); }, 2);
      printf("B=%4d nb=%4d L=%3d keys (%4d B) phase=%d: 8B-stores %.3f ms %.0f GB/s   16B-stores %.3f ms %.0f GB/s\n", B, nb, L, L * 8, phase, t8,
             bytes / t8 / 1e6, t16, t16 > 0 ? bytes / t16 / 1e6 : 0.0);
    }
  }
  return 0;
}
