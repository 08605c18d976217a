// LDS atomic rates on gfx950: cycles per wave-instruction with 16 waves per CU (2 x 512 threads), by address pattern.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); return 1; } } while (0)
constexpr int SLOTS = 7680, ITERS = 64, UNR = 8;
template <int OP>
__global__ __launch_bounds__(512) void k(const unsigned* __restrict__ addr, unsigned long long* out, unsigned long long* cyc) {
  extern __shared__ unsigned long long T[];
  unsigned* T32 = reinterpret_cast<unsigned*>(T);
  for (int i = threadIdx.x; i < SLOTS; i += 512) T[i] = OP == 2 ? ~0ull : 0ull;
  __syncthreads();
  unsigned a[UNR];
  for (int u = 0; u < UNR; ++u) a[u] = addr[(blockIdx.x * UNR + u) * 512 + threadIdx.x];
  unsigned long long acc = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (OP == 0) atomicAdd(&T32[a[u]], 1u);                                   // ds_add_u32
      if (OP == 1) acc += atomicAdd(&T32[a[u]], 1u);                            // ds_add_rtn_u32
      if (OP == 2) acc += atomicCAS(&T[a[u]], ~0ull, (unsigned long long)a[u] + it);   // ds_cmpst_rtn_b64
      if (OP == 3) acc += T[a[u]];                                              // ds_read_b64
      if (OP == 4) T[a[u]] = acc + it;                                          // ds_write_b64
      if (OP == 5) atomicAdd(&T[a[u]], 1ull);                                   // ds_add_u64
    }
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 512 + threadIdx.x] = acc + T[threadIdx.x];
}
static unsigned rng(unsigned long long& s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(s >> 33); }
int main() {
  const int blocks = 512;
  const size_t na = (size_t)blocks * UNR * 512;
  unsigned* h = (unsigned*)malloc(na * 4);
  unsigned *d; unsigned long long *out, *cyc, hc[512];
  CHECK(hipMalloc(&d, na * 4)); CHECK(hipMalloc(&out, blocks * 512 * 8)); CHECK(hipMalloc(&cyc, blocks * 8));
  const char* pat[] = {"lane-consecutive", "random of 7680", "random of 1000", "random of 95", "random of 8", "all one"};
  const char* ops[] = {"ds_add_u32", "ds_add_rtn_u32", "ds_cmpst_rtn_b64", "ds_read_b64", "ds_write_b64", "ds_add_u64"};
  for (int p = 0; p < 6; ++p) {
    unsigned long long s = 12345;
    for (size_t i = 0; i < na; ++i) {
      const unsigned r = rng(s);
      h[i] = p == 0 ? (unsigned)(i % 512) : p == 1 ? r % SLOTS : p == 2 ? (r % 1000) * 7 : p == 3 ? (r % 95) * 79 : p == 4 ? (r % 8) * 901 : 77;
    }
    CHECK(hipMemcpy(d, h, na * 4, hipMemcpyHostToDevice));
    for (int op = 0; op < 6; ++op) {
      for (int rep = 0; rep < 2; ++rep) {
        if (op == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), SLOTS * 8, 0, d, out, cyc);
        if (op == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), SLOTS * 8, 0, d, out, cyc);
        if (op == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(512), SLOTS * 8, 0, d, out, cyc);
        if (op == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(512), SLOTS * 8, 0, d, out, cyc);
        if (op == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(512), SLOTS * 8, 0, d, out, cyc);
        if (op == 5) hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(512), SLOTS * 8, 0, d, out, cyc);
        CHECK(hipDeviceSynchronize());
      }
      CHECK(hipMemcpy(hc, cyc, blocks * 8, hipMemcpyDeviceToHost));
      double sum = 0; for (int b = 0; b < blocks; ++b) sum += (double)hc[b];
      // a workgroup issues 8 waves x ITERS x UNR instructions; two workgroups share a CU
      printf("%-18s %-18s %7.1f cycles per wave-instruction per CU\n", pat[p], ops[op], sum / blocks / (ITERS * UNR * 8.0) / 2.0);
    }
  }
  return 0;
}
