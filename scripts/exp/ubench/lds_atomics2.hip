// LDS atomic rates as the wavefront-per-bucket kernel uses them: one-wavefront workgroups with a 9.5 KB table each (16-17 per
// CU), groups of four operations whose results are waited for before the next group.  Cycles per wave-instruction per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); return 1; } } while (0)
constexpr int SLOTS = 704, ITERS = 256, LDS_BYTES = 9472;
template <int OP, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k(const unsigned* __restrict__ addr, unsigned long long* out, unsigned long long* cyc) {
  extern __shared__ unsigned long long T[];
  for (int i = threadIdx.x; i < SLOTS; i += BLOCK) T[i] = OP == 2 ? ~0ull : 0ull;
  __syncthreads();
  unsigned long long acc = 0;
  const unsigned* my = addr + (size_t)(blockIdx.x * BLOCK + threadIdx.x) * 4;
  unsigned a[4];
  for (int u = 0; u < 4; ++u) a[u] = my[u] % SLOTS;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITERS; ++it) {
    unsigned long long r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (OP == 2) r[u] = atomicCAS(&T[a[u]], ~0ull, (unsigned long long)a[u] + 1);          // ds_cmpst_rtn_b64
      if (OP == 3) r[u] = T[a[u]];                                                            // ds_read_b64
      if (OP == 1) r[u] = atomicAdd(reinterpret_cast<unsigned*>(T) + a[u], 1u);              // ds_add_rtn_u32
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {                        // the next group's addresses depend on this group's answers
      acc += r[u];
      a[u] = (a[u] + (unsigned)(r[u] & 1u) * 0u + 7u * (unsigned)(it & 1)) % SLOTS;
      asm volatile("" : "+v"(a[u]) : "v"(r[u]));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * BLOCK + threadIdx.x] = acc;
}
static unsigned rng(unsigned long long& s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(s >> 33); }
int main() {
  const int per_cu = 16, cus = 256;
  const size_t threads = (size_t)per_cu * cus * 64, na = threads * 4;
  unsigned* h = (unsigned*)malloc(na * 4);
  unsigned *d; unsigned long long *out, *cyc, *hc = (unsigned long long*)malloc(per_cu * cus * 8);
  CHECK(hipMalloc(&d, na * 4)); CHECK(hipMalloc(&out, threads * 8)); CHECK(hipMalloc(&cyc, per_cu * cus * 8));
  const char* pat[] = {"random of 704", "random of 95", "random of 8"};
  const char* ops[] = {"", "ds_add_rtn_u32", "ds_cmpst_rtn_b64", "ds_read_b64"};
  for (int p = 0; p < 3; ++p) {
    unsigned long long s = 12345;
    for (size_t i = 0; i < na; ++i) { const unsigned r = rng(s); h[i] = p == 0 ? r % 704 : p == 1 ? (r % 95) * 7 : (r % 8) * 83; }
    CHECK(hipMemcpy(d, h, na * 4, hipMemcpyHostToDevice));
    for (int op = 1; op <= 3; ++op) {
      for (int shape = 0; shape < 2; ++shape) {                       // 16 one-wave workgroups per CU / 2 workgroups of 8 waves
        const int blocks = shape == 0 ? per_cu * cus : 2 * cus, block = shape == 0 ? 64 : 512;
        const size_t lds = shape == 0 ? LDS_BYTES : 8 * LDS_BYTES;
        for (int rep = 0; rep < 2; ++rep) {
#define L(OP, B) hipLaunchKernelGGL((k<OP, B>), dim3(blocks), dim3(B), lds, 0, d, out, cyc)
          if (shape == 0) { if (op == 1) L(1, 64); if (op == 2) L(2, 64); if (op == 3) L(3, 64); }
          else { if (op == 1) L(1, 512); if (op == 2) L(2, 512); if (op == 3) L(3, 512); }
          CHECK(hipDeviceSynchronize());
        }
        CHECK(hipMemcpy(hc, cyc, blocks * 8, hipMemcpyDeviceToHost));
        double sum = 0; for (int b = 0; b < blocks; ++b) sum += (double)hc[b];
        const double waves_per_cu = 16.0;
        printf("%-14s %-18s %s: %7.1f cycles per wave-instruction per CU, %7.0f cycles per group of four and wave\n", pat[p], ops[op],
               shape == 0 ? "16 x  64 threads" : " 2 x 512 threads", sum / blocks / (ITERS * 4.0) / waves_per_cu, sum / blocks / ITERS);
      }
    }
  }
  return 0;
}
