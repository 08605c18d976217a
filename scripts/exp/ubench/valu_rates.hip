// Issue cost of a few gfx950 vector instructions: one wavefront per SIMD runs a long chain of one instruction kind
// (8 independent chains, so latency does not show) between two s_memtime reads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 64
template <int KIND>
__global__ void k(unsigned* out, unsigned long long* cyc, int iters) {
  unsigned a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i;
  unsigned b = out[0] | 0x01020408u;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP; ++r) {
      const int i = r & 7;
      if (KIND == 0) a[i] = a[i] + b;
      if (KIND == 1) a[i] = __builtin_amdgcn_udot4(a[i], b, a[i], false);
      if (KIND == 2) a[i] = __builtin_amdgcn_perm(a[i], b, a[i] & 0x07070707u);
      if (KIND == 3) a[i] = a[i] * b;                                   // v_mul_lo_u32
      if (KIND == 4) a[i] = __umul24(a[i], b);
      if (KIND == 5) a[i] = (a[i] ^ b) & (a[i] >> 3);                    // bitop3-able? two ops at most
      if (KIND == 6) a[i] = __builtin_popcount(a[i]) + b;
      if (KIND == 7) a[i] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)a[i], 0x111, 0xf, 0xf, false) + a[i];
      if (KIND == 8) a[i] = __builtin_amdgcn_mbcnt_lo(a[i], b);
      if (KIND == 9) a[i] = __ffs(a[i]) + b;
      if (KIND == 10) { unsigned long long x = ((unsigned long long)a[i] << 32 | b) >> (a[i] & 31); a[i] = (unsigned)x; }  // 64-bit shift
      if (KIND == 11) a[i] = (a[i] << 3) | b;                            // v_lshl_or
      if (KIND == 12) a[i] = a[i] > b ? a[i] - b : b;                    // cmp + cndmask / max
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  unsigned s = 0;
  for (int i = 0; i < 8; ++i) s ^= a[i];
  out[threadIdx.x + blockIdx.x * blockDim.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  unsigned* out; unsigned long long* cyc;
  hipMalloc(&out, 1 << 20); hipMemset(out, 0, 1 << 20); hipMalloc(&cyc, 8 * 4096);
  const char* names[] = {"v_add_u32", "v_dot4_u32_u8", "v_perm_b32", "v_mul_lo_u32", "v_mul_u32_u24", "xor/and/shift mix", "v_bcnt + add", "dpp mov + add", "v_mbcnt_lo", "ffs + add", "64-bit shift", "v_lshl_or", "cmp+select"};
  const int iters = 2000;
  for (int waves = 1; waves <= 8; waves *= 2) {
    printf("%d wavefront(s) per workgroup of one CU-resident block (block = %d threads)\n", waves * 4, waves * 256);
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define RUN(K) { k<K><<<256 * 2, 1024>>>(out, cyc, 10); hipDeviceSynchronize(); hipEventRecord(e0); k<K><<<256 * 2, 1024>>>(out, cyc, iters); hipEventRecord(e1); hipDeviceSynchronize(); float ms; hipEventElapsedTime(&ms, e0, e1); \
    std::vector<unsigned long long> h(512); hipMemcpy(h.data(), cyc, 512 * 8, hipMemcpyDeviceToHost); double c = 0; for (auto x : h) c += x; c /= 512; \
    /* 2 blocks of 16 waves per CU = 8 waves per SIMD; per SIMD instructions = 8 waves * iters * REP */ \
    printf("%-20s %.3f ms  -> %.2f ns per wave-instruction per SIMD; memtime ticks per instr per SIMD %.3f\n", names[K], ms, ms * 1e6 / (8.0 * iters * REP), c / (8.0 * iters * REP)); }
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12)
  return 0;
}
