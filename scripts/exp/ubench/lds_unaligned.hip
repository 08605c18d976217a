// Does LDS take byte-misaligned 16-byte (and 8-byte) accesses on gfx950, and at what rate?  Every lane writes 16 bytes at
// LDS byte address 16 * lane * 2 + shift (ds_write_b128 through inline asm: the compiler splits what it cannot prove
// aligned), reads them back aligned, and the result is checked; then a timed loop of such writes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d (%s) at %d\n", (int)e, hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void check_k(int shift, unsigned* out, int wide) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[256 * 32 + 64];
  for (int i = threadIdx.x; i < (256 * 32 + 64) / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = 0;
  __syncthreads();
  const unsigned addr = (unsigned)(size_t)lds + threadIdx.x * 32 + shift;      // (LDS addresses are 32-bit offsets)
  u32x4 v = {threadIdx.x * 4 + 1, threadIdx.x * 4 + 2, threadIdx.x * 4 + 3, threadIdx.x * 4 + 4};
  if (wide) asm volatile("ds_write_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : : "v"(addr), "v"(v) : "memory");
  else {
    unsigned long long lo = ((unsigned long long)v.y << 32) | v.x;
    asm volatile("ds_write_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : : "v"(addr), "v"(lo) : "memory");
  }
  __syncthreads();
  unsigned bad = 0;
  for (int b = 0; b < (wide ? 16 : 8); ++b) {
    const unsigned want = (reinterpret_cast<unsigned char*>(&v))[b];
    if (lds[threadIdx.x * 32 + shift + b] != want) bad = 1;
  }
  if (bad) atomicAdd(out, 1u);
  u32x4 r;
  asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
  if (wide && (r.x != v.x || r.y != v.y || r.z != v.z || r.w != v.w)) atomicAdd(out + 1, 1u);
}
template <int SHIFT>
__global__ __launch_bounds__(256) void time_k(unsigned* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[256 * 32 + 64];
  const unsigned addr = (unsigned)(size_t)lds + threadIdx.x * 32 + SHIFT;
  u32x4 v = {threadIdx.x, 2, 3, 4};
  for (int i = 0; i < iters; ++i) {
    asm volatile("ds_write_b128 %0, %1" : : "v"(addr), "v"(v) : "memory");
    v.x += 1;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lds[threadIdx.x] == 255 && iters < 0) out[0] = 1;
}
int main() {
  unsigned* d; CHECK(hipMalloc(&d, 16)); unsigned h[2];
  for (int wide = 0; wide < 2; ++wide)
    for (int shift : {0, 1, 2, 4, 7, 8, 12, 13}) {
      CHECK(hipMemset(d, 0, 8));
      hipLaunchKernelGGL(check_k, dim3(1), dim3(256), 0, 0, shift, d, wide);
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemcpy(h, d, 8, hipMemcpyDeviceToHost));
      printf("%s shift %2d: %s (write mismatches %u, read mismatches %u)\n", wide ? "b128" : "b64 ", shift, (h[0] | h[1]) ? "WRONG" : "ok", h[0], h[1]);
    }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int s = 0; s < 4; ++s) {
    float ms;
    hipEventRecord(e0);
    if (s == 0) hipLaunchKernelGGL(time_k<0>, dim3(256), dim3(256), 0, 0, d, iters);
    if (s == 1) hipLaunchKernelGGL(time_k<1>, dim3(256), dim3(256), 0, 0, d, iters);
    if (s == 2) hipLaunchKernelGGL(time_k<4>, dim3(256), dim3(256), 0, 0, d, iters);
    if (s == 3) hipLaunchKernelGGL(time_k<8>, dim3(256), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); CHECK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms, e0, e1);
    const int sh[4] = {0, 1, 4, 8};
    printf("ds_write_b128 at shift %d: %.1f cycles per wave instruction and CU (4 waves per CU)\n", sh[s], ms * 1e-3 * 2.4e9 / (iters * 4.0));
  }
  return 0;
}
