// which way do wave_shr:1 / wave_shl:1 move data on gfx950, and what do lanes without a source keep?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
  const unsigned x = 100 + threadIdx.x;
  out[threadIdx.x] = (unsigned)__builtin_amdgcn_update_dpp((int)7777, (int)x, 0x138, 0xf, 0xf, false);
  out[64 + threadIdx.x] = (unsigned)__builtin_amdgcn_update_dpp((int)9999, (int)x, 0x130, 0xf, 0xf, false);
}
int main() {
  unsigned* d; unsigned h[128];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("wave_shr:1 lanes 0,1,2,31,32,63: %u %u %u %u %u %u\n", h[0], h[1], h[2], h[31], h[32], h[63]);
  printf("wave_shl:1 lanes 0,1,31,32,62,63: %u %u %u %u %u %u\n", h[64], h[65], h[64 + 31], h[64 + 32], h[64 + 62], h[64 + 63]);
  return 0;
}
