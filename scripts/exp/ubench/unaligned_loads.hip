// What does a byte-misaligned 16-byte load per lane cost on gfx950?  A streaming copy out[i] = in[i + shift] (16 bytes per
// lane, aligned stores), the source read (a) with one global_load_dwordx4 at the misaligned address, (b) with two ALIGNED
// 16-byte loads and four V_ALIGNBYTEs, (c) aligned (shift 0) as the reference.  GB/s of read + written bytes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_u __attribute__((ext_vector_type(4), aligned(1)));
template <int MODE>
__global__ __launch_bounds__(256) void k(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, size_t n16, int shift) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i < n16; i += stride) {
    u32x4 v;
    if (MODE == 0) {
      v = *reinterpret_cast<const u32x4_u*>(in + i * 16 + shift);
    } else {
      const u32x4* p = reinterpret_cast<const u32x4*>(in + i * 16 + (shift & ~15));
      const u32x4 a = p[0], b = p[1];
      const unsigned s = shift & 3;                        // (byte shift inside a dword; the dword part of the shift is folded below)
      const unsigned w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      const int d = (shift & 15) >> 2;
      v.x = __builtin_amdgcn_alignbyte(w[d + 1], w[d + 0], s);
      v.y = __builtin_amdgcn_alignbyte(w[d + 2], w[d + 1], s);
      v.z = __builtin_amdgcn_alignbyte(w[d + 3], w[d + 2], s);
      v.w = __builtin_amdgcn_alignbyte(w[(d + 4) & 7], w[d + 3], s);
    }
    *reinterpret_cast<u32x4*>(out + i * 16) = v;
  }
}
int main() {
  const size_t bytes = (size_t)4 << 30, n16 = bytes / 16 - 4;
  uint8_t *in, *out;
  CHECK(hipMalloc(&in, bytes + 64)); CHECK(hipMalloc(&out, bytes));
  CHECK(hipMemset(in, 1, bytes + 64));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode)
    for (int shift : {0, 1, 4, 7, 8, 13}) {
      float best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256 * 32), dim3(256), 0, 0, in, out, n16, shift);
        else hipLaunchKernelGGL(k<1>, dim3(256 * 32), dim3(256), 0, 0, in, out, n16, shift);
        hipEventRecord(e1); CHECK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
      }
      printf("%s shift %2d: %7.1f GB/s\n", mode == 0 ? "one misaligned dwordx4 " : "two aligned + alignbyte", shift, 2.0 * bytes / best / 1e6);
    }
  return 0;
}
