// Do the registers of a wavefront that is allocated 24 VGPRs (three granules of 8) survive while other wavefronts come and go on
// its SIMD?  Every lane writes a value of its own into v1..v23, sleeps, and counts the registers that changed.  The same kernel
// with a clobber of v31 (32 VGPRs allocated) is the control.  (scripts/exp/vgpr24/run.py)
#include <hip/hip_runtime.h>

#define BODY(TOP)                                                                                            \
  unsigned bad = 0, tmp = 0;                                                                                    \
  const unsigned gid = blockIdx.x * 64 + threadIdx.x;                                                           \
  const unsigned seed = (gid * 2654435761u) >> 8;                                                               \
  asm volatile(                                                                                                 \
      "v_add_u32 v1, %2, 1\n v_add_u32 v2, %2, 2\n v_add_u32 v3, %2, 3\n v_add_u32 v4, %2, 4\n"               \
      "v_add_u32 v5, %2, 5\n v_add_u32 v6, %2, 6\n v_add_u32 v7, %2, 7\n v_add_u32 v8, %2, 8\n"               \
      "v_add_u32 v9, %2, 9\n v_add_u32 v10, %2, 10\n v_add_u32 v11, %2, 11\n v_add_u32 v12, %2, 12\n"         \
      "v_add_u32 v13, %2, 13\n v_add_u32 v14, %2, 14\n v_add_u32 v15, %2, 15\n v_add_u32 v16, %2, 16\n"       \
      "v_add_u32 v17, %2, 17\n v_add_u32 v18, %2, 18\n v_add_u32 v19, %2, 19\n"                               \
      "s_mov_b32 s20, %3\n"                                                                                     \
      "1:\n s_sleep 4\n v_add_u32 v19, v19, 0\n s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n" \
      "v_mov_b32 %0, 0\n"                                                                                       \
      "v_sub_u32 %1, v1, %2\n v_cmp_ne_u32 vcc, 1, %1\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"                   \
      "v_sub_u32 %1, v2, %2\n v_cmp_ne_u32 vcc, 2, %1\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"                   \
      "v_sub_u32 %1, v3, %2\n v_cmp_ne_u32 vcc, 3, %1\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"                   \
      "v_sub_u32 %1, v4, %2\n v_cmp_ne_u32 vcc, 4, %1\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"                   \
      "v_sub_u32 %1, v5, %2\n v_cmp_ne_u32 vcc, 5, %1\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"                   \
      "v_sub_u32 %1, v6, %2\n v_cmp_ne_u32 vcc, 6, %1\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"                   \
      "v_sub_u32 %1, v7, %2\n v_cmp_ne_u32 vcc, 7, %1\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"                   \
      "v_sub_u32 %1, v8, %2\n v_cmp_ne_u32 vcc, 8, %1\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"                   \
      "v_sub_u32 %1, v9, %2\n v_cmp_ne_u32 vcc, 9, %1\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"                   \
      "v_sub_u32 %1, v10, %2\n v_cmp_ne_u32 vcc, 10, %1\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"                 \
      "v_sub_u32 %1, v11, %2\n v_cmp_ne_u32 vcc, 11, %1\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"                 \
      "v_sub_u32 %1, v12, %2\n v_cmp_ne_u32 vcc, 12, %1\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"                 \
      "v_sub_u32 %1, v13, %2\n v_cmp_ne_u32 vcc, 13, %1\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"                 \
      "v_sub_u32 %1, v14, %2\n v_cmp_ne_u32 vcc, 14, %1\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"                 \
      "v_sub_u32 %1, v15, %2\n v_cmp_ne_u32 vcc, 15, %1\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"                 \
      "v_sub_u32 %1, v16, %2\n v_cmp_ne_u32 vcc, 16, %1\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"                 \
      "v_sub_u32 %1, v17, %2\n v_cmp_ne_u32 vcc, 17, %1\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"                 \
      "v_sub_u32 %1, v18, %2\n v_cmp_ne_u32 vcc, 18, %1\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"                 \
      "v_sub_u32 %1, v19, %2\n v_cmp_ne_u32 vcc, 19, %1\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"                 \
      : "=&v"(bad), "=&v"(tmp)                                                                                  \
      : "v"(seed), "s"(spins)                                                                                   \
      : "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", \
        "v19", "s20", "vcc", TOP);                                                                     \
  out[gid] = bad;

// v20..v23 are left to the compiler (gid, seed, bad, tmp): 24 allocated in all
__global__ __launch_bounds__(64) void keep24(unsigned* out, int spins) { BODY("v23") }
__global__ __launch_bounds__(64) void keep32(unsigned* out, int spins) { BODY("v31") }

extern "C" int run(int which, unsigned* d_out, int n_waves, int spins, void* stream) {
  if (which == 24) hipLaunchKernelGGL(keep24, dim3(n_waves), dim3(64), 0, (hipStream_t)stream, d_out, spins);
  else hipLaunchKernelGGL(keep32, dim3(n_waves), dim3(64), 0, (hipStream_t)stream, d_out, spins);
  return (int)hipGetLastError();
}
