// probe (measurement only): the lane mapping of v_permlane16_swap / v_permlane32_swap on gfx950, and a 32-value wave reduce-scatter
// built from DPP row steps + the two swaps (no ds_bpermute) against the plain sum.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ static inline float dppf(float v, int ctrl) {
  switch (ctrl) {
    case 0xB1: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    case 0x4E: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    case 0x141: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    default: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  }
}
__global__ void probe(float* out) {
  const int lane = threadIdx.x;
  float a = (float)lane, b = 100.f + lane;
  // (inline asm: with hipcc 7.2 the builtin's SECOND result reads the first result's register)
  float x = a, y = b;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
  out[lane] = x;
  out[64 + lane] = y;
  x = a, y = b;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
  out[128 + lane] = x;
  out[192 + lane] = y;
  out[256 + lane] = dppf(a, 0x140);  // row_mirror
  out[320 + lane] = dppf(a, 0x141);  // row_half_mirror
}
int main() {
  float* d; hipMalloc(&d, 384 * 4);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  float h[384]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[6] = {"p16 vdst(a)", "p16 src(b)", "p32 vdst(a)", "p32 src(b)", "row_mirror", "row_half_mirror"};
  for (int k = 0; k < 6; ++k) { printf("%-16s", names[k]); for (int l = 0; l < 64; ++l) printf(" %g", h[k * 64 + l]); printf("\n"); }
  return 0;
}
