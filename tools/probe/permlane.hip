#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  auto r16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  auto r32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[threadIdx.x] = r16[0]; out[64 + threadIdx.x] = r16[1];
  out[128 + threadIdx.x] = r32[0]; out[192 + threadIdx.x] = r32[1];
  float v = (float)threadIdx.x;
  float s = v;
  s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0xB1, 0xF, 0xF, true));
  s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x4E, 0xF, 0xF, true));
  s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x141, 0xF, 0xF, true));
  s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x140, 0xF, 0xF, true));
  out[256 + threadIdx.x] = (unsigned)s;
}
int main() {
  unsigned* d; hipMalloc(&d, 320 * 4); k<<<1, 64>>>(d);
  unsigned h[320]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[5] = {"p16[0]", "p16[1]", "p32[0]", "p32[1]", "rowsum"};
  for (int r = 0; r < 5; ++r) { printf("%s:", names[r]); for (int i = 0; i < 64; ++i) printf(" %u", h[r * 64 + i]); printf("\n"); }
  return 0;
}
