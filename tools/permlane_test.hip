// Check the DPP / v_permlane*_swap lane exchanges of csrc/common.h (lane_xor<M>, wave_sum, wave_max,
// group_sum) against __shfl_xor on arbitrary data, and show the miscompiled builtin form.
//   hipcc --offload-arch=gfx950 -O3 -I music2midi_amd/csrc -I include tools/permlane_test.hip -o /tmp/permlane_test
#include "common.h"
#include <stdio.h>
using namespace m2m;
__device__ inline float xor32_builtin(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
}
__global__ void k(const float* in, float* out) {
  const float v = in[threadIdx.x];
  float* o = out + threadIdx.x;
  o[0 * 64] = __shfl_xor(v, 1, 64);  o[1 * 64] = lane_xor<1>(v);
  o[2 * 64] = __shfl_xor(v, 2, 64);  o[3 * 64] = lane_xor<2>(v);
  o[4 * 64] = __shfl_xor(v, 4, 64);  o[5 * 64] = lane_xor<4>(v);
  o[6 * 64] = __shfl_xor(v, 8, 64);  o[7 * 64] = lane_xor<8>(v);
  o[8 * 64] = __shfl_xor(v, 16, 64); o[9 * 64] = lane_xor<16>(v);
  o[10 * 64] = __shfl_xor(v, 32, 64); o[11 * 64] = lane_xor<32>(v);
  float s = v, m = v;
  for (int x = 32; x > 0; x >>= 1) { s += __shfl_xor(s, x, 64); m = fmaxf(m, __shfl_xor(m, x, 64)); }
  o[12 * 64] = s; o[13 * 64] = wave_sum(v);
  o[14 * 64] = m; o[15 * 64] = wave_max(v);
  float g8 = v, g16 = v;
  for (int x = 1; x < 8; x <<= 1) g8 += __shfl_xor(g8, x, 64);
  for (int x = 1; x < 16; x <<= 1) g16 += __shfl_xor(g16, x, 64);
  o[16 * 64] = g8; o[17 * 64] = group_sum<8>(v);
  o[18 * 64] = g16; o[19 * 64] = group_sum<16>(v);
  o[20 * 64] = v + __shfl_xor(v, 32, 64); o[21 * 64] = xor32_builtin(v);
}
int main() {
  float h[64], o[22 * 64], *di, *dout;
  unsigned st = 12345;
  for (int i = 0; i < 64; ++i) { st = st * 1664525u + 1013904223u; h[i] = (float)(st >> 8) / 65536.0f - 100.0f; }
  hipMalloc(&di, 256); hipMalloc(&dout, sizeof(o));
  hipMemcpy(di, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
  hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  const char* names[] = {"xor1", "xor2", "xor4", "xor8", "xor16", "xor32", "wave_sum", "wave_max", "group_sum<8>", "group_sum<16>", "builtin permlane32_swap(u,u)"};
  int total = 0;
  for (int t = 0; t < 11; ++t) {
    int bad = 0;
    for (int i = 0; i < 64; ++i) bad += o[(2 * t) * 64 + i] != o[(2 * t + 1) * 64 + i];
    printf("%-30s mismatching lanes: %d\n", names[t], bad);
    if (t < 10) total += bad;
  }
  return total != 0;
}
