// Probe of gfx950's transposing LDS read (ds_read_b64_tr_b16) as the weight-gradient kernel uses it: the LDS image is a straight
// copy of k-major rows ([k][n], 16-bit elements), and one 32x32x16 MFMA operand fragment — lane (r = lane & 31, h = lane >> 5)
// holding k = 8h .. 8h+7 of column n = r — is two transposed reads.  Prints PASS when every lane receives exactly those elements.
//   hipcc --offload-arch=gfx950 -O3 tools/trprobe.hip -o /tmp/trprobe && /tmp/trprobe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
constexpr int PITCH = 160;   // elements per k-row (320 B: four k-rows fall into four different 16-bank groups)

__global__ void probe(short* out) {
  __shared__ __align__(16) short img[16 * PITCH];
  for (int i = threadIdx.x; i < 16 * PITCH; i += 64) img[i] = (short)((i / PITCH) * 256 + (i % PITCH));   // value = k * 256 + n
  __syncthreads();
  const int lane = threadIdx.x, i = lane & 15, q = i >> 2, p = i & 3, g = lane >> 4, h = lane >> 5;
  const short* a = img + (8 * h + q) * PITCH + 16 * (g & 1) + 4 * p;
  const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)a);
  const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(a + 4 * PITCH));
  short* o = out + lane * 8;
  o[0] = lo.x; o[1] = lo.y; o[2] = lo.z; o[3] = lo.w; o[4] = hi.x; o[5] = hi.y; o[6] = hi.z; o[7] = hi.w;
}

int main() {
  short* d; short hbuf[64 * 8];
  if (hipMalloc(&d, sizeof(hbuf)) != hipSuccess) return 1;
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  if (hipMemcpy(hbuf, d, sizeof(hbuf), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane)
    for (int j = 0; j < 8; ++j) {
      const int want = (8 * (lane >> 5) + j) * 256 + (lane & 31), got = hbuf[lane * 8 + j];
      if (got != want) { if (bad < 8) printf("lane %d elem %d: got k=%d n=%d, want k=%d n=%d\n", lane, j, got / 256, got % 256, want / 256, want % 256); ++bad; }
    }
  printf(bad ? "FAIL (%d)\n" : "PASS%.0d\n", bad);
  return bad != 0;
}
