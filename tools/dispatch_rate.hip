// How long does the chip take just to START the workgroups of a launch?  Kernels whose workgroups do (almost) nothing, 256 threads
// each, grids of 256 .. 8192 workgroups, back to back on one stream; and the same with 20 KB / 36 KB of LDS per workgroup (the
// occupancy limits of the small-tile GEMM and the stripe kernels).  hipcc --offload-arch=gfx950 -O3 tools/dispatch_rate.hip -o /tmp/dr && /tmp/dr
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void nop_kernel(const float* __restrict__ in, float* __restrict__ out, int work) {
  extern __shared__ float lds[];
  float v = in[threadIdx.x & 63];
  for (int i = 0; i < work; ++i) v = v * 1.0001f + 0.5f;        // `work` dependent FMAs: ~4 cycles each
  if (v == 12345.678f) { lds[threadIdx.x] = v; out[blockIdx.x] = lds[255 - threadIdx.x]; }
}

int main() {
  float *in, *out;
  CK(hipMalloc(&in, 4096)); CK(hipMalloc(&out, 1 << 20)); CK(hipMemset(in, 0, 4096));
  CK(hipFuncSetAttribute((const void*)nop_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int lds : {0, 20 * 1024, 36 * 1024})
    for (int work : {0, 1000})
      for (int n : {256, 512, 1024, 2048, 4096, 8192}) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
          CK(hipEventRecord(e0, 0));
          for (int it = 0; it < 50; ++it) hipLaunchKernelGGL(nop_kernel, dim3(n), dim3(256), lds, 0, in, out, work);
          CK(hipEventRecord(e1, 0));
          CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        printf("LDS %2d KB  work %4d  %5d workgroups: %6.2f us per launch (back to back)  = %5.1f ns per workgroup\n", lds / 1024, work, n, best * 1e3 / 50,
               best * 1e6 / 50 / n);
      }
  return 0;
}
