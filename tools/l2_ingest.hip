// Per-CU ingest of an L2-resident stream: every workgroup (one per CU) reads the SAME region over and over — the weight stream of
// the row-panel kernels (norm_gemm_kernel, resid_panel_kernel: every workgroup sweeps all of W) and the K / V tiles of the encoder
// attention — 16 bytes per lane per load, PF loads per lane in flight.  Prints GB/s per CU for region sizes below and above the
// 4 MB L2 of an XCD, 256 / 512 / 1024 threads, workgroups started in step or at staggered offsets.
//   hipcc --offload-arch=gfx950 -O3 tools/l2_ingest.hip -o /tmp/l2_ingest && /tmp/l2_ingest
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int PF>
__global__ __launch_bounds__(1024) void ingest(const u32x4* __restrict__ buf, int region_v, int total_rounds, int stagger, unsigned* __restrict__ sink) {
  const int T = blockDim.x;
  const int nr = region_v / T;                                  // rounds per sweep of the region
  int pos = stagger ? (int)((blockIdx.x * 7u) % (unsigned)nr) : 0;
  u32x4 r[PF];
  u32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int u = 0; u < PF; ++u) { r[u] = buf[(size_t)pos * T + threadIdx.x]; pos = pos + 1 == nr ? 0 : pos + 1; }
  for (int i = 0; i + PF <= total_rounds; i += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      acc ^= r[u];
      r[u] = buf[(size_t)pos * T + threadIdx.x];
      pos = pos + 1 == nr ? 0 : pos + 1;
    }
  }
#pragma unroll
  for (int u = 0; u < PF; ++u) acc ^= r[u];
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;
}

int main() {
  const size_t BYTES = 64u << 20;
  u32x4* buf; unsigned* sink;
  CK(hipMalloc(&buf, BYTES)); CK(hipMemset(buf, 1, BYTES)); CK(hipMalloc(&sink, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int regions_kb[] = {1152, 4608, 16384};
  for (int rk : regions_kb)
    for (int threads : {256, 512, 1024})
      for (int stagger = 0; stagger < 2; ++stagger) {
        printf("region %5d KB, %4d threads, %s:", rk, threads, stagger ? "staggered" : "in step  ");
        for (int pf : {1, 2, 4, 8}) {
          const int region_v = rk * 1024 / 16;
          const size_t per_wg = 8u << 20;                         // bytes each workgroup reads
          const int rounds = (int)(per_wg / ((size_t)threads * 16));
          float best = 1e30f;
          for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            if (pf == 1) hipLaunchKernelGGL(ingest<1>, dim3(256), dim3(threads), 0, 0, buf, region_v, rounds, stagger, sink);
            else if (pf == 2) hipLaunchKernelGGL(ingest<2>, dim3(256), dim3(threads), 0, 0, buf, region_v, rounds, stagger, sink);
            else if (pf == 4) hipLaunchKernelGGL(ingest<4>, dim3(256), dim3(threads), 0, 0, buf, region_v, rounds, stagger, sink);
            else hipLaunchKernelGGL(ingest<8>, dim3(256), dim3(threads), 0, 0, buf, region_v, rounds, stagger, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
          }
          printf("  PF %d: %5.1f GB/s/CU", pf, per_wg / (best * 1e-3) * 1e-9);
        }
        printf("\n");
      }
  return 0;
}
