// Microbenchmark for a row-complete product (a workgroup owns whole output rows, so EVERY workgroup streams the whole weight
// matrix): G workgroups of 256 threads all read the SAME buffer (L2-resident after the first touch), 16 bytes per lane per load,
// PF loads in flight per lane.  How many GB/s does one CU pull from L2 — i.e. how long does a workgroup need for 384 x K x 2 bytes?
//   hipcc --offload-arch=gfx950 -O3 tools/shared_stream.hip -o /tmp/shared_stream && /tmp/shared_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int PF>
__global__ __launch_bounds__(256) void shared_stream_kernel(const u32x4* __restrict__ buf, int rounds, unsigned* __restrict__ sink) {
  const u32x4* p = buf + threadIdx.x;
  u32x4 r[PF];
  u32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int u = 0; u < PF; ++u) r[u] = p[(size_t)u * 256];
  int i = 0;
  for (; i + 2 * PF <= rounds; i += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) { acc ^= r[u]; r[u] = p[(size_t)(i + PF + u) * 256]; }
  }
#pragma unroll
  for (int u = 0; u < PF; ++u) acc ^= r[u];
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[blockIdx.x & 15] = acc.x;
}

template <int PF>
static int run(const u32x4* buf, int G, int kb, unsigned* sink, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  const int rounds = kb * 1024 / 4096 / PF * PF;                  // one round = 256 threads x 16 B = 4 KB
  const int iters = 200;
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL((shared_stream_kernel<PF>), dim3(G), dim3(256), 0, st, buf, rounds, sink);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  const double us = best * 1000.0 / iters, bytes = (double)rounds * 4096;
  printf("G %3d  %5d KB per workgroup (shared)  %2d loads in flight per lane (%3d KB per CU): %6.2f us per launch  %6.1f GB/s per CU  %5.2f TB/s total\n", G, kb, PF,
         PF * 4, us, bytes / us * 1e-3, bytes * G / us * 1e-6);
  return 0;
}

int main() {
  u32x4* buf; unsigned* sink;
  CK(hipMalloc(&buf, 8 << 20)); CK(hipMemset(buf, 1, 8 << 20)); CK(hipMalloc(&sink, 64));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int sizes[] = {416, 936, 1248, 1872};                     // (32 + 384) rows x K x 2 bytes for K = 512, 1152, 1536, 2304
  for (int G : {131, 261}) {
    for (int kb : sizes) {
      if (run<4>(buf, G, kb, sink, st, e0, e1)) return 1;
      if (run<8>(buf, G, kb, sink, st, e0, e1)) return 1;
      if (run<16>(buf, G, kb, sink, st, e0, e1)) return 1;
      if (run<24>(buf, G, kb, sink, st, e0, e1)) return 1;
    }
  }
  return 0;
}
