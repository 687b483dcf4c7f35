// Microbenchmark: how fast can ONE workgroup per CU pull a K/V-like stream (16 bytes per lane per load, every byte used
// once) — the ceiling of the decode attention kernels' stream phase (DESIGN_HISTORY.md 4.3: ~38 GB/s per CU measured in-kernel).
// Sweeps the number of workgroups (128 = one chain of 16 clips, 256 = the whole chip), the rounds each wave keeps in
// flight, the cache policy (non-temporal / default) and whether the data can sit in the Infinity Cache.
//   hipcc --offload-arch=gfx950 -O3 tools/stream_rate.hip -o /tmp/stream_rate && /tmp/stream_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int PF, bool NT>
__global__ __launch_bounds__(1024) void stream_kernel(const u32x4* __restrict__ buf, size_t wg_stride_v, int rounds, unsigned* __restrict__ sink) {
  const u32x4* p = buf + (size_t)blockIdx.x * wg_stride_v + threadIdx.x;
  u32x4 r[PF];
  u32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int u = 0; u < PF; ++u) r[u] = NT ? __builtin_nontemporal_load(p + (size_t)u * 1024) : p[(size_t)u * 1024];
  int i = 0;
  for (; i + 2 * PF <= rounds; i += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      acc ^= r[u];
      r[u] = NT ? __builtin_nontemporal_load(p + (size_t)(i + PF + u) * 1024) : p[(size_t)(i + PF + u) * 1024];
    }
  }
#pragma unroll
  for (int u = 0; u < PF; ++u) acc ^= r[u];
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;
}

template <int PF, bool NT>
static int run(const u32x4* buf, size_t buf_v, int G, int kb_per_wg, bool rotate, unsigned* sink, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  const int rounds = kb_per_wg * 1024 / (1024 * 16);              // one round = 1024 threads x 16 B = 16 KB
  const size_t per_launch_v = (size_t)G * rounds * 1024;
  const int iters = 200;
  float best = 1e30f, sum = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int it = 0; it < iters; ++it) {
      const size_t off = rotate ? ((size_t)it * per_launch_v) % (buf_v - per_launch_v) : 0;
      hipLaunchKernelGGL((stream_kernel<PF, NT>), dim3(G), dim3(1024), 0, st, buf + off, (size_t)rounds * 1024, rounds, sink);
    }
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best; sum += ms;
  }
  const double us = best * 1000.0 / iters, bytes = (double)per_launch_v * 16;
  printf("G %3d  %3d KB/WG  PF %d  %-7s %-12s: %6.2f us per launch (incl. ~2 us launch gap)  %6.2f TB/s  %5.1f GB/s per CU\n", G, kb_per_wg, PF,
         NT ? "nt" : "default", rotate ? "rotating 1GB" : "same data", us, bytes / us * 1e-6, bytes / us * 1e-3 / G);
  return 0;
}

int main() {
  const size_t bytes = (size_t)1 << 30;
  u32x4* buf; unsigned* sink;
  CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 1, bytes)); CK(hipMalloc(&sink, 64));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t buf_v = bytes / 16;
  for (int G : {128, 256})
    for (int kb : {128, 256, 512})
      for (int rot = 0; rot < 2; ++rot) {
        if (run<1, true>(buf, buf_v, G, kb, rot, sink, st, e0, e1)) return 1;
        if (run<2, true>(buf, buf_v, G, kb, rot, sink, st, e0, e1)) return 1;
        if (run<4, true>(buf, buf_v, G, kb, rot, sink, st, e0, e1)) return 1;
        if (run<8, true>(buf, buf_v, G, kb, rot, sink, st, e0, e1)) return 1;
        if (run<2, false>(buf, buf_v, G, kb, rot, sink, st, e0, e1)) return 1;
        if (run<4, false>(buf, buf_v, G, kb, rot, sink, st, e0, e1)) return 1;
      }
  return 0;
}
