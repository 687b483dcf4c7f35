// Do dependent-kernel chains on DIFFERENT streams overlap on this machine?  Each chain is a hipGraph of N kernels
// that spin ~D us on G workgroups; we time 1 chain alone and 2 / 4 chains launched together.
// hipcc --offload-arch=gfx950 -O3 tools/stream_overlap.hip -o tools/stream_overlap.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void spin(float* p, int us) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)us * 100) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) p[blockIdx.x] += 1.0f;
}
int main() {
  float* p; (void)hipMalloc(&p, 1 << 20); (void)hipMemset(p, 0, 1 << 20);
  const int NCH = 4, N = 100;
  hipStream_t st[NCH]; hipGraph_t g[NCH]; hipGraphExec_t ge[NCH];
  for (int grid : {8, 64, 128, 256}) for (int thr : {256, 1024}) for (int us : {3, 10}) {
    for (int c = 0; c < NCH; ++c) {
      (void)hipStreamCreateWithFlags(&st[c], hipStreamNonBlocking);
      (void)hipStreamBeginCapture(st[c], hipStreamCaptureModeThreadLocal);
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin, dim3(grid), dim3(thr), 0, st[c], p + c * 4096, us);
      (void)hipStreamEndCapture(st[c], &g[c]);
      (void)hipGraphInstantiate(&ge[c], g[c], nullptr, nullptr, 0);
    }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int nch : {1, 2, 4}) {
      for (int c = 0; c < nch; ++c) (void)hipGraphLaunch(ge[c], st[c]);
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0, st[0]);
      for (int c = 1; c < nch; ++c) (void)hipStreamWaitEvent(st[c], e0, 0);
      for (int rep = 0; rep < 3; ++rep) for (int c = 0; c < nch; ++c) (void)hipGraphLaunch(ge[c], st[c]);
      for (int c = 1; c < nch; ++c) { hipEvent_t e; (void)hipEventCreate(&e); (void)hipEventRecord(e, st[c]); (void)hipStreamWaitEvent(st[0], e, 0); }
      (void)hipEventRecord(e1, st[0]);
      (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      printf("grid %3d x %4d, %2d us/kernel: %d chain(s): %.2f us per kernel-slot (ideal %d + ~2)\n", grid, thr, us, nch, ms * 1000.f / (3 * N), us);
    }
    for (int c = 0; c < NCH; ++c) { (void)hipGraphExecDestroy(ge[c]); (void)hipGraphDestroy(g[c]); (void)hipStreamDestroy(st[c]); }
  }
  return 0;
}
