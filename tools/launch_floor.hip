// Microbenchmark: cost of a dependent kernel boundary inside a hipGraph on this machine, for the
// launch shapes the decode step uses.  hipcc --offload-arch=gfx950 -O3 tools/launch_floor.hip -o gpurun_out/launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void trivial(float* p, const int* state) {
  if (state[1]) return;
  if (threadIdx.x == 0) p[blockIdx.x] += 1.0f;
}
// one dependent global round trip before the store (what every decode kernel has at least once)
__global__ void one_trip(float* p, const float* q, const int* state) {
  const int t = state[0];
  float v = q[(blockIdx.x * blockDim.x + threadIdx.x + t) & 4095];
  if (v == 12345.f) p[0] = v;
  if (threadIdx.x == 0) p[blockIdx.x] += v;
}

static float run(hipStream_t st, int n_kernels, dim3 grid, dim3 block, int kind, float* p, float* q, int* state, int reps) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < n_kernels; ++i) {
    if (kind == 0) hipLaunchKernelGGL(trivial, grid, block, 0, st, p, state);
    else hipLaunchKernelGGL(one_trip, grid, block, 0, st, p, q, state);
  }
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, st);
  hipEventRecord(e0, st);
  for (int i = 0; i < reps; ++i) hipGraphLaunch(ge, st);
  hipEventRecord(e1, st);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return ms * 1000.f / (reps * n_kernels);
}

int main() {
  float *p, *q; int* state;
  CK(hipMalloc(&p, 1 << 20)); CK(hipMalloc(&q, 1 << 20)); CK(hipMalloc(&state, 64));
  CK(hipMemset(p, 0, 1 << 20)); CK(hipMemset(q, 0, 1 << 20)); CK(hipMemset(state, 0, 64));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  struct { int g, b; } shapes[] = {{1, 1024}, {12, 512}, {48, 512}, {72, 512}, {256, 1024}, {256, 256}};
  for (auto s : shapes)
    for (int kind = 0; kind < 2; ++kind)
      for (int n : {8, 50, 400})
        printf("grid %3d x %4d  %-8s  %3d kernels/graph: %.2f us per kernel\n", s.g, s.b, kind ? "one_trip" : "trivial", n,
               run(st, n, dim3(s.g), dim3(s.b), kind, p, q, state, 2000 / n + 5));
  // eager launches for comparison
  {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st);
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(trivial, dim3(48), dim3(512), 0, st, p, state);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("eager 48 x 512 trivial: %.2f us per kernel\n", ms * 1000.f / 2000);
  }
  return 0;
}
