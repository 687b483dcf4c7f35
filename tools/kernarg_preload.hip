// Does kernarg preloading (gfx940+: the first kernel-argument dwords arrive in SGPRs with the wave, no s_load round trip) shorten a
// dependent chain of small kernels like the decode step's?  Each kernel reads a row its predecessor wrote and writes the next one.
//   hipcc --offload-arch=gfx950 -O3 tools/kernarg_preload.hip -o /tmp/kp_plain
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 tools/kernarg_preload.hip -o /tmp/kp_pre
// Prints microseconds per kernel for the struct-argument form (never preloaded: aggregates are passed by reference to the kernarg
// segment) and the scalar-argument form (preloaded in the second build).
#include <hip/hip_runtime.h>
#include <stdio.h>
struct Args { const long long* x; long long* y; const float* w; int d; int pad; const int* state; long long filler[6]; };
__global__ __launch_bounds__(1024) void k_struct(Args a) {
  const int b = blockIdx.x, t = threadIdx.x;
  const int st = *a.state;
  if (t < a.d) a.y[(long long)b * a.d + t] = a.x[(long long)b * a.d + t] + (long long)(a.w[t] * 2.f) + st;
}
__global__ __launch_bounds__(1024) void k_scalar(const long long* x, long long* y, const float* w, int d, const int* state) {
  const int b = blockIdx.x, t = threadIdx.x;
  const int st = *state;
  if (t < d) y[(long long)b * d + t] = x[(long long)b * d + t] + (long long)(w[t] * 2.f) + st;
}
int main() {
  const int B = 256, d = 384, N = 400;
  long long *x, *y; float* w; int* state;
  hipMalloc(&x, B * d * 8); hipMalloc(&y, B * d * 8); hipMalloc(&w, d * 4); hipMalloc(&state, 4);
  hipMemset(x, 0, B * d * 8); hipMemset(y, 0, B * d * 8); hipMemset(w, 0, d * 4); hipMemset(state, 0, 4);
  hipStream_t st; hipStreamCreate(&st);
  for (int form = 0; form < 2; ++form) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < N; ++i) {
      const long long* src = (i & 1) ? y : x; long long* dst = (i & 1) ? x : y;
      if (form == 0) { Args a{src, dst, w, d, 0, state, {0}}; k_struct<<<B, 1024, 0, st>>>(a); }
      else k_scalar<<<B, 1024, 0, st>>>(src, dst, w, d, state);
    }
    hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st);
    for (int r = 0; r < 10; ++r) hipGraphLaunch(ge, st);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s arguments: %.3f us per dependent kernel\n", form == 0 ? "struct" : "scalar", ms * 1e3 / (10.0 * N));
  }
  return 0;
}
