// Microbenchmark: how fast do device-scope 64-bit integer atomics retire when many workgroups add
// into the same small row block (the decoder's fixed-point residual stream: 32 rows x 384 int64)?
// Decides how finely a fused feed-forward kernel may split d_ff (one atomic per output per chunk).
//   hipcc --offload-arch=gfx950 -O3 tools/atomic_rate.hip -o gpurun_out/atomic_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// grid (chunks, mtiles), 384 threads: thread t adds into rows [16*mtile, +rows) column t
__global__ void add_rows(long long* x, int rows, int ld, const int* state) {
  if (state[1]) return;
  long long* p = x + (long long)blockIdx.y * 16 * ld + threadIdx.x;
  const long long v = blockIdx.x + 1;
  for (int i = 0; i < rows; ++i) atomicAdd(reinterpret_cast<unsigned long long*>(p + (long long)i * ld), (unsigned long long)v);
}
// same traffic as plain stores into private slices (the no-contention floor)
__global__ void store_rows(long long* x, int rows, int ld, const int* state) {
  if (state[1]) return;
  long long* p = x + ((long long)blockIdx.x * gridDim.y + blockIdx.y) * 16 * ld + threadIdx.x;
  for (int i = 0; i < rows; ++i) p[(long long)i * ld] = blockIdx.x;
}

static float run(hipStream_t st, int kind, dim3 grid, int rows, long long* x, int* state) {
  const int n_kernels = 50, reps = 40;
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < n_kernels; ++i) {
    if (kind == 0) hipLaunchKernelGGL(add_rows, grid, dim3(384), 0, st, x, rows, 384, state);
    else hipLaunchKernelGGL(store_rows, grid, dim3(384), 0, st, x, rows, 384, state);
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
  long long* x; int* state;
  CK(hipMalloc(&x, 64 << 20)); CK(hipMalloc(&state, 64));
  CK(hipMemset(x, 0, 64 << 20)); CK(hipMemset(state, 0, 64));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const int chunks[] = {1, 8, 18, 36, 72, 144, 288};
  for (int c : chunks)
    for (int rows : {1, 16})
      printf("chunks %3d x 2 mtiles, %2d rows/wg (%7d atomics): atomic %.2f us   store %.2f us per kernel\n", c, rows,
             c * 2 * rows * 384, run(st, 0, dim3(c, 2), rows, x, state), run(st, 1, dim3(c, 2), rows, x, state));
  return 0;
}
