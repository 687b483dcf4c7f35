// Microbenchmark: what would a PERSISTENT decode step cost per phase?  A grid-wide barrier (device-scope atomics, every
// wave polls with a bounded spin) against the ~2 us dependent-kernel boundary tools/launch_floor.hip measures, with the
// things a decode phase does around it: one dependent read of a row another workgroup (another XCD) wrote in the phase
// before, and a re-read of a per-workgroup slice of "weights" (does the acquire fence throw the L2 contents away?).
//   hipcc --offload-arch=gfx950 -O3 tools/barrier_floor.hip -o gpurun_out/barrier_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int SPIN_CAP = 1 << 16;   // every poll loop ends: a stuck barrier sets *err and the kernel drains

struct Args {
  unsigned* ctr;        // [0]: flat counter; [32 + 32*x]: per-XCD counters; [512]: release word of the two-level barrier
  float* rows;          // [G][64] one row per workgroup, rewritten every phase
  const float* weights; // [G][wfloats]
  int wfloats;
  int phases, mode, trip;
  int* err;
  float* sink;
};

__device__ inline unsigned ld_agent(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// mode 0: flat: one counter, everyone polls it
__device__ inline void barrier_flat(const Args& a, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(a.ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int n = 0;
    while (ld_agent(a.ctr) < target && ++n < SPIN_CAP) __builtin_amdgcn_s_sleep(1);
    if (n >= SPIN_CAP) *a.err = 1;
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
  }
  __syncthreads();
}
// mode 1: the add does not wait for its return (no-return atomic), polls without sleep
__device__ inline void barrier_flat_nosleep(const Args& a, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(a.ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int n = 0;
    while (ld_agent(a.ctr) < target && ++n < SPIN_CAP) {}
    if (n >= SPIN_CAP) *a.err = 1;
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
  }
  __syncthreads();
}
// mode 2: two levels: arrivals counted per XCD (8 counters, 32 arrivals each at G = 256), the last arrival of an XCD adds to
// the global counter, everyone polls the global counter
__device__ inline void barrier_two_level(const Args& a, unsigned phase, unsigned per_xcd, unsigned n_xcd, unsigned xcd) {
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned* lc = a.ctr + 32 + 32 * xcd;
    const unsigned old = __hip_atomic_fetch_add(lc, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == (phase + 1) * per_xcd) __hip_atomic_fetch_add(a.ctr + 512, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int n = 0;
    const unsigned target = (phase + 1) * n_xcd;
    while (ld_agent(a.ctr + 512) < target && ++n < SPIN_CAP) {}
    if (n >= SPIN_CAP) *a.err = 1;
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void persistent(Args a) {
  const unsigned G = gridDim.x, b = blockIdx.x;
  unsigned xcd = b & 7;   // round-robin placement of workgroups over the 8 XCDs (MI355X_MICROARCH.md)
  float acc = 0.f;
  for (int ph = 0; ph < a.phases; ++ph) {
    // (a) dependent read of the row workgroup (b + 37) % G wrote in the previous phase (another XCD), device-coherent
    if (a.trip) {
      const float* src = a.rows + (size_t)((b + 37) % G) * 64;
      acc += __hip_atomic_load(src + (threadIdx.x & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // (b) weights slice (read-only for the whole run)
    for (int i = threadIdx.x; i < a.wfloats; i += blockDim.x) acc += a.weights[(size_t)b * a.wfloats + i];
    // (c) publish this phase's row
    if (threadIdx.x < 64) __hip_atomic_store(a.rows + (size_t)b * 64 + threadIdx.x, acc * 1e-9f + ph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.mode == 0) barrier_flat(a, (ph + 1) * G);
    else if (a.mode == 1) barrier_flat_nosleep(a, (ph + 1) * G);
    else barrier_two_level(a, ph, G / 8, 8, xcd);
  }
  if (acc == 12345.678f) a.sink[0] = acc;
}

int main() {
  unsigned* ctr; float *rows, *weights, *sink; int* err;
  const int GMAX = 512, WMAX = 16384;
  CK(hipMalloc(&ctr, 4096)); CK(hipMalloc(&rows, GMAX * 64 * 4)); CK(hipMalloc(&weights, (size_t)GMAX * WMAX * 4));
  CK(hipMalloc(&sink, 64)); CK(hipMalloc(&err, 64));
  CK(hipMemset(rows, 0, GMAX * 64 * 4)); CK(hipMemset(weights, 0, (size_t)GMAX * WMAX * 4)); CK(hipMemset(err, 0, 64));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[] = {"flat+sleep", "flat", "two-level"};
  for (int G : {64, 128, 256, 512})
    for (int mode = 0; mode < 3; ++mode)
      for (int trip = 0; trip < 2; ++trip)
        for (int wf : {0, 4096, 16384}) {
          if (wf && !trip) continue;
          Args a{ctr, rows, weights, wf, 0, mode, trip, err, sink};
          float us[2];
          for (int k = 0; k < 2; ++k) {
            a.phases = k ? 2200 : 200;
            CK(hipMemsetAsync(ctr, 0, 4096, st));
            hipEventRecord(e0, st);
            hipLaunchKernelGGL(persistent, dim3(G), dim3(256), 0, st, a);
            hipEventRecord(e1, st);
            CK(hipEventSynchronize(e1));
            float ms; hipEventElapsedTime(&ms, e0, e1); us[k] = ms * 1000.f;
          }
          int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
          printf("G %3d  %-10s  trip %d  weights %6d B/WG: %.3f us per phase%s\n", G, names[mode], trip, wf * 4, (us[1] - us[0]) / 2000.f,
                 herr ? "  (SPIN CAP HIT)" : "");
          if (herr) { CK(hipMemset(err, 0, 64)); }
        }
  return 0;
}
